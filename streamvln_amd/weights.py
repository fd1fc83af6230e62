"""Seeded, version-proof weight synthesis (no checkpoint exists offline).

Tensor names follow the HF state-dict layout of the reference checkpoint
(`model.vision_tower.vision_tower.vision_model.*`, `model.mm_projector.{0,2}.*`,
`model.layers.*`, `model.norm`, `lm_head`), so a real checkpoint can be fed through the
same `set_tensor` path later.

Every element is a pure function of (global seed, tensor name, flat index):

    z  = splitmix64(seed_t + idx * GOLDEN)         seed_t = fnv1a64(name) ^ splitmix64(seed)
    m  = z >> 40                                   24-bit integer
    v  = (float32(m) - 2^23) * (a / 2^23)          uniform in [-a, a), exact in fp32
    w  = base + v                                  base = 1 for norm gains, else 0

    W  = bf16(w)                                   round-to-nearest-even: weights are bf16 values

The identical arithmetic runs on the device (`csrc/misc.hip` synth_kernel), in numpy here, so the
HIP engine (bf16 or fp32 parity mode), the CPU oracle and the imported reference all see
bit-identical weights without 15 GB crossing PCIe.
"""
from __future__ import annotations

from dataclasses import dataclass
from typing import Dict, Iterable, List, Tuple

import numpy as np

from .config import StreamVLNConfig

_MASK = (1 << 64) - 1
_GOLDEN = 0x9E3779B97F4A7C15
VT = "model.vision_tower.vision_tower.vision_model."


def fnv1a64(name: str) -> int:
    h = 0xCBF29CE484222325
    for b in name.encode("utf-8"):
        h ^= b
        h = (h * 0x100000001B3) & _MASK
    return h


def splitmix64_int(z: int) -> int:
    z &= _MASK
    z = ((z ^ (z >> 30)) * 0xBF58476D1CE4E5B9) & _MASK
    z = ((z ^ (z >> 27)) * 0x94D049BB133111EB) & _MASK
    return z ^ (z >> 31)


def tensor_seed(seed: int, name: str) -> int:
    return fnv1a64(name) ^ splitmix64_int(seed)


def _splitmix64(z: np.ndarray) -> np.ndarray:
    z = (z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
    z = (z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
    return z ^ (z >> np.uint64(31))


def synth_flat(seed_t: int, start: int, count: int, half_width: float, base: float) -> np.ndarray:
    """Elements [start, start+count) of a synthetic tensor, fp32."""
    with np.errstate(over="ignore"):
        idx = np.arange(start, start + count, dtype=np.uint64)
        z = _splitmix64(np.uint64(seed_t) + idx * np.uint64(_GOLDEN))
    m = (z >> np.uint64(40)).astype(np.float32)
    step = np.float32(half_width) / np.float32(8388608.0)
    v = (m - np.float32(8388608.0)) * step
    if base != 0.0:
        v = np.float32(base) + v
    return v.astype(np.float32)


def round_to_bf16(x: np.ndarray) -> np.ndarray:
    """fp32 -> bf16 (round-to-nearest-even) -> fp32, matching the device cast."""
    u = np.ascontiguousarray(x, dtype=np.float32).view(np.uint32)
    r = (u + np.uint32(0x7FFF) + ((u >> np.uint32(16)) & np.uint32(1))) & np.uint32(0xFFFF0000)
    return r.view(np.float32)


def to_bf16_bits(x: np.ndarray) -> np.ndarray:
    u = np.ascontiguousarray(x, dtype=np.float32).view(np.uint32)
    return ((u + np.uint32(0x7FFF) + ((u >> np.uint32(16)) & np.uint32(1))) >> np.uint32(16)).astype(np.uint16)


@dataclass(frozen=True)
class TensorSpec:
    name: str
    shape: Tuple[int, ...]
    half_width: float   # uniform half width `a`
    base: float         # 1.0 for norm gains

    @property
    def numel(self) -> int:
        n = 1
        for s in self.shape:
            n *= s
        return n


def _lin(name, out_f, in_f, bias, gain=1.0) -> List[TensorSpec]:
    a = gain * (3.0 ** 0.5) / (in_f ** 0.5)
    specs = [TensorSpec(name + ".weight", (out_f, in_f), a, 0.0)]
    if bias:
        specs.append(TensorSpec(name + ".bias", (out_f,), 0.05, 0.0))
    return specs


def tensor_specs(cfg: StreamVLNConfig) -> List[TensorSpec]:
    """Canonical (HF-layout) tensors of the path, in a fixed order."""
    s: List[TensorSpec] = []
    p = cfg.v_patch
    s.append(TensorSpec(VT + "embeddings.patch_embedding.weight", (cfg.v_hidden, 3, p, p),
                        (3.0 ** 0.5) / (cfg.patch_k ** 0.5), 0.0))
    s.append(TensorSpec(VT + "embeddings.patch_embedding.bias", (cfg.v_hidden,), 0.05, 0.0))
    s.append(TensorSpec(VT + "embeddings.position_embedding.weight", (cfg.v_tokens, cfg.v_hidden), 0.05, 0.0))
    for i in range(cfg.v_layers):
        L = f"{VT}encoder.layers.{i}."
        s.append(TensorSpec(L + "layer_norm1.weight", (cfg.v_hidden,), 0.1, 1.0))
        s.append(TensorSpec(L + "layer_norm1.bias", (cfg.v_hidden,), 0.05, 0.0))
        for nm in ("q_proj", "k_proj", "v_proj", "out_proj"):
            s += _lin(L + "self_attn." + nm, cfg.v_hidden, cfg.v_hidden, True)
        s.append(TensorSpec(L + "layer_norm2.weight", (cfg.v_hidden,), 0.1, 1.0))
        s.append(TensorSpec(L + "layer_norm2.bias", (cfg.v_hidden,), 0.05, 0.0))
        s += _lin(L + "mlp.fc1", cfg.v_inter, cfg.v_hidden, True)
        s += _lin(L + "mlp.fc2", cfg.v_hidden, cfg.v_inter, True)
    s += _lin("model.mm_projector.0", cfg.hidden, cfg.v_hidden, True)
    s += _lin("model.mm_projector.2", cfg.hidden, cfg.hidden, True, gain=0.25)
    s.append(TensorSpec("model.embed_tokens.weight", (cfg.vocab, cfg.hidden), 0.05, 0.0))
    for i in range(cfg.layers):
        L = f"model.layers.{i}."
        s.append(TensorSpec(L + "input_layernorm.weight", (cfg.hidden,), 0.1, 1.0))
        s += _lin(L + "self_attn.q_proj", cfg.q_dim, cfg.hidden, True)
        s += _lin(L + "self_attn.k_proj", cfg.kv_dim, cfg.hidden, True)
        s += _lin(L + "self_attn.v_proj", cfg.kv_dim, cfg.hidden, True)
        s += _lin(L + "self_attn.o_proj", cfg.hidden, cfg.q_dim, False, gain=0.5)
        s.append(TensorSpec(L + "post_attention_layernorm.weight", (cfg.hidden,), 0.1, 1.0))
        s += _lin(L + "mlp.gate_proj", cfg.inter, cfg.hidden, False)
        s += _lin(L + "mlp.up_proj", cfg.inter, cfg.hidden, False)
        s += _lin(L + "mlp.down_proj", cfg.hidden, cfg.inter, False, gain=0.5)
    s.append(TensorSpec("model.norm.weight", (cfg.hidden,), 0.1, 1.0))
    s += _lin("lm_head", cfg.vocab, cfg.hidden, False)
    return s


def synth_tensor(spec: TensorSpec, seed: int, bf16_round: bool = True, chunk: int = 1 << 24) -> np.ndarray:
    seed_t = tensor_seed(seed, spec.name)
    out = np.empty(spec.numel, dtype=np.float32)
    for s0 in range(0, spec.numel, chunk):                  # chunked: the 545 M-element vocab tensors stay within a few 100 MB of scratch
        n = min(chunk, spec.numel - s0)
        v = synth_flat(seed_t, s0, n, spec.half_width, spec.base)
        out[s0:s0 + n] = round_to_bf16(v) if bf16_round else v
    return out.reshape(spec.shape)


def synth_state_dict(cfg: StreamVLNConfig, seed: int, bf16_round: bool = True,
                     only: Iterable[str] | None = None, workers: int = 1) -> Dict[str, np.ndarray]:
    """All canonical tensors as fp32 numpy arrays (bf16-rounded values when `bf16_round`).  workers > 1: the 16 M-element chunks of all
    tensors are generated by a thread pool (numpy releases the GIL inside its loops): the 7.6 B values of the true-size model take
    about a minute on 16 threads instead of several."""
    want = None if only is None else set(only)
    specs = [s for s in tensor_specs(cfg) if want is None or s.name in want]
    if workers <= 1:
        return {s.name: synth_tensor(s, seed, bf16_round) for s in specs}
    from concurrent.futures import ThreadPoolExecutor
    chunk = 1 << 24
    flat = {s.name: np.empty(s.numel, dtype=np.float32) for s in specs}

    def work(job):
        s, s0 = job
        n = min(chunk, s.numel - s0)
        v = synth_flat(tensor_seed(seed, s.name), s0, n, s.half_width, s.base)
        flat[s.name][s0:s0 + n] = round_to_bf16(v) if bf16_round else v
    jobs = [(s, s0) for s in specs for s0 in range(0, s.numel, chunk)]
    with ThreadPoolExecutor(max_workers=workers) as ex:
        list(ex.map(work, jobs))
    return {s.name: flat[s.name].reshape(s.shape) for s in specs}
