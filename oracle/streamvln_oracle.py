"""CPU oracle: a plain fp32 restatement of StreamVLN's streaming-inference path.

TEST INFRASTRUCTURE ONLY.  Nothing under `streamvln_amd/` imports this file; only
`tests/`, `__graft_entry__.smoke()` and `bench.py`'s `cpu_baseline` leg may.  The shipped
path is the HIP engine behind `include/streamvln_hip.h` and fails loudly without it.

What it restates (reference file:line under /root/reference unless noted):
  preprocess            llava/model/multimodal_encoder/siglip_encoder.py:47-67
  SigLIP embeddings     siglip_encoder.py:169-174
  SigLIP encoder layer  siglip_encoder.py:197-305 (eager attention, fp32 softmax)
  vision tower output   siglip_encoder.py:563-589 (hidden_states[-1], no post_layernorm)
  mm_projector          llava/model/multimodal_projector/builder.py:41-48 (Linear-GELU(erf)-Linear)
  get_2dPool            streamvln/model/stream_video_vln.py:53-73 (bilinear, align_corners=False)
  encode_rgbd           stream_video_vln.py:102-142
  token splice          stream_video_vln.py:144-291 (batch of one env; :241-244 per-turn truncation to
                        config.tokenizer_model_max_length)
  turn protocol         stream_video_vln.py:353-479 + transformers 4.45.1 GenerationMixin
                        (greedy; cache_position = arange(L_total)[P:])
  repetition penalty    transformers RepetitionPenaltyLogitsProcessor (third-party, generation/logits_process.py;
                        GenerationMixin applies generation_config.repetition_penalty under greedy decoding too)
  Qwen2 decoder         transformers Qwen2 equations (third-party, pinned 4.45.1 in the
                        reference's requirements.txt:140; same math as the container copy
                        transformers/models/qwen2/modeling_qwen2.py:35-48,91-135,150-173,238-252,269-299)

Parity pin: `oracle/make_golden.py` imports the reference's own modules in the build
container, loads the same synthetic weights into them and asserts this file reproduces
their outputs (vision tower, projector+pool, splice, multi-turn greedy decode) before it
writes `tests/golden/*.npz`.  The reference has no tests or fixtures of its own
(SURVEY.md F8), so those generated vectors are the pin.

All arithmetic is fp32 on the CPU via torch tensor ops (matmul / exp / erf / tanh); no
nn.Module of transformers or of the reference is used here.

Quantisation-emulating modes (`Fp8Emu`; round 4): NO REFERENCE COUNTERPART -- the reference is bf16 only
(streamvln_eval.py:526).  They restate the numeric scheme of the engine's opt-in e4m3 modes (svln_set_fp8_decode /
svln_set_fp8_gemm: per-row scales amax / 448, round-to-nearest-even OCP e4m3, fp32 accumulate) so that the HIP fp8 kernels are
compared with the SAME scheme instead of with bf16; parity of these modes is therefore pinned by this project only.
"""
from __future__ import annotations

import math
import threading
from typing import Dict, List, Optional, Sequence

import numpy as np
import torch

IMAGE_TOKEN_INDEX = -200    # streamvln/utils/utils.py:9
MEMORY_TOKEN_INDEX = -300   # streamvln/utils/utils.py:15
VT = "model.vision_tower.vision_tower.vision_model."

W = Dict[str, torch.Tensor]


# ----------------------------------------------------------------------------- preprocess
def siglip_preprocess(rgb: np.ndarray, size: int = 384) -> np.ndarray:
    """uint8 [H,W,3] -> fp32 [3,size,size] in [-1,1].  siglip_encoder.py:47-67:
    PIL bicubic resize to (size,size) (aspect not preserved), *1/255 (in fp64, cast fp32),
    (x-0.5)/0.5, channels first."""
    from PIL import Image
    img = Image.fromarray(np.asarray(rgb, dtype=np.uint8)).convert("RGB")
    img = img.resize((size, size), resample=Image.BICUBIC)
    a = np.asarray(img).astype(np.float64) * (1.0 / 255.0)
    a = a.astype(np.float32)
    a = (a - np.float32(0.5)) / np.float32(0.5)
    return np.ascontiguousarray(a.transpose(2, 0, 1))


# ----------------------------------------------------------------------------- small math
def layer_norm(x, g, b, eps):
    mu = x.mean(-1, keepdim=True)
    var = ((x - mu) ** 2).mean(-1, keepdim=True)
    return (x - mu) * torch.rsqrt(var + eps) * g + b


def gelu_tanh(x):           # ACT2FN["gelu_pytorch_tanh"], siglip_encoder.py:83,247
    return 0.5 * x * (1.0 + torch.tanh(math.sqrt(2.0 / math.pi) * (x + 0.044715 * x ** 3)))


def gelu_erf(x):            # nn.GELU() default, multimodal_projector/builder.py:45
    return 0.5 * x * (1.0 + torch.erf(x / math.sqrt(2.0)))


def silu(x):
    return x / (1.0 + torch.exp(-x))


def rms_norm(x, g, eps):    # Qwen2RMSNorm (modeling_qwen2.py:238-252)
    var = (x * x).mean(-1, keepdim=True)
    return g * (x * torch.rsqrt(var + eps))


def linear(x, w, b=None):
    y = x @ w.t()
    return y if b is None else y + b


# ----------------------------------------------------------------------------- opt-in e4m3 modes (no reference counterpart)
def qdq_e4m3_rows(x: torch.Tensor) -> torch.Tensor:
    """Per-row e4m3 quantise -> dequantise, the arithmetic of gemv.hip `quant_fp8_rows_kernel`: scale = amax / 448 (1 for an all-zero
    row), q = e4m3_rne(clamp(x * (1 / scale), +-448)), value = q * scale.  fp32 in, fp32 out."""
    amax = x.abs().amax(-1, keepdim=True)
    sc = torch.where(amax > 0, amax / 448.0, torch.ones_like(amax))
    inv = 1.0 / sc
    q = (x * inv).clamp(-448.0, 448.0).to(torch.float8_e4m3fn).to(torch.float32)
    return q * sc


_DQ_LOCK = threading.Lock()


class Fp8Emu:
    """The engine's two opt-in e4m3 modes as a CPU restatement (extension, no reference counterpart):
      decode  (svln_set_fp8_decode): the single-token decode step's four projections and EVERY lm_head product read per-row e4m3
              copies of the weights; activations stay in the engine dtype (weight-only, "w8").
      gemm    (svln_set_fp8_gemm): the multi-row (prefill) products q/k/v, o, gate/up, down multiply per-row e4m3 activations with
              the same e4m3 weight copies ("w8a8"); bias / residual / norms / attention / lm_head are not quantised.
    act_dtype: the engine rounds activations to its storage type before they are quantised (bf16: the modes exist on bf16 engines
    only); None keeps fp32.
    store_dtype (default None = fp32 throughout): round every tensor the engine STORES between kernels to this type as well (norm
    outputs, q / k / v before and after RoPE, the attention output, the residual stream after each residual add, the SwiGLU product) --
    the unfused kernel sequence of a prefill with more than 256 rows.  An activation quantiser turns a bf16-sized difference delta of
    its input into sqrt(delta * step) (a fraction delta / step of the elements moves by a whole e4m3 step), so an fp32-storage
    emulation sits 2-4 % of a layer's contribution away from the bf16 engine for that reason alone; with the storage rounding restated
    the comparison is sharp again (tests/test_fp8_gpu.py, teacher-forced layers)."""

    LINEARS = ("self_attn.q_proj", "self_attn.k_proj", "self_attn.v_proj", "self_attn.o_proj", "mlp.gate_proj", "mlp.up_proj",
               "mlp.down_proj")

    def __init__(self, decode: bool = False, gemm: bool = False, act_dtype: Optional[torch.dtype] = torch.bfloat16,
                 store_dtype: Optional[torch.dtype] = None):
        self.decode, self.gemm, self.act_dtype, self.store_dtype = bool(decode), bool(gemm), act_dtype, store_dtype
        self._dq: Dict[str, torch.Tensor] = {}

    def store(self, x: torch.Tensor) -> torch.Tensor:
        return x if self.store_dtype is None else x.to(self.store_dtype).to(torch.float32)

    def weight(self, w: W, name: str) -> torch.Tensor:
        """dequantised e4m3 copy of w[name] (q/k/v rows are quantised row by row, so the engine's fused qkv matrix gives the same)"""
        t = self._dq.get(name)
        if t is None:
            with _DQ_LOCK:                      # (tests run several oracle episodes on threads over one shared cache of copies)
                t = self._dq.get(name)
                if t is None:
                    t = self._dq[name] = qdq_e4m3_rows(w[name])
        return t

    def act(self, x: torch.Tensor) -> torch.Tensor:
        if self.act_dtype is not None:
            x = x.to(self.act_dtype).to(torch.float32)
        return qdq_e4m3_rows(x)

    def lin(self, w: W, name: str, x: torch.Tensor, b, phase: str) -> torch.Tensor:
        """phase 'prefill' = a multi-row product of svln_generate's prefill; 'decode' = the single-row decode step"""
        if phase == "prefill" and self.gemm:
            return linear(self.act(x), self.weight(w, name), b)
        if phase == "decode" and self.decode:
            return linear(x, self.weight(w, name), b)
        return linear(x, w[name], b)


# ----------------------------------------------------------------------------- vision
def siglip_embeddings(w: W, cfg, pixels: torch.Tensor) -> torch.Tensor:
    """[F,3,S,S] -> [F,729,Hv]: valid 14x14/14 conv + bias, flatten, + pos-emb (siglip_encoder.py:169-174).
    384 = 27*14 + 6: the last 6 rows/cols are dropped by the valid conv."""
    F_, C, _, _ = pixels.shape
    p, side = cfg.v_patch, cfg.v_side
    x = pixels[:, :, : side * p, : side * p]
    x = x.reshape(F_, C, side, p, side, p).permute(0, 2, 4, 1, 3, 5).reshape(F_, side * side, C * p * p)
    wk = w[VT + "embeddings.patch_embedding.weight"].reshape(cfg.v_hidden, -1)
    y = x @ wk.t() + w[VT + "embeddings.patch_embedding.bias"]
    return y + w[VT + "embeddings.position_embedding.weight"][None]


def siglip_attention(w: W, cfg, L: str, x: torch.Tensor) -> torch.Tensor:
    """siglip_encoder.py:197-239: 16 heads x hd 72, scale hd^-0.5, softmax in fp32, all biases."""
    F_, S, Hv = x.shape
    nh, hd = cfg.v_heads, cfg.v_head_dim
    q = linear(x, w[L + "self_attn.q_proj.weight"], w[L + "self_attn.q_proj.bias"])
    k = linear(x, w[L + "self_attn.k_proj.weight"], w[L + "self_attn.k_proj.bias"])
    v = linear(x, w[L + "self_attn.v_proj.weight"], w[L + "self_attn.v_proj.bias"])
    q = q.view(F_, S, nh, hd).transpose(1, 2)
    k = k.view(F_, S, nh, hd).transpose(1, 2)
    v = v.view(F_, S, nh, hd).transpose(1, 2)
    s = (q @ k.transpose(2, 3)) * (hd ** -0.5)
    s = s - s.amax(-1, keepdim=True)
    p = torch.exp(s)
    p = p / p.sum(-1, keepdim=True)
    o = (p @ v).transpose(1, 2).reshape(F_, S, Hv)
    return linear(o, w[L + "self_attn.out_proj.weight"], w[L + "self_attn.out_proj.bias"])


def siglip_layer(w: W, cfg, i: int, x: torch.Tensor) -> torch.Tensor:
    """siglip_encoder.py:269-305 (pre-LN residual block)."""
    L = f"{VT}encoder.layers.{i}."
    h = layer_norm(x, w[L + "layer_norm1.weight"], w[L + "layer_norm1.bias"], cfg.v_eps)
    x = x + siglip_attention(w, cfg, L, h)
    h = layer_norm(x, w[L + "layer_norm2.weight"], w[L + "layer_norm2.bias"], cfg.v_eps)
    h = linear(h, w[L + "mlp.fc1.weight"], w[L + "mlp.fc1.bias"])
    h = gelu_tanh(h)
    h = linear(h, w[L + "mlp.fc2.weight"], w[L + "mlp.fc2.bias"])
    return x + h


def vision_tower(w: W, cfg, pixels: torch.Tensor) -> torch.Tensor:
    """SigLipVisionTower.forward (siglip_encoder.py:576-589): hidden_states[-1] of the
    26-layer encoder = output of the last run layer, before post_layernorm; head = Identity."""
    x = siglip_embeddings(w, cfg, pixels)
    for i in range(cfg.v_layers):
        x = siglip_layer(w, cfg, i, x)
    assert x.shape[-2] == cfg.v_tokens
    return x


def mm_projector(w: W, x: torch.Tensor) -> torch.Tensor:
    """mlp2x_gelu (multimodal_projector/builder.py:41-48)."""
    h = linear(x, w["model.mm_projector.0.weight"], w["model.mm_projector.0.bias"])
    h = gelu_erf(h)
    return linear(h, w["model.mm_projector.2.weight"], w["model.mm_projector.2.bias"])


def bilinear_taps(n_in: int, n_out: int):
    """Source indices/weights of F.interpolate(mode='bilinear', align_corners=False):
    src = max((dst + 0.5) * n_in/n_out - 0.5, 0); i0 = floor(src); i1 = min(i0+1, n_in-1)."""
    scale = n_in / n_out
    taps = []
    for d in range(n_out):
        src = max((d + 0.5) * scale - 0.5, 0.0)
        i0 = min(int(math.floor(src)), n_in - 1)
        i1 = min(i0 + 1, n_in - 1)
        l1 = np.float32(src - i0)
        taps.append((i0, i1, np.float32(1.0) - l1, l1))
    return taps


def pool_bilinear(cfg, feat: torch.Tensor) -> torch.Tensor:
    """get_2dPool, mode 'bilinear' (stream_video_vln.py:53-73): [F,729,C] -> [F,196,C]."""
    F_, _, C = feat.shape
    side, out = cfg.v_side, cfg.pool_side
    x = feat.view(F_, side, side, C)
    taps = bilinear_taps(side, out)
    rows = []
    for (y0, y1, wy0, wy1) in taps:
        cols = []
        for (x0, x1, wx0, wx1) in taps:
            # ATen upsample_bilinear2d: w0y*(w0x*a + w1x*b) + w1y*(w0x*c + w1x*d)
            top = float(wx0) * x[:, y0, x0] + float(wx1) * x[:, y0, x1]
            bot = float(wx0) * x[:, y1, x0] + float(wx1) * x[:, y1, x1]
            cols.append(float(wy0) * top + float(wy1) * bot)
        rows.append(torch.stack(cols, 1))
    return torch.stack(rows, 1).reshape(F_, out * out, C)


def encode_rgbd(w: W, cfg, images: torch.Tensor, time_ids, num_history: Optional[int]):
    """stream_video_vln.py:102-142 for a batch of one env.
    images [1,V,3,S,S] -> (image_feats [V',196,H], memory_feats [N*196,H] or None).
    depths/poses/intrinsics/task_ids are accepted by the reference but never read."""
    assert images.shape[0] == 1
    V = images.shape[1]
    feats = vision_tower(w, cfg, images[0])                      # [V,729,Hv]
    memory = None
    if V != 1:
        start_idx = time_ids[0][0] if time_ids[0] is not None else 0
        if start_idx != 0:
            nh = num_history
            his = pool_bilinear(cfg, mm_projector(w, feats[:nh]))  # [N,196,H]
            memory = his.flatten(0, 1)
            feats = feats[nh:]
    img = pool_bilinear(cfg, mm_projector(w, feats))
    return img, memory


def prune_memory_tokens(mem: torch.Tensor, keep: int):
    """EXTENSION -- NO REFERENCE COUNTERPART (BASELINE configs[3] "32 pruned slow-memory tokens"): parity
    unpinned by the reference, this function IS the definition the HIP kernels (misc.hip: mem_*_kernel) are tested against.
    score_i = cos(mem_i, mean_j mem_j); the `keep` rows with the smallest score survive (ties: lower index), original order.
    mem [N,H] -> (indices [keep] ascending, scores [N])."""
    mu = mem.mean(0)
    score = (mem @ mu) / torch.clamp(mem.norm(dim=1) * mu.norm(), min=1e-20)
    order = sorted(range(mem.shape[0]), key=lambda i: (float(score[i]), i))
    idx = torch.tensor(sorted(order[:keep]), dtype=torch.long)
    return idx, score


def splice_embeds(w: W, input_ids: Sequence[int], image_feats, memory_feats) -> torch.Tensor:
    """prepare_inputs_labels_for_multimodal (stream_video_vln.py:182-238), one sample:
    text spans are embedded, each -200 is replaced by the next frame's 196 rows and each
    -300 by the memory block, in order of appearance."""
    emb = w["model.embed_tokens.weight"]
    out, img_id, mem_id = [], 0, 0
    for t in input_ids:
        t = int(t)
        if t == IMAGE_TOKEN_INDEX:
            out.append(image_feats[img_id]); img_id += 1
        elif t == MEMORY_TOKEN_INDEX:
            assert mem_id == 0 and memory_feats is not None
            out.append(memory_feats); mem_id += 1
        else:
            out.append(emb[t][None])
    return torch.cat(out, 0)


# ----------------------------------------------------------------------------- Qwen2
def rope_cos_sin(positions: torch.Tensor, hd: int, theta: float):
    """Qwen2RotaryEmbedding (modeling_qwen2.py:91-117): inv_freq = theta^(-2i/hd), fp32."""
    inv = 1.0 / (theta ** (torch.arange(0, hd, 2, dtype=torch.float32) / hd))
    fr = positions.to(torch.float32)[:, None] * inv[None, :]
    emb = torch.cat((fr, fr), -1)
    return emb.cos(), emb.sin()


def rotate_half(x):
    h = x.shape[-1] // 2
    return torch.cat((-x[..., h:], x[..., :h]), -1)


class KVCache:
    """Contiguous per-layer K/V (what DynamicCache holds)."""

    def __init__(self, layers: int):
        self.k: List[Optional[torch.Tensor]] = [None] * layers
        self.v: List[Optional[torch.Tensor]] = [None] * layers

    def __len__(self):
        return 0 if self.k[0] is None else self.k[0].shape[1]

    def update(self, i, k, v):
        self.k[i] = k if self.k[i] is None else torch.cat((self.k[i], k), 1)
        self.v[i] = v if self.v[i] is None else torch.cat((self.v[i], v), 1)
        return self.k[i], self.v[i]


def qwen2_layer(w: W, cfg, i: int, x: torch.Tensor, pos: torch.Tensor, cache: KVCache, q8: Optional[Fp8Emu] = None,
                phase: str = "prefill") -> torch.Tensor:
    """One decoder layer on T new positions (modeling_qwen2.py:195-235,269-299).
    x [T,H]; pos [T] absolute positions; causal over cache + new (bottom-right aligned).
    q8 / phase: opt-in e4m3 emulation of the seven projections (Fp8Emu; extension, no reference counterpart)."""
    L = f"model.layers.{i}."
    T = x.shape[0]
    nq, nkv, hd = cfg.q_heads, cfg.kv_heads, cfg.head_dim
    if q8 is None:
        lin = lambda name, t, b=None: linear(t, w[L + name + ".weight"], b)
        st = lambda t: t
    else:
        lin = lambda name, t, b=None: q8.lin(w, L + name + ".weight", t, b, phase)
        st = q8.store                                # identity unless the emulation also restates the engine's storage rounding
    h = st(rms_norm(x, w[L + "input_layernorm.weight"], cfg.rms_eps))
    q = st(lin("self_attn.q_proj", h, w[L + "self_attn.q_proj.bias"])).view(T, nq, hd)
    k = st(lin("self_attn.k_proj", h, w[L + "self_attn.k_proj.bias"])).view(T, nkv, hd)
    v = st(lin("self_attn.v_proj", h, w[L + "self_attn.v_proj.bias"])).view(T, nkv, hd)
    cos, sin = rope_cos_sin(pos, hd, cfg.rope_theta)
    q = st(q * cos[:, None] + rotate_half(q) * sin[:, None])
    k = st(k * cos[:, None] + rotate_half(k) * sin[:, None])
    kc, vc = cache.update(i, k.transpose(0, 1), v.transpose(0, 1))     # [nkv, len, hd]
    S = kc.shape[1]
    g = nq // nkv
    qh = q.transpose(0, 1).reshape(nkv, g, T, hd)
    s = torch.einsum("kgtd,ksd->kgts", qh, kc) * (hd ** -0.5)
    key_pos = torch.arange(S)[None, :]
    mask = key_pos > pos[:, None]                                       # key visible iff key_pos <= query pos
    s = s.masked_fill(mask[None, None], float("-inf"))
    s = s - s.amax(-1, keepdim=True)
    p = torch.exp(s)
    p = p / p.sum(-1, keepdim=True)
    o = st(torch.einsum("kgts,ksd->kgtd", p, vc).reshape(nq, T, hd).transpose(0, 1).reshape(T, nq * hd))
    x = st(x + lin("self_attn.o_proj", o))
    h = st(rms_norm(x, w[L + "post_attention_layernorm.weight"], cfg.rms_eps))
    gate = lin("mlp.gate_proj", h)
    up = lin("mlp.up_proj", h)
    return st(x + lin("mlp.down_proj", st(silu(gate) * up)))


def qwen2_forward(w: W, cfg, x: torch.Tensor, start: int, cache: KVCache, q8: Optional[Fp8Emu] = None, phase: str = "prefill",
                  layer_taps: Optional[list] = None):
    """Run T new positions [start, start+T) through all layers; returns final-norm hidden [T,H].
    layer_taps (test tap): receives the LAST row of the residual stream after every layer."""
    assert len(cache) == start
    pos = torch.arange(start, start + x.shape[0])
    for i in range(cfg.layers):
        x = qwen2_layer(w, cfg, i, x, pos, cache, q8, phase)
        if layer_taps is not None:
            layer_taps.append(x[-1].clone())
    return rms_norm(x, w["model.norm.weight"], cfg.rms_eps)


def lm_logits(w: W, h_last: torch.Tensor, q8: Optional[Fp8Emu] = None) -> torch.Tensor:
    """q8 with .decode: the engine's lm_head GEMV streams the e4m3 copy for the prefill's token too (svln_set_fp8_decode)"""
    if q8 is not None and q8.decode:
        return h_last @ q8.weight(w, "lm_head.weight").t()
    return h_last @ w["lm_head.weight"].t()


def repetition_penalty(logits: torch.Tensor, generated: Sequence[int], penalty: float) -> torch.Tensor:
    """transformers RepetitionPenaltyLogitsProcessor.__call__ (third-party; pinned 4.45.1 in requirements.txt:140, same formula in the
    container's copy): score = gather(scores, input_ids); score = where(score < 0, score * penalty, score / penalty); scatter back.
    `generated` = the ids produced so far in this turn (the prompt enters as inputs_embeds and has no ids)."""
    if penalty == 1.0 or not generated:
        return logits
    out = logits.clone()
    idx = torch.tensor(sorted(set(int(t) for t in generated)), dtype=torch.long)
    sc = out[idx]
    out[idx] = torch.where(sc < 0, sc * penalty, sc / penalty)
    return out


def greedy_pick(logits: torch.Tensor):
    """argmax with lowest-index tie break (torch.argmax on CPU) + top-2 margin."""
    top2 = torch.topk(logits, 2)
    tok = int(torch.argmax(logits))
    return tok, float(top2.values[0] - top2.values[1])


# ----------------------------------------------------------------------------- session
class GenerateOutput:
    def __init__(self, sequences, past_key_values, hidden, margins):
        self.sequences = sequences            # int64 [1, n_new]  (new tokens only, EOS included)
        self.past_key_values = past_key_values
        self.hidden = hidden                  # fp32 [n_new, H]: final-norm hidden that produced each token
        self.margins = margins                # top-2 logit margin per generated token
        self.cache_len = len(past_key_values)  # KV length right after this turn


class OracleStreamVLN:
    """Same call surface as the reference's StreamVLNForCausalLM for the streaming path
    (reset / reset_for_env / generate), computing everything in fp32 on the CPU."""

    def __init__(self, cfg, weights: Dict[str, np.ndarray], num_history: Optional[int] = None, memory_keep: int = 0,
                 fp8: Optional[Fp8Emu] = None):
        from types import SimpleNamespace
        self.cfg = cfg
        self.fp8 = fp8                          # opt-in e4m3 emulation (extension, no reference counterpart); None = the reference's arithmetic
        self.layer_taps: List[torch.Tensor] = []   # last-row residual stream after every layer of the latest prefill (test tap)
        # test mode (no reference counterpart): a queue of token lists, one per coming generate() call -- the call decodes exactly those tokens
        # (teacher forcing) while `own_picks` keeps what this model would have picked at each step and `margins` its top-1 / top-2 gap
        self.teacher_tokens: List[List[int]] = []
        self.own_picks: List[int] = []
        # the two call-time knobs of the reference, under the reference's attribute names
        self.config = SimpleNamespace(tokenizer_model_max_length=None)          # stream_video_vln.py:241
        self.generation_config = SimpleNamespace(repetition_penalty=1.0)        # GenerationConfig default
        self.w = {k: torch.from_numpy(np.ascontiguousarray(v, dtype=np.float32)) for k, v in weights.items()}
        self.num_history = num_history
        self.memory_keep = memory_keep          # > 0: opt-in extension prune_memory_tokens (no reference counterpart)
        self.reset(1)

    def reset(self, env_num: int):                       # stream_video_vln.py:473-475
        self.curr_t = [0] * env_num
        self.cache = [dict() for _ in range(env_num)]    # (reference aliases one dict; per-env here)

    def reset_for_env(self, env_idx: int):               # stream_video_vln.py:477-479
        self.curr_t[env_idx] = 0
        self.cache[env_idx] = dict()

    @torch.no_grad()
    def generate(self, inputs, images, env_id=0, time_ids=None, past_key_values: Optional[KVCache] = None,
                 max_new_tokens: int = 10000, eos_token_ids: Sequence[int] = (), **_ignored) -> GenerateOutput:
        cfg, w = self.cfg, self.w
        ids = [int(t) for t in np.asarray(inputs).reshape(-1)]
        if len(ids) == 1:
            raise NotImplementedError("single-token `inputs` bypasses the multimodal path in the reference")
        images = torch.as_tensor(np.asarray(images), dtype=torch.float32)
        img, mem = encode_rgbd(w, cfg, images, time_ids, self.num_history)
        if mem is not None and 0 < self.memory_keep < mem.shape[0]:
            mem = mem[prune_memory_tokens(mem, self.memory_keep)[0]]
        new = splice_embeds(w, ids, img, mem)
        tml = getattr(self.config, "tokenizer_model_max_length", None)
        if tml is not None:                                  # stream_video_vln.py:241-244: new_input_embeds[:tokenizer_model_max_length]
            new = new[:tml]
        st = self.cache[env_id]
        st["inputs_embeds"] = new if self.curr_t[env_id] == 0 else torch.cat((st["inputs_embeds"], new), 0)
        self.curr_t[env_id] += 1
        E = st["inputs_embeds"]
        cache = past_key_values if past_key_values is not None else KVCache(cfg.layers)
        P, L_total = len(cache), E.shape[0]
        assert L_total > P
        eos = set(int(e) for e in eos_token_ids)
        self.layer_taps = []
        h = qwen2_forward(w, cfg, E[P:], P, cache, self.fp8, "prefill", self.layer_taps)[-1]
        out, hid, margins = [], [], []
        pen = float(getattr(self.generation_config, "repetition_penalty", 1.0) or 1.0)
        forced = self.teacher_tokens.pop(0) if self.teacher_tokens else None
        self.own_picks = []
        while True:
            tok, margin = greedy_pick(repetition_penalty(lm_logits(w, h, self.fp8), out, pen))
            self.own_picks.append(tok)
            if forced is not None:                           # test mode: continue from the given token, keep this model's own pick and margin
                tok = int(forced[len(out)])
            out.append(tok); hid.append(h.clone()); margins.append(margin)
            if (forced is not None and len(out) >= len(forced)) or (forced is None and (tok in eos or len(out) >= max_new_tokens)):
                break
            x = w["model.embed_tokens.weight"][tok][None]
            h = qwen2_forward(w, cfg, x, len(cache), cache, self.fp8, "decode")[-1]
        assert len(cache) == L_total + len(out) - 1
        go = GenerateOutput(torch.tensor([out], dtype=torch.long), cache, torch.stack(hid), margins)
        go.own_picks = list(self.own_picks)                  # == the sequence unless teacher_tokens steered this turn
        return go
