"""Host-side mirror of the reference's operator surface for the streaming path.

`StreamVLNForCausalLM` here keeps the names, argument meaning and error behaviour of
streamvln/model/stream_video_vln.py (`generate`, `reset`, `reset_for_env`, `get_vision_tower`,
`get_model`, `.model.num_history`) so `VLNEvaluator.step` / the Habitat eval loop
(streamvln/streamvln_agent.py:169-258, streamvln/streamvln_eval.py:290-350) call it unchanged.
All arithmetic happens in the HIP engine behind include/streamvln_hip.h; PyTorch is used only
for the caller's tensors (device memory in / token ids out).
"""
from __future__ import annotations

import ctypes as C
from types import SimpleNamespace
from typing import Dict, Iterable, Optional, Sequence

import numpy as np
import torch

from . import _lib
from .config import CONFIGS, StreamVLNConfig
from .weights import tensor_seed, tensor_specs


def _check(rc):
    _lib.check(rc)


class SigLipImageProcessor:
    """llava/model/multimodal_encoder/siglip_encoder.py:34-67: bicubic resize to 384x384 (aspect not preserved), x/255,
    (x-0.5)/0.5, channels first.

    Two backends with bit-identical results (tests/test_preprocess.py):
      * "hip"  -- the processor a model hands out (`model.get_vision_tower().image_processor`): upload of the uint8 frame +
                  one HIP kernel reproducing Pillow's fixed-point bicubic (csrc/preprocess.hip); returns a CUDA tensor, so the
                  harness's `dict_to_cuda` is a no-op.  No fallback: it raises without the engine / a GPU.
      * "pil"  -- stand-alone `SigLipImageProcessor()`: Pillow on the host, exactly what the reference runs (host utility
                  for callers without an engine, and the comparison leg of the tests)."""

    def __init__(self, size=(384, 384), engine=None, device_index: int = 0):
        self.image_mean, self.image_std = (0.5, 0.5, 0.5), (0.5, 0.5, 0.5)
        self.size = size
        self.rescale_factor = 1 / 255
        self.crop_size = {"height": size[0], "width": size[1]}
        self._engine = engine                 # (lib, handle) of the owning model
        self.device_index = device_index
        self.backend = "hip" if engine is not None else "pil"
        self._ext_stream = None               # the engine's HIP stream as a torch stream (lazily wrapped)

    # -- host path (Pillow) ----------------------------------------------------------------------------
    def _pil(self, rgb) -> torch.Tensor:
        from PIL import Image
        img = rgb if isinstance(rgb, Image.Image) else Image.fromarray(np.asarray(rgb, dtype=np.uint8))
        img = img.convert("RGB").resize((self.size[1], self.size[0]), resample=Image.BICUBIC)
        a = (np.asarray(img).astype(np.float64) * self.rescale_factor).astype(np.float32)
        a = (a - np.float32(0.5)) / np.float32(0.5)
        return torch.from_numpy(np.ascontiguousarray(a.transpose(2, 0, 1)))

    # -- GPU path --------------------------------------------------------------------------------------
    @staticmethod
    def _as_u8(rgb) -> np.ndarray:
        if not isinstance(rgb, np.ndarray):                       # PIL image (the reference callers pass Image.fromarray(rgb))
            rgb = np.asarray(rgb.convert("RGB"))
        a = np.ascontiguousarray(rgb, dtype=np.uint8)
        if a.ndim != 3 or a.shape[2] != 3:
            raise ValueError(f"expected an RGB frame [H, W, 3], got shape {a.shape}")
        return a

    def _hip(self, frames) -> torch.Tensor:
        """frames: list of uint8 [H,W,3] arrays of one geometry -> CUDA fp32 [n,3,S,S]"""
        if self._engine is None:
            raise RuntimeError("the model that owns this image processor was closed")
        lib, h = self._engine
        n, (H, W, _) = len(frames), frames[0].shape
        buf = frames[0] if n == 1 else np.ascontiguousarray(np.stack(frames))
        dev = torch.device("cuda", self.device_index)
        out = torch.empty((n, 3, self.size[0], self.size[1]), dtype=torch.float32, device=dev)
        # the engine writes `out` on its own stream: order the two streams on the device instead of stalling the host twice per frame
        # (the call returns once the frame bytes are staged; svln_encode_frames on the same engine is ordered behind the kernel)
        ext, cur = self._engine_stream(dev), torch.cuda.current_stream(dev)
        same = cur.cuda_stream == ext.cuda_stream          # the caller runs torch on the engine's stream (model.torch_stream): already ordered
        if not same:
            ext.wait_stream(cur)
        _check(lib.svln_preprocess_frames_enqueue(h, buf.ctypes.data, n, H, W, 0, out.data_ptr()))
        if not same:
            cur.wait_stream(ext)  # torch work on `out` (and any later reuse of its memory, which stays on this stream) follows the kernel
        return out

    def _engine_stream(self, dev):
        if self._ext_stream is None:
            lib, h = self._engine
            sp = C.c_void_p()
            _check(lib.svln_engine_stream(h, C.byref(sp)))
            self._ext_stream = torch.cuda.ExternalStream(sp.value, device=dev)
        return self._ext_stream

    def preprocess_array(self, rgb) -> torch.Tensor:
        if self.backend == "pil":
            return self._pil(rgb)
        return self._hip([self._as_u8(rgb)])[0]

    def preprocess(self, images, return_tensors="pt"):
        from PIL import Image
        if isinstance(images, Image.Image) or (isinstance(images, np.ndarray) and images.ndim == 3):
            images = [images]
        if self.backend == "pil":
            vals = [self._pil(im) for im in images]
        else:
            frames = [self._as_u8(im) for im in images]
            if len({f.shape for f in frames}) == 1:
                vals = list(self._hip(frames))                    # one upload + one launch for the batch
            else:
                vals = [self._hip([f])[0] for f in frames]
        if return_tensors == "pt":
            return {"pixel_values": torch.stack(vals)}
        return {"pixel_values": [v.cpu().numpy() for v in vals]}


class _VisionTower:
    def __init__(self, cfg: StreamVLNConfig, engine=None, device_index: int = 0):
        self.config = SimpleNamespace(hidden_size=cfg.v_hidden, image_size=cfg.v_image, patch_size=cfg.v_patch)
        self.image_processor = SigLipImageProcessor((cfg.v_image, cfg.v_image), engine=engine, device_index=device_index)
        self.is_loaded = True

    @property
    def num_patches_per_side(self):
        return self.config.image_size // self.config.patch_size

    @property
    def num_patches(self):
        return self.num_patches_per_side ** 2

    @property
    def hidden_size(self):
        return self.config.hidden_size


class KVHandle:
    """Opaque `past_key_values` handle: the KV pages live in the engine, keyed by env."""

    def __init__(self, env_id: int, epoch: int, length: int):
        self.env_id, self.epoch, self.length = env_id, epoch, length

    def get_seq_length(self):
        return self.length

    def __len__(self):
        return self.length


class GenerateOutput(dict):
    """`return_dict_in_generate=True` result: `.sequences` (new tokens only, EOS included) and
    `.past_key_values` (streamvln_eval.py:334-335)."""

    def __getattr__(self, k):
        try:
            return self[k]
        except KeyError as e:
            raise AttributeError(k) from e


class StreamVLNForCausalLM:
    def __init__(self, config: StreamVLNConfig, dtype: torch.dtype = torch.bfloat16, device: int = 0, max_envs: int = 1,
                 max_frames: int = 9, max_positions: Optional[int] = None):
        lib = _lib.load()
        self.cfg = config
        self.dtype = dtype
        self.device_index = device
        c = _lib.SvlnConfig(
            v_hidden=config.v_hidden, v_inter=config.v_inter, v_heads=config.v_heads, v_layers=config.v_layers,
            v_patch=config.v_patch, v_image=config.v_image, v_eps=config.v_eps,
            hidden=config.hidden, layers=config.layers, q_heads=config.q_heads, kv_heads=config.kv_heads,
            head_dim=config.head_dim, inter=config.inter, vocab=config.vocab, rope_theta=config.rope_theta,
            rms_eps=config.rms_eps, max_positions=max_positions or config.max_positions, max_envs=max_envs,
            max_frames=max_frames, dtype=_lib.SVLN_F32 if dtype == torch.float32 else _lib.SVLN_BF16)
        h = C.c_void_p()
        _check(lib.svln_create(C.byref(c), device, C.byref(h)))
        self._lib, self._h = lib, h
        self.max_envs, self.max_frames = max_envs, max_frames
        self._tower = _VisionTower(config, engine=(lib, h), device_index=device)
        # `.model` = StreamVLNModel in the reference; callers set `.model.num_history` (streamvln_eval.py:531)
        self.model = SimpleNamespace(num_history=None, get_vision_tower=lambda: self._tower)
        # tokenizer_model_max_length: read at every call like the reference does (stream_video_vln.py:241-244); None = no truncation
        self.config = SimpleNamespace(mm_spatial_pool_mode="bilinear", hidden_size=config.hidden, vocab_size=config.vocab,
                                      tokenizer_model_max_length=None)
        # Qwen2-7B-Instruct <|im_end|>, <|endoftext|>; repetition_penalty as transformers' GenerationConfig defaults it
        self.generation_config = SimpleNamespace(eos_token_id=[151645, 151643], repetition_penalty=1.0)
        self._row_limit, self._rep_penalty = 0, 1.0           # what the engine currently holds
        self._tickets: Dict[int, tuple] = {}
        self.reset(max_envs)

    # ---- construction -------------------------------------------------------------------------
    @classmethod
    def from_config(cls, config, seed: Optional[int] = None, **kw):
        if isinstance(config, str):
            config = CONFIGS[config]
        m = cls(config, **kw)
        if seed is not None:
            m.load_synthetic(seed)
        return m

    @classmethod
    def from_pretrained(cls, path, config=None, torch_dtype=torch.bfloat16, attn_implementation=None, low_cpu_mem_usage=False,
                        device: int = 0, **kw):
        """Checkpoint ingestion (HF safetensors layout).  No checkpoint exists offline."""
        import glob
        import os
        files = sorted(glob.glob(os.path.join(str(path), "*.safetensors")))
        if not files:
            raise FileNotFoundError(f"no *.safetensors under {path}")
        from safetensors import safe_open
        from .config import TRUE, config_from_hf
        import json
        # dims: an explicit StreamVLNConfig, else the harness's HF config object, else the checkpoint's config.json, else Qwen2-7B
        if isinstance(config, StreamVLNConfig):
            cfg = config
        elif config is not None:
            cfg = config_from_hf(config)
        elif os.path.exists(os.path.join(str(path), "config.json")):
            cfg = config_from_hf(json.load(open(os.path.join(str(path), "config.json"))))
        else:
            cfg = TRUE
        m = cls(cfg, dtype=torch_dtype, device=device, **kw)
        hf = config if config is not None and not isinstance(config, StreamVLNConfig) else None
        if hf is None and os.path.exists(os.path.join(str(path), "config.json")):
            hf = json.load(open(os.path.join(str(path), "config.json")))
        if hf is not None:
            tml = hf.get("tokenizer_model_max_length") if isinstance(hf, dict) else getattr(hf, "tokenizer_model_max_length", None)
            m.config.tokenizer_model_max_length = None if tml is None else int(tml)
        gen = os.path.join(str(path), "generation_config.json")
        if os.path.exists(gen):                                   # the checkpoint's generation defaults
            m.apply_generation_config(json.load(open(gen)))
        for f in files:
            with safe_open(f, framework="pt") as sf:
                for name in sf.keys():
                    if name in m.tensor_names():
                        m.set_tensor(name, sf.get_tensor(name))
        m.check_weights()
        return m

    #: generation_config.json keys that would change what `generate(do_sample=False, num_beams=1, max_new_tokens=...)` returns in
    #: transformers 4.45.1 and that the engine does not implement -> an error at load time, never a silently different answer.
    #: (value = the neutral setting.)  Sampling knobs (temperature / top_k / top_p / typical_p ...) are inert under the harness's
    #: explicit do_sample=False and are ignored, as HF itself does (with a warning).
    _UNSUPPORTED_GENERATION_KEYS = {
        "no_repeat_ngram_size": 0, "encoder_repetition_penalty": 1.0, "encoder_no_repeat_ngram_size": 0, "bad_words_ids": None,
        "min_length": 0, "min_new_tokens": None, "forced_bos_token_id": None, "forced_eos_token_id": None, "suppress_tokens": None,
        "begin_suppress_tokens": None, "sequence_bias": None, "penalty_alpha": None, "guidance_scale": None, "num_beam_groups": 1,
        "diversity_penalty": 0.0, "length_penalty": 1.0, "exponential_decay_length_penalty": None, "renormalize_logits": False,
        "remove_invalid_values": False, "forced_decoder_ids": None, "constraints": None, "force_words_ids": None,
        "prompt_lookup_num_tokens": None, "assistant_model": None, "dola_layers": None, "watermarking_config": None,
    }

    def apply_generation_config(self, gen: dict):
        """A checkpoint's generation_config.json -> the stop ids and the repetition penalty of the greedy loop; any other key that
        changes greedy decoding raises."""
        eos = gen.get("eos_token_id")
        if eos is not None:
            self.generation_config.eos_token_id = list(eos) if isinstance(eos, (list, tuple)) else [int(eos)]
        rp = gen.get("repetition_penalty")
        if rp is not None:
            if not float(rp) > 0:
                raise ValueError(f"generation_config.repetition_penalty must be > 0, got {rp}")
            self.generation_config.repetition_penalty = float(rp)
        bad = []
        for k, neutral in self._UNSUPPORTED_GENERATION_KEYS.items():
            v = gen.get(k, neutral)
            if v is None or v == neutral or (isinstance(v, (list, dict)) and not v):
                continue
            if k == "length_penalty" and int(gen.get("num_beams", 1)) == 1:
                continue                                          # beam-search only
            bad.append(f"{k}={v!r}")
        if bad:
            raise NotImplementedError("generation_config.json changes greedy decoding in ways the HIP path does not implement: "
                                      + ", ".join(bad))

    def _sync_call_config(self):
        """config.tokenizer_model_max_length and generation_config.repetition_penalty are plain attributes the caller may change
        between calls (the reference reads them at call time): push them to the engine when they differ from what it holds.
        The row limit has no in-flight restriction and is pushed at every call; the penalty cannot change while scheduler turns are
        in flight (the engine refuses it), so a changed value raises here instead of being ignored."""
        tml = getattr(self.config, "tokenizer_model_max_length", None)
        if tml is not None and int(tml) < 1:
            # the reference would splice `new_input_embeds[:0]` -- an empty turn -- and fail further down (stream_video_vln.py:241-244);
            # the engine's 0 means "no truncation", so 0 must never reach it
            raise ValueError(f"config.tokenizer_model_max_length must be None (no truncation) or >= 1, got {tml!r}")
        lim = 0 if tml is None else int(tml)
        if lim != self._row_limit:
            _check(self._lib.svln_set_turn_row_limit(self._h, lim))
            self._row_limit = lim
        rp = float(getattr(self.generation_config, "repetition_penalty", 1.0) or 1.0)
        if rp != self._rep_penalty:
            if self._tickets:
                raise RuntimeError("generation_config.repetition_penalty changed while turns are in flight: collect or cancel them first")
            _check(self._lib.svln_set_repetition_penalty(self._h, rp))
            self._rep_penalty = rp

    def tensor_names(self):
        return {s.name for s in tensor_specs(self.cfg)}

    def load_synthetic(self, seed: int):
        """Seeded random-init weights generated on the device (streamvln_amd/weights.py)."""
        for s in tensor_specs(self.cfg):
            _check(self._lib.svln_synth_tensor(self._h, s.name.encode(), tensor_seed(seed, s.name), s.half_width, s.base))
        self.check_weights()
        return self

    def set_tensor(self, name: str, value):
        if isinstance(value, torch.Tensor):
            t = value.detach()
            if t.dtype not in (torch.float32, torch.bfloat16):
                t = t.float()
            t = t.contiguous()
            dt = _lib.SVLN_F32 if t.dtype == torch.float32 else _lib.SVLN_BF16
            if t.is_cuda:
                torch.cuda.synchronize()
            _check(self._lib.svln_set_tensor(self._h, name.encode(), C.c_void_p(t.data_ptr()), dt, t.numel(), int(t.is_cuda)))
        else:
            a = np.ascontiguousarray(value, dtype=np.float32)
            _check(self._lib.svln_set_tensor(self._h, name.encode(), a.ctypes.data_as(C.c_void_p), _lib.SVLN_F32, a.size, 0))

    def load_state_dict(self, sd: Dict[str, object]):
        for k, v in sd.items():
            self.set_tensor(k, v)
        self.check_weights()
        return self

    def check_weights(self):
        _check(self._lib.svln_weights_ready(self._h))

    def get_tensor(self, name: str) -> np.ndarray:
        spec = {s.name: s for s in tensor_specs(self.cfg)}[name]
        out = np.empty(spec.numel, dtype=np.float32)
        _check(self._lib.svln_get_tensor_f32(self._h, name.encode(), out.ctypes.data_as(C.POINTER(C.c_float)), out.size))
        return out.reshape(spec.shape)

    # ---- nn.Module-flavoured no-ops the harness calls (streamvln_eval.py:531-539) -----------------
    def requires_grad_(self, flag=False):
        return self

    def to(self, *a, **k):
        return self

    def eval(self):
        return self

    def get_model(self):
        return self.model

    def get_vision_tower(self):
        return self._tower

    def _order_engine_after(self, pix):
        """The engine's stream is about to read `pix`: order it behind the torch work that produced the tensor (on the device, no host
        stall).  Nothing to do when the caller runs torch on the engine's stream (`torch_stream`)."""
        ext, cur = self._tower.image_processor._engine_stream(pix.device), torch.cuda.current_stream(pix.device)
        if cur.cuda_stream != ext.cuda_stream:
            ext.wait_stream(cur)

    def _order_torch_after(self, pix):
        """... and the caller's stream behind that read, so that freeing `pix` cannot hand its memory to later torch work too early."""
        ext, cur = self._tower.image_processor._engine_stream(pix.device), torch.cuda.current_stream(pix.device)
        if cur.cuda_stream != ext.cuda_stream:
            cur.wait_stream(ext)

    @property
    def torch_stream(self):
        """The engine's HIP stream as a torch stream.  A harness that wraps its loop in `with torch.cuda.stream(model.torch_stream):`
        runs its own tensor ops (stack / to / cat) on the stream the engine uses, so no cross-stream ordering (two event records and
        two stream waits per frame and per turn) is needed; any other stream works too, at that cost."""
        return self._tower.image_processor._engine_stream(self.device)

    def frame_ring(self, slots: int, height: int, width: int) -> np.ndarray:
        """Engine-owned pinned frame ring (svln_frame_ring) as a uint8 array [slots, height, width, 3].  Frames written into a slot and
        handed to the image processor as that slot's view (`ring[k]`) are read by the GPU in place: no host-side staging copy.  Rewrite
        a slot only after `frame_ring_wait(k)` (or any call that synchronises, e.g. `generate`'s return for frames preprocessed before)."""
        base, stride = C.c_void_p(), C.c_int64()
        _check(self._lib.svln_frame_ring(self._h, int(slots), int(height), int(width), C.byref(base), C.byref(stride)))
        raw = (C.c_uint8 * (stride.value * slots)).from_address(base.value)
        arr = np.frombuffer(raw, dtype=np.uint8)
        self._ring = np.lib.stride_tricks.as_strided(arr, shape=(slots, height, width, 3), strides=(stride.value, width * 3, 3, 1))
        return self._ring

    def frame_ring_wait(self, slot: int):
        _check(self._lib.svln_frame_ring_wait(self._h, int(slot)))

    @property
    def device(self):
        return torch.device("cuda", self.device_index)

    # ---- session state (stream_video_vln.py:473-479) ----------------------------------------------
    def reset(self, env_num: int):
        """`model.reset(world_size)` in the reference harness (streamvln_eval.py:542): sizes the per-env host state.  The
        reference indexes that state with `env_id = rank` while each process only ever drives ONE env, so `env_num` may
        exceed the engine's `max_envs`: an env_id is bound to one of the engine's slots on first use, and at most
        `max_envs` env_ids can be live at a time."""
        if env_num < 1:
            raise ValueError("env_num must be >= 1")
        self.curr_t = [0] * env_num
        self._epoch = [0] * env_num
        self._slots: Dict[int, int] = {}
        _check(self._lib.svln_batch_cancel(self._h, -1))           # turns still in the scheduler belong to the old state
        self._tickets.clear()
        for i in range(self.max_envs):
            _check(self._lib.svln_reset_env(self._h, i))

    def _slot(self, env_id: int) -> int:
        s = self._slots.get(env_id)
        if s is None:
            used = set(self._slots.values())
            free = [i for i in range(self.max_envs) if i not in used]
            if not free:
                raise ValueError(f"env_id {env_id}: all {self.max_envs} engine slots are bound to other envs "
                                 f"({sorted(self._slots)}); build the model with a larger max_envs")
            s = self._slots[env_id] = free[0]
        return s

    def reset_for_env(self, env_idx: int):
        if not (0 <= env_idx < len(self.curr_t)):
            raise IndexError(f"env_id {env_idx} out of range")
        self.curr_t[env_idx] = 0
        self._epoch[env_idx] += 1
        if env_idx in self._slots:
            _check(self._lib.svln_reset_env(self._h, self._slots[env_idx]))       # (drops the env's scheduler turn, if any)
        for slot in [k for k, v in self._tickets.items() if v[0] == env_idx]:
            del self._tickets[slot]

    # ---- the call (stream_video_vln.py:353-407) -------------------------------------------------------
    def _parse_call(self, inputs, images, kwargs):
        """argument handling shared by generate / generate_batch; returns (ids, pixels [V,3,S,S] fp32, V, n_memory, env_id, past,
        max_new, eos)"""
        kwargs = dict(kwargs)
        for k in ("position_ids", "attention_mask", "task_type", "image_sizes", "depths", "poses", "intrinsics", "task_ids"):
            kwargs.pop(k, None)
        time_ids = kwargs.pop("time_ids", None)
        if "inputs_embeds" in kwargs:
            raise NotImplementedError("`inputs_embeds` is not supported")
        env_id = kwargs.pop("env_id", None)
        past = kwargs.pop("past_key_values", None)
        max_new = int(kwargs.pop("max_new_tokens", 10000))
        eos = kwargs.pop("eos_token_ids", None)
        if kwargs.get("do_sample", False) or kwargs.get("num_beams", 1) != 1:
            raise NotImplementedError("only greedy decoding (do_sample=False, num_beams=1) is on the path")
        if eos is None:
            eos = self.generation_config.eos_token_id
        eos = [int(e) for e in (eos if isinstance(eos, (list, tuple)) else [eos])]
        if images is None:
            raise NotImplementedError("text-only turns are not part of the streaming path")
        if env_id is None or not (0 <= env_id < len(self.curr_t)):
            raise IndexError(f"env_id {env_id} out of range")
        ids = torch.as_tensor(inputs).reshape(-1).to("cpu", torch.int64)
        if ids.numel() == 1:
            raise NotImplementedError("single-token `inputs` bypasses the multimodal path in the reference "
                                      "(stream_video_vln.py:149)")
        B, V = images.shape[0], images.shape[1]
        if B != 1:
            raise NotImplementedError("one env per request (reference harnesses run batch 1); use generate_batch for several envs")
        n_memory = 0
        if V != 1:                                            # encode_rgbd, stream_video_vln.py:111-130
            start_idx = time_ids[0][0] if (time_ids is not None and time_ids[0] is not None) else 0
            if start_idx != 0:
                if self.model.num_history is None:
                    raise TypeError("model.num_history must be set before a <memory> turn")
                n_memory = int(self.model.num_history)
        pix = images[0].to(torch.float32).contiguous()
        return ids, pix, V, n_memory, env_id, past, max_new, eos

    def _begin_turn(self, env_id, past):
        """KV handle / per-env embeds bookkeeping of StreamVLNForCausalLM.generate (stream_video_vln.py:396-401)"""
        if past is None:
            _check(self._lib.svln_kv_reset(self._h, self._slot(env_id)))
        elif not isinstance(past, KVHandle) or past.env_id != env_id or past.epoch != self._epoch[env_id]:
            raise ValueError("past_key_values does not belong to this env's current window")
        if self.curr_t[env_id] == 0:
            ne, kl = C.c_int32(), C.c_int32()
            _check(self._lib.svln_env_state(self._h, self._slot(env_id), C.byref(ne), C.byref(kl)))
            if ne.value != 0:
                _check(self._lib.svln_reset_env(self._h, self._slot(env_id)))
        self.curr_t[env_id] += 1

    def _result(self, env_id, tokens, inputs):
        ne, kl = C.c_int32(), C.c_int32()
        _check(self._lib.svln_env_state(self._h, self._slot(env_id), C.byref(ne), C.byref(kl)))
        dev = inputs.device if isinstance(inputs, torch.Tensor) else "cpu"
        seq = torch.from_numpy(np.asarray(tokens, dtype=np.int64).copy()).unsqueeze(0).to(dev)
        return GenerateOutput(sequences=seq, past_key_values=KVHandle(env_id, self._epoch[env_id], kl.value))

    @torch.no_grad()
    def generate(self, inputs=None, images=None, image_sizes=None, depths=None, poses=None, intrinsics=None, task_ids=None,
                 **kwargs):
        """One model turn = ONE crossing into the engine (svln_turn: encode_rgbd, the KV / embeds bookkeeping of the reference's generate,
        splice, prefill + greedy decode)."""
        ids, pix, V, n_memory, env_id, past, max_new, eos = self._parse_call(inputs, images, kwargs)
        self._sync_call_config()
        # the reference's bookkeeping that needs no engine call: the KV handle must be this env's current one; curr_t counts the turns
        if past is not None and (not isinstance(past, KVHandle) or past.env_id != env_id or past.epoch != self._epoch[env_id]):
            raise ValueError("past_key_values does not belong to this env's current window")
        slot = self._slot(env_id)
        on_dev = int(pix.is_cuda)
        ids_np = ids.numpy()
        cap = min(max_new, self.cfg.max_positions)
        buf = self._turn_buffers(cap, eos)
        a = buf["args"]
        a.pixels, a.n_frames, a.pixels_on_device, a.env = pix.data_ptr(), V, on_dev, slot
        a.ids, a.n_ids, a.n_memory = ids_np.ctypes.data, ids_np.size, n_memory
        a.new_window, a.new_episode, a.max_new_tokens = int(past is None), int(self.curr_t[env_id] == 0), max_new
        if on_dev:
            self._order_engine_after(pix)
        rc = self._lib.svln_turn(self._h, buf["args_ref"], buf["out_p"], cap, buf["n_out_ref"], buf["kv_ref"])
        if on_dev:
            self._order_torch_after(pix)
        _check(rc)
        self.curr_t[env_id] += 1
        dev = inputs.device if isinstance(inputs, torch.Tensor) else "cpu"
        seq = torch.from_numpy(buf["out"][: buf["n_out"].value].copy()).unsqueeze(0)
        if dev != "cpu" and str(dev) != "cpu":
            seq = seq.to(dev)
        return GenerateOutput(sequences=seq, past_key_values=KVHandle(env_id, self._epoch[env_id], buf["kv"].value))

    def _turn_buffers(self, cap, eos):
        """ctypes scratch of generate(), built once per (capacity, eos list): no per-turn allocation or pointer casting"""
        key = (cap, tuple(eos))
        buf = getattr(self, "_tbuf", None)
        if buf is None or buf["key"] != key:
            out = np.zeros(cap, dtype=np.int64)
            eos_np = np.asarray(eos, dtype=np.int64)
            args = _lib.SvlnTurnArgs()
            args.eos_ids, args.n_eos = eos_np.ctypes.data, eos_np.size
            n_out, kv = C.c_int32(), C.c_int32()
            buf = self._tbuf = {"key": key, "out": out, "out_p": out.ctypes.data_as(C.POINTER(C.c_int64)), "eos": eos_np, "args": args,
                                "args_ref": C.byref(args), "n_out": n_out, "n_out_ref": C.byref(n_out), "kv": kv, "kv_ref": C.byref(kv)}
        return buf

    @torch.no_grad()
    def generate_batch(self, requests, max_new_tokens: int = 10000, eos_token_ids=None):
        """Several envs' turns executed together (build-side extension, SURVEY.md 8f-1 / BASELINE configs[4]).
        `requests`: list of kwargs dicts as passed to `generate` (distinct env_id, at most 8).  Per-env results are those of
        `generate` called env by env (same protocol, same KV / embeds state); the dense layers and every decode step run
        once for the whole batch.  Returns a list of GenerateOutput in request order."""
        if not (1 <= len(requests) <= 8):
            raise ValueError("1..8 requests per batch")
        parsed = []
        for r in requests:
            r = dict(r)
            r.setdefault("max_new_tokens", max_new_tokens)
            if eos_token_ids is not None:
                r.setdefault("eos_token_ids", eos_token_ids)
            parsed.append((r,) + self._parse_call(r.pop("inputs", None), r.pop("images", None), r))
        if len({p[5] for p in parsed}) != len(parsed):
            raise ValueError("duplicate env_id in batch")
        max_new, eos = parsed[0][7], parsed[0][8]
        if any(p[7] != max_new or p[8] != eos for p in parsed):
            raise ValueError("max_new_tokens / eos_token_ids must be the same for every request of a batch")
        if self._tickets:
            raise RuntimeError("generate_batch needs an idle scheduler: collect or cancel the submitted turns first")
        self._sync_call_config()
        # vision: encode the frames of as many requests as fit the frame buffer per call, then splice per env
        i = 0
        while i < len(parsed):
            j, frames = i, 0
            while j < len(parsed) and (j == i or frames + parsed[j][3] <= self.max_frames):
                frames += parsed[j][3]
                j += 1
            if frames > self.max_frames:
                raise ValueError(f"a request has {frames} frames but the engine was built with max_frames={self.max_frames}")
            pix = torch.cat([p[2] for p in parsed[i:j]], 0).contiguous()
            on_dev = int(pix.is_cuda)
            if on_dev:
                self._order_engine_after(pix)
            _check(self._lib.svln_encode_frames(self._h, C.c_void_p(pix.data_ptr()), frames, on_dev))
            if on_dev:
                self._order_torch_after(pix)
            base = 0
            for (_, ids, _, V, n_memory, env_id, past, _, _) in parsed[i:j]:
                self._begin_turn(env_id, past)
                ids_np = np.ascontiguousarray(ids.numpy())
                _check(self._lib.svln_append_turn_at(self._h, self._slot(env_id), ids_np.ctypes.data_as(C.POINTER(C.c_int64)), ids_np.size, base, n_memory))
                base += V
            i = j
        envs = np.asarray([self._slot(p[5]) for p in parsed], dtype=np.int32)
        cap = min(max_new, self.cfg.max_positions)
        out = np.zeros((len(parsed), cap), dtype=np.int64)
        n_out = np.zeros(len(parsed), dtype=np.int32)
        eos_np = np.asarray(eos, dtype=np.int64)
        _check(self._lib.svln_generate_batch(self._h, envs.ctypes.data_as(C.POINTER(C.c_int32)), len(parsed), max_new,
                                             eos_np.ctypes.data_as(C.POINTER(C.c_int64)), eos_np.size,
                                             out.ctypes.data_as(C.POINTER(C.c_int64)), cap, n_out.ctypes.data_as(C.POINTER(C.c_int32))))
        return [self._result(p[5], out[k, : n_out[k]], requests[k].get("inputs")) for k, p in enumerate(parsed)]

    # ---- iteration-level scheduler: envs whose turns fall due at different times (streamvln_dagger.py:232-313) ----
    @torch.no_grad()
    def submit(self, inputs=None, images=None, **kwargs):
        """Start one env's turn without waiting for it: same arguments as `generate`.  The turn joins the next `step_batch`
        iteration, sharing its pass over the weights with whatever the other envs in flight are doing (prefill or decode).
        Returns a ticket; the result arrives from `step_batch`."""
        ids, pix, V, n_memory, env_id, past, max_new, eos = self._parse_call(inputs, images, kwargs)
        # refuse BEFORE the env's state changes (frames encoded, curr_t advanced, rows spliced): a turn the scheduler cannot take must
        # leave the env exactly as it was, so that the caller can retry it later
        if any(v[0] == env_id for v in self._tickets.values()):
            raise RuntimeError(f"env_id {env_id} already has a turn in flight")
        if len(self._tickets) >= 8:
            raise RuntimeError("at most 8 turns in flight: collect finished turns with step_batch first")
        self._sync_call_config()
        on_dev = int(pix.is_cuda)
        if on_dev:
            self._order_engine_after(pix)
        _check(self._lib.svln_encode_frames(self._h, C.c_void_p(pix.data_ptr()), V, on_dev))
        if on_dev:
            self._order_torch_after(pix)
        self._begin_turn(env_id, past)
        ids_np = np.ascontiguousarray(ids.numpy())
        _check(self._lib.svln_append_turn(self._h, self._slot(env_id), ids_np.ctypes.data_as(C.POINTER(C.c_int64)), ids_np.size, n_memory))
        eos_np = np.asarray(eos, dtype=np.int64)
        slot = C.c_int32()
        _check(self._lib.svln_batch_submit(self._h, self._slot(env_id), min(max_new, self.cfg.max_positions),
                                           eos_np.ctypes.data_as(C.POINTER(C.c_int64)), eos_np.size, C.byref(slot)))
        self._tickets[slot.value] = (env_id, inputs)
        return SimpleNamespace(env_id=env_id, slot=slot.value)

    @torch.no_grad()
    def step_batch(self):
        """One scheduler iteration (one pass over the weights for every turn in flight).  Returns (finished, running): `finished` is
        a list of (ticket, GenerateOutput) for the turns that ended in this iteration, `running` the number still in flight."""
        running, nf = C.c_int32(), C.c_int32()
        fin = (C.c_int32 * 8)()
        try:
            _check(self._lib.svln_batch_step(self._h, C.byref(running), fin, C.byref(nf)))
        except Exception:
            # the engine has dropped the failing turn(s); drop the rest too so that tickets and engine slots cannot disagree
            self._lib.svln_batch_cancel(self._h, -1)
            self._tickets.clear()
            raise
        done = []
        cap = self.cfg.max_positions
        for k in range(nf.value):
            out = np.zeros(cap, dtype=np.int64)
            n, env = C.c_int32(), C.c_int32()
            _check(self._lib.svln_batch_result(self._h, fin[k], C.byref(env), out.ctypes.data_as(C.POINTER(C.c_int64)), cap, C.byref(n)))
            env_id, inputs = self._tickets.pop(fin[k])
            done.append((SimpleNamespace(env_id=env_id, slot=fin[k]), self._result(env_id, out[: n.value], inputs)))
        return done, running.value

    def cancel(self, ticket=None):
        """Drop a submitted turn (ticket from `submit`), or every turn in flight (ticket=None)."""
        slot = -1 if ticket is None else int(ticket.slot)
        _check(self._lib.svln_batch_cancel(self._h, slot))
        if ticket is None:
            self._tickets.clear()
        else:
            self._tickets.pop(slot, None)

    def last_hidden_batch(self, slot: int) -> np.ndarray:
        buf = np.empty((8, self.cfg.hidden), dtype=np.float32)
        n = C.c_int32()
        _check(self._lib.svln_get_hidden_batch(self._h, slot, buf.ctypes.data_as(C.POINTER(C.c_float)), 8, C.byref(n)))
        return buf[: n.value].copy()

    # ---- parity taps (tests) / perf helpers (bench) ---------------------------------------------------
    def last_hidden(self) -> np.ndarray:
        buf = np.empty((64, self.cfg.hidden), dtype=np.float32)
        n = C.c_int32()
        _check(self._lib.svln_get_hidden(self._h, buf.ctypes.data_as(C.POINTER(C.c_float)), 64, C.byref(n)))
        return buf[: n.value].copy()

    def set_layer_taps(self, enable: bool, probe_layer: int = -1):
        """test taps of the prefill: last residual row after every LLM layer; all rows around `probe_layer` (see include/streamvln_hip.h)"""
        _check(self._lib.svln_set_layer_taps(self._h, int(enable), int(probe_layer)))

    def layer_taps(self) -> np.ndarray:
        out = np.empty((self.cfg.layers, self.cfg.hidden), dtype=np.float32)
        _check(self._lib.svln_get_layer_taps(self._h, out.ctypes.data_as(C.POINTER(C.c_float))))
        return out

    def layer_probe(self, which: int, max_rows: int = 512) -> np.ndarray:
        """operand `which` of the probed layer as the last prefill saw it (0 x_in, 1 x_out, 2 attention out, 3 x after attention,
        4 post-attention norm, 5 SwiGLU product, 6 input norm, 7 q|k|v after RoPE): fp32 [rows, cols]"""
        widest = max(self.cfg.inter, self.cfg.q_dim + 2 * self.cfg.kv_dim, self.cfg.hidden)
        out = np.empty(max_rows * widest, dtype=np.float32)
        n, cols = C.c_int32(), C.c_int32()
        _check(self._lib.svln_get_layer_probe(self._h, int(which), out.ctypes.data_as(C.POINTER(C.c_float)), out.size, C.byref(n), C.byref(cols)))
        return out[: n.value * cols.value].reshape(n.value, cols.value).copy()

    def env_state(self, env_id=0):
        ne, kl = C.c_int32(), C.c_int32()
        _check(self._lib.svln_env_state(self._h, self._slot(env_id), C.byref(ne), C.byref(kl)))
        return ne.value, kl.value

    def get_embeds(self, env_id, start, n) -> np.ndarray:
        out = np.empty((n, self.cfg.hidden), dtype=np.float32)
        _check(self._lib.svln_get_embeds(self._h, self._slot(env_id), start, n, out.ctypes.data_as(C.POINTER(C.c_float))))
        return out

    def preprocess_time(self, reset: bool = False):
        """(GPU ms, frames) spent in the image processor's HIP path since the last reset"""
        ms, n = C.c_double(), C.c_int64()
        _check(self._lib.svln_preprocess_time(self._h, C.byref(ms), C.byref(n), int(reset)))
        return ms.value, n.value

    def set_feature_cache(self, capacity_frames: int):
        """Memoise pooled frame features by pixel content (opt-in; 0 = re-encode every frame like the reference)."""
        _check(self._lib.svln_set_feature_cache(self._h, int(capacity_frames)))

    def feature_cache_stats(self):
        hits, misses = C.c_int64(), C.c_int64()
        _check(self._lib.svln_feature_cache_stats(self._h, C.byref(hits), C.byref(misses)))
        return hits.value, misses.value

    def set_decode_graph(self, enable: bool):
        _check(self._lib.svln_set_decode_graph(self._h, int(enable)))

    def set_decode_persistent(self, enable: bool):
        """single-env decode step as attention + ONE persistent launch per layer (svln_set_decode_persistent) instead of six launches"""
        _check(self._lib.svln_set_decode_persistent(self._h, int(enable)))

    def set_memory_prune(self, keep_tokens: int):
        """Opt-in extension (BASELINE configs[3]; no reference counterpart): `<memory>` expands to the `keep_tokens`
        memory tokens least similar to the mean memory token instead of all num_history x 196.  0 restores the reference behaviour."""
        _check(self._lib.svln_set_memory_prune(self._h, int(keep_tokens)))

    def set_fp8_decode(self, enable: bool):
        """Opt-in extension (the reference is bf16 only): decode steps and the lm_head read e4m3 copies of the LLM
        weights (per-row scale).  Prefill, vision and generate_batch keep bf16.  bf16 engines only."""
        _check(self._lib.svln_set_fp8_decode(self._h, int(enable)))

    def set_fp8_gemm(self, enable: bool):
        """Opt-in extension (SURVEY.md 8f-2 / BASELINE configs[4]): the LLM's multi-row products (prefill; decode steps of >= 4 batched
        envs) run as e4m3 MFMA products with per-row weight and activation scales.  bf16 engines only; reduced precision."""
        _check(self._lib.svln_set_fp8_gemm(self._h, int(enable)))

    def sync(self):
        _check(self._lib.svln_sync(self._h))

    def close(self):
        if getattr(self, "_h", None):
            proc = self._tower.image_processor
            if proc._ext_stream is not None:       # nothing on the torch side may still be ordered against the engine's stream
                dev = proc._ext_stream.device
                proc._ext_stream.synchronize()
                torch.cuda.current_stream(dev).synchronize()
                if torch.cuda.current_stream(dev).cuda_stream == proc._ext_stream.cuda_stream:
                    # the caller adopted the engine's stream (`torch_stream`): it is about to be destroyed, so torch goes back to its
                    # default stream (a later collective or kernel on a destroyed stream fails with hipErrorInvalidValue)
                    torch.cuda.set_stream(torch.cuda.default_stream(dev))
                proc._ext_stream = None
            proc._engine = None
            proc.backend = "closed"
            self._lib.svln_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
