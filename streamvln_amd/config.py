"""Model dimensions of the StreamVLN streaming-inference path.

Reference sources for the numbers:
  * vision tower defaults  -- llava/model/multimodal_encoder/siglip_encoder.py:73-86
    (27 layers built, the last one deleted at :570 -> 26 run; head = Identity)
  * projector mlp2x_gelu   -- llava/model/multimodal_projector/builder.py:41-48
  * pooled tokens / frame  -- streamvln/model/stream_video_vln.py:53-73 (27x27 -> 14x14)
  * LLM = Qwen2-7B-Instruct -- scripts/streamvln_train_slurm.sh:13

`head_dim` values (72 vision, 128 LLM) are structural for the HIP attention kernels
and are kept by every config, including TINY.
"""
from __future__ import annotations

from dataclasses import dataclass, asdict, replace
import math


@dataclass(frozen=True)
class StreamVLNConfig:
    name: str
    # SigLIP vision tower
    v_hidden: int = 1152
    v_inter: int = 4304
    v_heads: int = 16
    v_layers: int = 26          # layers actually run (27 built, last deleted)
    v_patch: int = 14
    v_image: int = 384
    v_eps: float = 1e-6
    # projector + pooling
    pool_stride: int = 2
    # Qwen2 LLM
    hidden: int = 3584
    layers: int = 28
    q_heads: int = 28
    kv_heads: int = 4
    head_dim: int = 128
    inter: int = 18944
    vocab: int = 152064
    rope_theta: float = 1e6
    rms_eps: float = 1e-6
    max_positions: int = 4096   # --model_max_length 4096 (streamvln_eval.py:501)

    @property
    def v_head_dim(self) -> int:
        return self.v_hidden // self.v_heads

    @property
    def v_side(self) -> int:            # 27
        return self.v_image // self.v_patch

    @property
    def v_tokens(self) -> int:          # 729
        return self.v_side ** 2

    @property
    def pool_side(self) -> int:         # ceil(27/2) = 14  (stream_video_vln.py:66)
        return math.ceil(self.v_side / self.pool_stride)

    @property
    def pool_tokens(self) -> int:       # 196
        return self.pool_side ** 2

    @property
    def patch_k(self) -> int:           # 3*14*14 = 588
        return 3 * self.v_patch * self.v_patch

    @property
    def q_dim(self) -> int:
        return self.q_heads * self.head_dim

    @property
    def kv_dim(self) -> int:
        return self.kv_heads * self.head_dim

    def to_dict(self):
        return asdict(self)


#: StreamVLN-Qwen-1.5 at true size (SigLIP-so400m/14-384 + Qwen2-7B).
TRUE = StreamVLNConfig(name="streamvln_qwen2_7b")

#: Small configuration with the same structure (head dims, patching, pooling, GQA)
#: used for end-to-end parity runs that the CPU oracle finishes in seconds.
TINY = StreamVLNConfig(
    name="tiny",
    v_hidden=144, v_inter=320, v_heads=2, v_layers=2,
    hidden=512, layers=2, q_heads=4, kv_heads=2, inter=1024, vocab=4096,
)

#: One-layer true-dimension configuration (single ViT layer + single LLM layer, small vocab)
#: for true-dim golden vectors.
TRUE1 = StreamVLNConfig(name="true_dims_1layer", v_layers=1, layers=1, vocab=8192)

#: True dimensions, four ViT layers + four LLM layers, FULL vocabulary (152 064): the fused inter-layer hand-offs
#: (down_proj reduce -> next input_layernorm, fc2 reduce -> next layer_norm1) and the full-size lm_head / arg-max at true width.
TRUE4 = StreamVLNConfig(name="true_dims_4layer", v_layers=4, layers=4)

CONFIGS = {c.name: c for c in (TRUE, TINY, TRUE1, TRUE4)}

IGNORE_INDEX = -100          # streamvln/utils/utils.py:8
IMAGE_TOKEN_INDEX = -200     # streamvln/utils/utils.py:9
MEMORY_TOKEN_INDEX = -300    # streamvln/utils/utils.py:15
DEFAULT_IMAGE_TOKEN = "<image>"
DEFAULT_MEMORY_TOKEN = "<memory>"
DEFAULT_VIDEO_TOKEN = "<video>"


def config_from_hf(hf, base: "StreamVLNConfig" = None, name: str = "from_checkpoint") -> "StreamVLNConfig":
    """Map a checkpoint's HF config (the dict of `config.json`, or the `LlavaQwenConfig` / `AutoConfig` object the reference's
    harness builds, streamvln_eval.py:521-527) onto StreamVLNConfig.  Only the Qwen2 fields vary between checkpoints of this
    family; the SigLIP-so400m tower dims are literals in the reference (siglip_encoder.py:73-86) and stay those of `base`.
    Unsupported structure (head_dim != 128, sliding window, rope scaling, a pooling mode other than bilinear) is an error here
    rather than a wrong answer on the GPU."""
    base = base or TRUE
    get = (lambda k, d=None: hf.get(k, d)) if isinstance(hf, dict) else (lambda k, d=None: getattr(hf, k, d))
    hidden = int(get("hidden_size", base.hidden))
    q_heads = int(get("num_attention_heads", base.q_heads))
    kv_heads = int(get("num_key_value_heads", q_heads))
    head_dim = int(get("head_dim", None) or hidden // q_heads)
    if head_dim != 128:
        raise ValueError(f"head_dim {head_dim} is not supported (the attention kernels are built for 128)")
    if get("rope_scaling", None) not in (None, {}):
        raise ValueError("rope_scaling is not supported (the reference forces it to None, stream_video_vln.py:40)")
    if get("use_sliding_window", False):
        raise ValueError("sliding-window attention is not supported")
    mode = get("mm_spatial_pool_mode", "bilinear")
    if mode != "bilinear":
        raise ValueError(f"Unexpected mm_spatial_pool_mode: {mode}")       # stream_video_vln.py:69-70
    ptype = get("mm_projector_type", "mlp2x_gelu")
    if ptype != "mlp2x_gelu":
        raise ValueError(f"mm_projector_type {ptype} is not supported (mlp2x_gelu only)")
    if q_heads % kv_heads != 0 or q_heads // kv_heads > 32:
        raise ValueError("num_attention_heads must be a multiple of num_key_value_heads with a group size <= 32")
    return replace(base, name=name, hidden=hidden, layers=int(get("num_hidden_layers", base.layers)), q_heads=q_heads, kv_heads=kv_heads,
                   head_dim=head_dim, inter=int(get("intermediate_size", base.inter)), vocab=int(get("vocab_size", base.vocab)),
                   rope_theta=float(get("rope_theta", base.rope_theta)), rms_eps=float(get("rms_norm_eps", base.rms_eps)),
                   pool_stride=int(get("mm_spatial_pool_stride", base.pool_stride)))
