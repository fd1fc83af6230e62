"""Import the reference's own modules (build container only) to pin the oracle.

TEST INFRASTRUCTURE ONLY -- used by `oracle/make_golden.py`; `/root/reference` does not
exist on the GPU box, so nothing in the GPU tests, smoke() or bench.py imports this.

Two harness-side shims (the reference stays untouched):
  1. a stub `llava.model.multimodal_resampler.qformer` (the real file imports helpers
     that transformers 5.x removed; it is unused on this path);
  2. a config subclass whose `rope_scaling = None` assignment
     (streamvln/model/stream_video_vln.py:40) does not null `rope_parameters`.
`StreamVLNForCausalLM.generate()` is written against transformers 4.45.1 internals and
fails under the installed 5.x (stream_video_vln.py:430), so the greedy loop below drives
the reference's `prepare_inputs_labels_for_multimodal` + `forward` with the 4.45.1 protocol
(cache_position = arange(L_total)[P:], EOS appended but never fed).
"""
from __future__ import annotations

import sys
import types

import numpy as np
import torch

REF = "/root/reference"


def import_reference():
    for p in (REF, REF + "/streamvln"):
        if p not in sys.path:
            sys.path.insert(0, p)
    name = "llava.model.multimodal_resampler.qformer"
    if name not in sys.modules:
        stub = types.ModuleType(name)

        class Qformer(torch.nn.Module):        # never instantiated on this path
            def __init__(self, *a, **k):
                super().__init__()

        stub.Qformer = Qformer
        sys.modules[name] = stub
    from model.stream_video_vln import StreamVLNForCausalLM           # noqa: E402
    from llava.model.language_model.llava_qwen import LlavaQwenConfig  # noqa: E402
    from llava.model.multimodal_encoder.siglip_encoder import (        # noqa: E402
        SigLipVisionConfig, SigLipVisionModel, SigLipImageProcessor)
    return StreamVLNForCausalLM, LlavaQwenConfig, SigLipVisionConfig, SigLipVisionModel, SigLipImageProcessor


def build_reference_model(cfg, weights, num_history):
    """Construct the reference StreamVLNForCausalLM (fp32, eager attention, CPU) for `cfg`
    and load the synthetic `weights` (HF names) into it."""
    StreamVLN, LlavaQwenConfig, SigLipVisionConfig, SigLipVisionModel, _ = import_reference()

    class _Cfg(LlavaQwenConfig):
        def __setattr__(self, k, v):
            if k == "rope_scaling" and v is None:
                return
            super().__setattr__(k, v)

    hf = _Cfg(
        vocab_size=cfg.vocab, hidden_size=cfg.hidden, intermediate_size=cfg.inter,
        num_hidden_layers=cfg.layers, num_attention_heads=cfg.q_heads, num_key_value_heads=cfg.kv_heads,
        max_position_embeddings=cfg.max_positions, rms_norm_eps=cfg.rms_eps,
        rope_parameters={"rope_type": "default", "rope_theta": cfg.rope_theta},
        tie_word_embeddings=False, attention_dropout=0.0, use_sliding_window=False,
    )
    hf.mm_vision_tower = "google/siglip-so400m-patch14-384"
    hf.mm_projector_type = "mlp2x_gelu"
    hf.mm_hidden_size = cfg.v_hidden
    hf.mm_spatial_pool_mode = "bilinear"
    hf.mm_patch_merge_type = "flat"
    hf.num_history = num_history
    hf._attn_implementation = "eager"
    model = StreamVLN(hf)
    tower = model.get_model().get_vision_tower()
    # mirror SigLipVisionTower.load_model (siglip_encoder.py:563-574) without the hub fetch
    vcfg = SigLipVisionConfig(hidden_size=cfg.v_hidden, intermediate_size=cfg.v_inter,
                              num_hidden_layers=cfg.v_layers + 1, num_attention_heads=cfg.v_heads,
                              image_size=cfg.v_image, patch_size=cfg.v_patch, layer_norm_eps=cfg.v_eps)
    vcfg._attn_implementation = "eager"
    tower.config = vcfg
    tower.vision_tower = SigLipVisionModel(vcfg)
    del tower.vision_tower.vision_model.encoder.layers[-1:]
    tower.vision_tower.vision_model.head = torch.nn.Identity()
    tower.vision_tower.requires_grad_(False)
    tower.is_loaded = True
    model.model.num_history = num_history
    model.requires_grad_(False)
    model.float().eval()
    sd = model.state_dict()
    missing = []
    with torch.no_grad():
        for k, v in sd.items():
            if k in weights:
                v.copy_(torch.from_numpy(np.asarray(weights[k], dtype=np.float32)).view_as(v))
            elif "post_layernorm" in k or "rotary" in k or "position_ids" in k:
                continue
            else:
                missing.append(k)
    assert not missing, missing[:8]
    unused = [k for k in weights if k not in sd]
    assert not unused, unused[:8]
    return model


@torch.no_grad()
def reference_turn(model, env_id, inputs, images, time_ids, past, max_new_tokens, eos_ids):
    """One `generate` turn through the reference's own code paths, restating only the
    4.45.1 GenerationMixin bookkeeping that 5.x no longer matches.
    Returns (new token ids, cache, final-norm hidden per generated token, inputs_embeds of this turn)."""
    from transformers import DynamicCache
    V = images.shape[1]
    dummy = dict(depths=torch.zeros(1, V, 4, 4), poses=torch.zeros(1, V, 4, 4), intrinsics=torch.zeros(1, V, 4, 4))
    (_, _, _, _, inputs_embeds, _) = model.prepare_inputs_labels_for_multimodal(
        inputs, None, None, None, None, images, None, dummy["depths"], dummy["poses"], dummy["intrinsics"],
        time_ids, [0])
    # StreamVLNForCausalLM.generate, stream_video_vln.py:396-401
    if model.curr_t[env_id] == 0:
        model.cache[env_id]["inputs_embeds"] = inputs_embeds
    else:
        model.cache[env_id]["inputs_embeds"] = torch.cat([model.cache[env_id]["inputs_embeds"], inputs_embeds], dim=1)
    model.curr_t[env_id] += 1
    E = model.cache[env_id]["inputs_embeds"]
    cache = past if past is not None else DynamicCache()
    P, L_total = cache.get_seq_length(), E.shape[1]
    eos = set(int(e) for e in eos_ids)

    def fwd(**kw):
        o = model.model(past_key_values=cache, use_cache=True, **kw)      # Qwen2Model incl. final norm
        h = o.last_hidden_state[0, -1]
        return h, model.lm_head(h)

    # generation_config.repetition_penalty: GenerationMixin._get_logits_processor adds transformers' own
    # RepetitionPenaltyLogitsProcessor (greedy included); its input_ids are the generated ids only (the prompt is inputs_embeds)
    pen = float(getattr(model.generation_config, "repetition_penalty", None) or 1.0)
    proc = None
    if pen != 1.0:
        from transformers import RepetitionPenaltyLogitsProcessor
        proc = RepetitionPenaltyLogitsProcessor(penalty=pen)
    h, logits = fwd(inputs_embeds=E[:, P:])
    out, hid = [], []
    while True:
        scores = logits.float()
        if proc is not None and out:
            scores = proc(torch.tensor([out], dtype=torch.long), scores[None].clone())[0]
        tok = int(torch.argmax(scores))
        out.append(tok); hid.append(h.clone())
        if tok in eos or len(out) >= max_new_tokens:
            break
        h, logits = fwd(input_ids=torch.tensor([[tok]]))
    assert cache.get_seq_length() == L_total + len(out) - 1
    return out, cache, torch.stack(hid), inputs_embeds[0]
