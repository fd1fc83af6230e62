"""Generate `tests/golden/*.npz` by running the REFERENCE's own modules in the build container
and pin the CPU oracle against them.

TEST INFRASTRUCTURE ONLY.  Run here (needs /root/reference):  python -m oracle.make_golden
The fixtures hold inputs by seed and expected outputs (token ids, top-2 margins, final-norm
hidden states, sampled vision features, cache lengths) -- data only, no reference source.

Scenarios
  tiny_episode   TINY config, 36 env steps = 9 model turns over 3 windows
                 (num_frames 12, num_future_steps 4, num_history 2): episode start, steady turns,
                 two window restarts with a 2-frame <memory> block; EOS set = ids % 3 == 2 so
                 turns stop after a varying number of tokens (cap 6).
  true1_episode  true dimensions, one ViT layer + one LLM layer, vocab 8192, 36 env steps = 9 turns:
                 first turn (181 ids, T=376), seven steady turns (T=214) and the window restart at
                 step 32 (9 views, 1568-row <memory> block, T=1952); decode capped at 3 tokens.
  true4_episode  true dimensions, 4 ViT + 4 LLM layers, FULL vocabulary 152 064: first turn + one
                 steady turn, 4 tokens each (inter-layer fused norms, full-size lm_head / arg-max).
  tiny_truncate  TINY, config.tokenizer_model_max_length = 150: the reference truncates every turn's spliced rows
                 (stream_video_vln.py:241-244).
  tiny_penalty   TINY, generation_config.repetition_penalty = 1.3 through transformers' own
                 RepetitionPenaltyLogitsProcessor in the restated 4.45.1 greedy loop.
Usage: python -m oracle.make_golden [scenario ... | preprocess]   (default: everything)
Each scenario is driven through the same `StreamingAgent` twice (reference, oracle); the
script asserts ids equal and hidden/features within 2e-4 abs+rel before writing.
"""
from __future__ import annotations

import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from streamvln_amd import weights as Wt                      # noqa: E402
from streamvln_amd.synthetic import synthetic_frame          # noqa: E402
from oracle import ref_harness as RH                          # noqa: E402
from oracle import streamvln_oracle as O                      # noqa: E402

GOLD = os.path.join(ROOT, "tests", "golden")
sys.path.insert(0, os.path.join(ROOT, "tests"))
from scenarios import SCENARIOS, SEED, apply_knobs, run_scenario            # noqa: E402


class _RefOut:
    def __init__(self, ids, cache, hidden, embeds):
        self.sequences = torch.tensor([ids], dtype=torch.long)
        self.past_key_values = cache
        self.hidden = hidden
        self.embeds = embeds
        self.cache_len = cache.get_seq_length()      # at turn time (the cache object keeps growing)


class RefAdapter:
    """Reference model behind the agent's call surface."""

    def __init__(self, model):
        self.m = model

    def reset_for_env(self, i):
        self.m.reset_for_env(i)

    def generate(self, inputs, images, env_id, time_ids, past_key_values, max_new_tokens, eos_token_ids, **_):
        ids, cache, hid, emb = RH.reference_turn(self.m, env_id, inputs, images.float(), time_ids, past_key_values,
                                                 max_new_tokens, eos_token_ids)
        return _RefOut(ids, cache, hid, emb)


def run(model, sc, tap_embeds):
    embeds = []
    log = run_scenario(model, sc, preprocess=lambda rgb: torch.from_numpy(O.siglip_preprocess(rgb)),
                       on_turn=lambda t, rec: embeds.append(tap_embeds(rec["out"])))
    return log, embeds


def close(a, b, tol=2e-4):
    a, b = np.asarray(a, dtype=np.float64), np.asarray(b, dtype=np.float64)
    return np.abs(a - b).max() <= tol * (1.0 + np.abs(b).max())


def structured_frames():
    """640x480 uint8 frames with smooth, saturated and high-contrast content (bicubic overshoot -> clip8 on both ends)."""
    y, x = np.mgrid[0:480, 0:640]
    grad = np.stack([(x * 255 // 639), (y * 255 // 479), ((x + y) * 255 // 1118)], -1).astype(np.uint8)
    checker = (((x // 3 + y // 5) % 2) * 255).astype(np.uint8)[..., None].repeat(3, -1)
    stripes = np.stack([((x % 7) < 2) * 255, ((y % 4) < 1) * 255, ((x * y) % 256)], -1).astype(np.uint8)
    return {"grad": grad, "checker": checker, "stripes": stripes}


def preprocess_fixture():
    """G1: image processor pin (siglip_encoder.py:47-67) through the reference's own class: 64 sampled values of frame 0 (as in
    round 1) plus, for 4 synthetic and 3 structured frames, sha256 of the whole fp32 [3,384,384] output and of its uint8 form
    (the PIL resize result, CHW), per-channel sums and one full row -- data only."""
    import hashlib
    _, _, _, _, Proc = RH.import_reference()
    from PIL import Image
    proc = Proc()
    frames = {f"synthetic_{s}": synthetic_frame(0, s) for s in range(4)}
    frames.update(structured_frames())
    fx = {}
    for k, frame in frames.items():
        pv = proc.preprocess(images=Image.fromarray(frame).convert("RGB"), return_tensors="pt")["pixel_values"][0].numpy()
        mine = O.siglip_preprocess(frame)
        assert np.array_equal(pv, mine), k
        u8 = np.rint((pv.astype(np.float64) * 0.5 + 0.5) * 255.0).astype(np.uint8)          # CHW uint8 = the PIL resize result
        fx[f"{k}_sha256_f32"] = np.frombuffer(hashlib.sha256(np.ascontiguousarray(pv).tobytes()).digest(), dtype=np.uint8)
        fx[f"{k}_sha256_u8"] = np.frombuffer(hashlib.sha256(np.ascontiguousarray(u8).tobytes()).digest(), dtype=np.uint8)
        fx[f"{k}_chan_sum_u8"] = u8.reshape(3, -1).astype(np.int64).sum(1)
        fx[f"{k}_row100_u8"] = u8[:, 100, :].copy()
        if k == "synthetic_0":
            idx = np.random.default_rng(5).integers(0, pv.size, 64)
            fx.update(flat_idx=idx, values=pv.reshape(-1)[idx], sum=np.float64(pv.astype(np.float64).sum()))
    fx["frame_keys"] = np.asarray(sorted(frames.keys()))
    np.savez_compressed(os.path.join(GOLD, "preprocess.npz"), **fx)
    print(f"preprocess: exact on {len(frames)} frames")


def main():
    os.makedirs(GOLD, exist_ok=True)
    torch.set_num_threads(8)
    want = sys.argv[1:] or list(SCENARIOS) + ["preprocess"]
    for name, sc in SCENARIOS.items():
        if name not in want:
            continue
        cfg = sc["cfg"]
        t0 = time.time()
        sd = Wt.synth_state_dict(cfg, SEED, bf16_round=True)
        ref = RH.build_reference_model(cfg, sd, sc["num_history"])
        ref.reset(1)
        orc = O.OracleStreamVLN(cfg, sd, num_history=sc["num_history"])
        apply_knobs(ref, sc)
        apply_knobs(orc, sc)
        log_r, emb_r = run(RefAdapter(ref), sc, lambda out: out.embeds)
        # oracle: per-turn embeds = the rows appended to its cache this turn
        seen = [0]

        def tap(out, orc=orc, seen=seen):
            E = orc.cache[0].get("inputs_embeds")
            if orc.curr_t[0] == 1:
                seen[0] = 0
            new = E[seen[0]:].clone()
            seen[0] = E.shape[0]
            return new
        log_o, emb_o = run(orc, sc, tap)
        assert len(log_r) == len(log_o)
        fx = {"n_turns": np.int64(len(log_r)), "seed": np.int64(SEED)}
        for t, (r, o) in enumerate(zip(log_r, log_o)):
            ids_r = r["out"].sequences[0].tolist()
            ids_o = o["out"].sequences[0].tolist()
            assert ids_r == ids_o, (name, t, ids_r, ids_o)
            assert r["n_inputs"] == o["n_inputs"] and r["views"] == o["views"]
            assert close(o["out"].hidden, r["out"].hidden), (name, t, "hidden")
            assert close(emb_o[t], emb_r[t]), (name, t, "embeds")
            cache_len = r["out"].cache_len
            assert cache_len == o["out"].cache_len
            e = emb_r[t].numpy()
            fx[f"t{t}_step_id"] = np.int64(r["step_id"])
            fx[f"t{t}_views"] = np.int64(r["views"])
            fx[f"t{t}_memory"] = np.int64(r["memory"])
            fx[f"t{t}_n_inputs"] = np.int64(r["n_inputs"])
            fx[f"t{t}_ids"] = np.asarray(ids_r, dtype=np.int64)
            fx[f"t{t}_margins"] = np.asarray(o["out"].margins, dtype=np.float32)
            fx[f"t{t}_hidden"] = r["out"].hidden.numpy().astype(np.float32)
            fx[f"t{t}_cache_len"] = np.int64(cache_len)
            fx[f"t{t}_embeds_rows"] = np.int64(e.shape[0])
            # vision/splice taps: 4 full columns, 3 full rows and per-row sums of this turn's embeds
            fx[f"t{t}_embeds_cols"] = e[:, [0, 1, e.shape[1] // 2, e.shape[1] - 1]].astype(np.float32)
            fx[f"t{t}_embeds_sel"] = e[[0, e.shape[0] // 2, e.shape[0] - 1]].astype(np.float32)
            fx[f"t{t}_embeds_rowsum"] = e.sum(1).astype(np.float32)
            print(f"{name} turn {t}: step {r['step_id']} views {r['views']} mem {r['memory']} "
                  f"T_embeds {e.shape[0]} ids {ids_r} min-margin {min(o['out'].margins):.4f}")
        np.savez_compressed(os.path.join(GOLD, name + ".npz"), **fx)
        print(f"{name}: wrote {len(log_r)} turns in {time.time() - t0:.1f}s")
        del ref, orc, sd, log_r, log_o, emb_r, emb_o
        import gc
        gc.collect()

    if "preprocess" in want:
        preprocess_fixture()


if __name__ == "__main__":
    main()
