"""CPU: the oracle (fp32 restatement) reproduces the fixtures generated from the REFERENCE's own modules
(oracle/make_golden.py).  tiny_episode is replayed in full; true1_episode (true dimensions) for its first turn."""
import numpy as np
import pytest
import torch

from oracle import streamvln_oracle as O
from scenarios import SCENARIOS, SEED, apply_knobs, run_scenario
from streamvln_amd import weights as W
from streamvln_amd.synthetic import synthetic_frame
from util import load_golden


def _replay(name, steps=None):
    sc, g = SCENARIOS[name], load_golden(name)
    cfg = sc["cfg"]
    orc = O.OracleStreamVLN(cfg, W.synth_state_dict(cfg, SEED), num_history=sc["num_history"])
    apply_knobs(orc, sc)
    log = run_scenario(orc, sc, preprocess=lambda rgb: torch.from_numpy(O.siglip_preprocess(rgb)), steps=steps)
    for t, rec in enumerate(log):
        out = rec["out"]
        assert out.sequences[0].tolist() == g[f"t{t}_ids"].tolist()
        assert out.cache_len == int(g[f"t{t}_cache_len"])
        assert np.abs(out.hidden.numpy() - g[f"t{t}_hidden"]).max() <= 2e-4 * (1 + np.abs(g[f"t{t}_hidden"]).max())
        assert rec["views"] == int(g[f"t{t}_views"]) and rec["memory"] == bool(g[f"t{t}_memory"])
    return len(log)


def test_tiny_episode_matches_reference_fixture():
    assert _replay("tiny_episode") == 9


def test_truncated_turns_match_reference_fixture():
    """config.tokenizer_model_max_length = 150 through the reference's own prepare_inputs_labels_for_multimodal (stream_video_vln.py:241-244)"""
    assert _replay("tiny_truncate") == 3
    g = load_golden("tiny_truncate")
    assert [int(g[f"t{t}_embeds_rows"]) for t in range(3)] == [150, 150, 150]


def test_repetition_penalty_matches_reference_fixture_and_transformers():
    """generation_config.repetition_penalty = 1.3: the fixture was produced with transformers' own RepetitionPenaltyLogitsProcessor inside
    the restated greedy loop (oracle/ref_harness.py); the oracle's formula equals that class on random logits, negative values included."""
    assert _replay("tiny_penalty") == 4
    from transformers import RepetitionPenaltyLogitsProcessor
    g = torch.Generator().manual_seed(3)
    logits = torch.randn(500, generator=g) * 3
    gen = [7, 7, 13, 499, 0, 250]
    exp = RepetitionPenaltyLogitsProcessor(penalty=1.3)(torch.tensor([gen]), logits[None].clone())[0]
    got = O.repetition_penalty(logits, gen, 1.3)
    assert torch.equal(got, exp) and not torch.equal(got, logits)
    assert torch.equal(O.repetition_penalty(logits, gen, 1.0), logits) and torch.equal(O.repetition_penalty(logits, [], 1.3), logits)


def test_true_dims_first_turn_matches_reference_fixture():
    assert _replay("true1_episode", steps=1) == 1


def test_preprocess_matches_reference_fixture():
    g = load_golden("preprocess")
    pv = O.siglip_preprocess(synthetic_frame(0, 0))
    assert pv.shape == (3, 384, 384) and pv.dtype == np.float32
    assert np.array_equal(pv.reshape(-1)[g["flat_idx"]], g["values"])
    assert abs(float(pv.astype(np.float64).sum()) - float(g["sum"])) < 1e-6


def test_cache_bookkeeping_and_incremental_equals_one_shot():
    """a-9: after a turn cache_len = L_total + n_new - 1; next turn prefill starts at the previous EOS embed;
    the incremental first-token of a later turn equals an uncached forward over all embeds."""
    sc = SCENARIOS["tiny_episode"]
    cfg = sc["cfg"]
    w = W.synth_state_dict(cfg, SEED)
    orc = O.OracleStreamVLN(cfg, w, num_history=2)
    img = torch.from_numpy(O.siglip_preprocess(synthetic_frame(0, 0)))[None, None]
    o1 = orc.generate(np.array([[11, 12, -200, 13]]), img, env_id=0, time_ids=[[0]], max_new_tokens=3)
    L1 = orc.cache[0]["inputs_embeds"].shape[0]
    assert L1 == 3 + 196 and o1.cache_len == L1 + 3 - 1
    ids2 = o1.sequences[0].tolist() + [21, -200, 22]
    o2 = orc.generate(np.array([ids2]), img, env_id=0, time_ids=[[0, 1, 2, 3, 4]], past_key_values=o1.past_key_values, max_new_tokens=1)
    E = orc.cache[0]["inputs_embeds"]
    assert E.shape[0] == L1 + 3 + 2 + 196
    h = O.qwen2_forward(orc.w, cfg, E, 0, O.KVCache(cfg.layers))[-1]
    tok, _ = O.greedy_pick(O.lm_logits(orc.w, h))
    assert tok == int(o2.sequences[0, 0])
    assert np.abs(h.numpy() - o2.hidden[0].numpy()).max() < 1e-4


def test_pool_restatement_equals_interpolate():
    from streamvln_amd.config import TINY
    x = torch.rand(2, 729, 16)
    ref = torch.nn.functional.interpolate(x.view(2, 27, 27, 16).permute(0, 3, 1, 2), size=[14, 14], mode="bilinear")
    assert torch.allclose(O.pool_bilinear(TINY, x), ref.permute(0, 2, 3, 1).reshape(2, 196, 16), atol=1e-5)


def test_memory_prune_extension_definition():
    """prune_memory_tokens (opt-in extension; the reference has no counterpart, SURVEY a-13): keeps the `keep` tokens least similar to
    the mean token, in order, ties to the lower index; the oracle's memory turns shrink accordingly."""
    g = torch.Generator().manual_seed(5)
    mem = torch.randn(50, 16, generator=g) + 2.0
    mem[7] = -mem[7]                      # the most atypical token
    mem[20] = mem[10]                     # an exact tie
    idx, score = O.prune_memory_tokens(mem, 5)
    assert idx.tolist() == sorted(idx.tolist()) and len(idx) == 5 and 7 in idx.tolist()
    ref = torch.nn.functional.cosine_similarity(mem, mem.mean(0, keepdim=True), dim=1)
    assert torch.allclose(score, ref, atol=1e-6)
    assert set(idx.tolist()) == set(sorted(range(50), key=lambda i: (float(ref[i]), i))[:5])
    full, _ = O.prune_memory_tokens(mem, 50)
    assert full.tolist() == list(range(50))
    k = [i for i in range(50) if float(score[i]) <= float(score[10])]
    idx2, _ = O.prune_memory_tokens(mem, len(k) - 1)     # cut between the tied pair: 10 stays, 20 goes
    assert 10 in idx2.tolist() and 20 not in idx2.tolist()


def test_fp8_emulation_arithmetic_and_neutral_mode():
    """oracle.Fp8Emu (extension, no reference counterpart): the per-row e4m3 quantise -> dequantise restates gemv.hip quant_fp8_rows_kernel
    (scale = amax / 448, 1 for a zero row, RNE, idempotent on its own output, every value a multiple of an e4m3 step of its row);
    with both modes off the oracle's arithmetic -- hence the reference-pinned fixture -- is untouched; with them on it changes."""
    g = torch.Generator().manual_seed(5)
    x = torch.randn(6, 64, generator=g) * torch.tensor([1e-3, 1.0, 50.0, 3e4, 1.0, 1.0])[:, None]
    x[4] = 0.0
    x[5, 0] = 1000.0                                      # an outlier sets the row's scale
    y = O.qdq_e4m3_rows(x)
    assert torch.equal(O.qdq_e4m3_rows(y), y)             # idempotent
    assert torch.equal(y[4], torch.zeros(64)) and float(y[5, 0]) == 1000.0
    sc = x.abs().amax(-1, keepdim=True) / 448.0
    sc[4] = 1.0
    q = y / sc
    assert torch.allclose(q, q.to(torch.float8_e4m3fn).float(), rtol=1e-6, atol=0)   # every dequantised value sits on the e4m3 grid of its row
    rel = ((y - x).norm(dim=1) / x.norm(dim=1).clamp(min=1e-30))[:4]
    assert float(rel.max()) < 0.04 and float(rel.min()) > 0.01         # 3 mantissa bits: ~2.5 % per element
    assert float((y - x).abs().max() / x.abs().max()) <= 2.0 ** -4
    # the emulation object: weight cache, activation rounding to bf16 first
    emu = O.Fp8Emu(decode=True, gemm=True)
    w = {"a.weight": torch.randn(8, 32, generator=g)}
    assert emu.weight(w, "a.weight") is emu.weight(w, "a.weight")
    a = torch.randn(3, 32, generator=g)
    assert torch.equal(emu.act(a), O.qdq_e4m3_rows(a.bfloat16().float()))
    assert torch.equal(emu.lin(w, "a.weight", a, None, "decode"), a @ emu.weight(w, "a.weight").t())
    assert torch.equal(emu.lin(w, "a.weight", a, None, "prefill"), emu.act(a) @ emu.weight(w, "a.weight").t())
    off = O.Fp8Emu()
    assert torch.equal(off.lin(w, "a.weight", a, None, "prefill"), a @ w["a.weight"].t())
    # end to end on the tiny episode's first turns: neutral emulation == plain oracle bit for bit; the modes change the hidden rows
    sc_t = dict(SCENARIOS["tiny_episode"], eos_mod=0)
    cfg = sc_t["cfg"]
    sd = W.synth_state_dict(cfg, SEED)
    pre = lambda rgb: torch.from_numpy(O.siglip_preprocess(rgb))
    runs = {}
    for key, fp8 in (("plain", None), ("neutral", O.Fp8Emu()), ("decode", O.Fp8Emu(decode=True)), ("gemm", O.Fp8Emu(gemm=True))):
        orc = O.OracleStreamVLN(cfg, sd, num_history=sc_t["num_history"], fp8=fp8)
        log = run_scenario(orc, sc_t, preprocess=pre, steps=5)
        runs[key] = [(r["out"].sequences[0].tolist(), r["out"].hidden.numpy()) for r in log]
        assert len(orc.layer_taps) == cfg.layers
    assert runs["plain"][0][0] == runs["neutral"][0][0] and np.array_equal(runs["plain"][0][1], runs["neutral"][0][1])
    h0 = runs["plain"][0][1]
    assert np.array_equal(runs["decode"][0][1][0], h0[0])              # weight-only decode mode: the prefill row is untouched ...
    d = np.linalg.norm(runs["decode"][0][1][1] - h0[1]) / np.linalg.norm(h0[1])
    assert (0.0 < d < 0.3) or runs["decode"][0][0][0] != runs["plain"][0][0][0]     # ... the decode rows move (unless the e4m3 lm_head already changed the fed token)
    gmm = np.linalg.norm(runs["gemm"][0][1][0] - h0[0]) / np.linalg.norm(h0[0])
    assert 0.005 < gmm < 0.3, gmm


def test_teacher_tokens_mode_keeps_the_models_own_picks():
    """OracleStreamVLN.teacher_tokens (test mode): a turn decodes the given tokens; with the model's own tokens nothing changes, with a
    changed second token the first row, the first pick and its margin stay and the sequence is the forced one."""
    sc = SCENARIOS["tiny_episode"]
    cfg = sc["cfg"]
    pre = lambda rgb: torch.from_numpy(O.siglip_preprocess(rgb))
    sd = W.synth_state_dict(cfg, SEED)
    free = run_scenario(O.OracleStreamVLN(cfg, sd, num_history=sc["num_history"]), sc, preprocess=pre, steps=1)[0]["out"]
    ids = free.sequences[0].tolist()
    assert len(ids) >= 2
    orc = O.OracleStreamVLN(cfg, sd, num_history=sc["num_history"])
    orc.teacher_tokens = [list(ids)]
    same = run_scenario(orc, sc, preprocess=pre, steps=1)[0]["out"]
    assert same.sequences[0].tolist() == ids and orc.own_picks == ids and orc.teacher_tokens == []
    assert torch.equal(same.hidden, free.hidden) and same.margins == free.margins and same.cache_len == free.cache_len
    forced = [ids[0], (ids[1] + 1) % cfg.vocab] + ids[2:]
    orc = O.OracleStreamVLN(cfg, sd, num_history=sc["num_history"])
    orc.teacher_tokens = [forced]
    other = run_scenario(orc, sc, preprocess=pre, steps=1)[0]["out"]
    assert other.sequences[0].tolist() == forced and orc.own_picks[:2] == ids[:2]
    assert torch.equal(other.hidden[:2], free.hidden[:2]) and other.margins[:2] == free.margins[:2]
    assert len(forced) < 3 or not torch.equal(other.hidden[2], free.hidden[2])
