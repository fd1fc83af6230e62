"""CPU: host-side logic (agent turn protocol, splice descriptors, weights generator, image processor, C ABI surface)."""
import os
import re
import subprocess

import numpy as np
import pytest
import torch

from streamvln_amd import _lib, weights as W
from streamvln_amd.agent import StreamingAgent, parse_actions
from streamvln_amd.config import TINY, TRUE, IMAGE_TOKEN_INDEX, MEMORY_TOKEN_INDEX
from streamvln_amd.model import SigLipImageProcessor
from streamvln_amd.synthetic import SyntheticPromptEncoder, synthetic_frame
from oracle import streamvln_oracle as O

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


class FakeModel:
    """records generate calls; returns 2 tokens per turn"""

    def __init__(self):
        self.calls, self.resets = [], 0

    def reset_for_env(self, i):
        self.resets += 1

    def generate(self, **kw):
        self.calls.append(kw)
        n = len(self.calls)
        out = type("O", (), {})()
        out.sequences = torch.tensor([[100 + n, 200 + n]])
        out.past_key_values = ("kv", n)
        return out


def _agent(model, **kw):
    return StreamingAgent(model, SyntheticPromptEncoder(TINY, first_len=20, memory_len=26, later_len=8),
                          preprocess=lambda rgb: torch.zeros(3, 4, 4) + float(rgb), **kw)


def test_turn_protocol_matches_reference_loop():
    """streamvln_eval.py:290-350 with num_frames 8, num_future_steps 4, num_history 2"""
    m = FakeModel()
    ag = _agent(m, num_frames=8, num_future_steps=4, num_history=2)
    for step in range(20):
        ag.act(step)
    c = m.calls
    assert [k["time_ids"][0][0] for k in c] == [0, 0, 8, 8, 16]                 # window starts
    assert [k["images"].shape[1] for k in c] == [1, 1, 3, 1, 3]                  # 2 history frames + current at restarts
    assert [k["past_key_values"] for k in c] == [None, ("kv", 1), None, ("kv", 3), None]
    # first turn: new prompt only; later turn: previous output ids prepended (streamvln_eval.py:305-306)
    assert c[0]["inputs"].shape[1] == 20 and c[1]["inputs"].shape[1] == 2 + 8
    assert c[1]["inputs"][0, :2].tolist() == [101, 201]
    assert c[2]["inputs"].shape[1] == 26 and (c[2]["inputs"] == MEMORY_TOKEN_INDEX).sum() == 1
    assert (c[0]["inputs"] == MEMORY_TOKEN_INDEX).sum() == 0 and (c[0]["inputs"] == IMAGE_TOKEN_INDEX).sum() == 1
    # history frames = rgb_list[0 : t0 : t0 // num_history]  -> steps 0 and 4 at t0 = 8; 0 and 8 at t0 = 16
    assert c[2]["images"][0, :, 0, 0, 0].tolist() == [0.0, 4.0, 8.0]
    assert c[4]["images"][0, :, 0, 0, 0].tolist() == [0.0, 8.0, 16.0]
    assert m.resets == 1 + 2                                                       # constructor + two window resets
    assert c[0]["do_sample"] is False and c[0]["num_beams"] == 1 and c[0]["use_cache"] is True


def test_agent_keeps_ids_on_the_host_unless_asked():
    """The engine reads token ids from the host: the agent does not send them through `device` (the reference callers do,
    streamvln_eval.py:326; the model accepts both).  `ids_device=None` restores the reference placement."""
    m = FakeModel()
    ag = _agent(m, num_frames=8, num_future_steps=4, num_history=2, device="meta")
    ag.observe(0)
    req = ag._build_request("")
    assert req["inputs"].device.type == "cpu" and req["images"].device.type == "meta"
    ag2 = _agent(m, num_frames=8, num_future_steps=4, num_history=2, device="meta", ids_device=None)
    ag2.observe(0)
    assert ag2._build_request("")["inputs"].device.type == "meta"


def test_step_flavour_window_reset():
    """streamvln_agent.py:169-258: run_model=False steps only record; reset when (step_id+1) % num_frames == 0"""
    m = FakeModel()
    ag = _agent(m, num_frames=4, num_future_steps=4, num_history=2)
    acts, _, _ = ag.step(0, 1, "go", run_model=True)
    assert acts == [1, 1, 1, 1] and len(m.calls) == 1
    for s in (1, 2, 3):
        ag.step_id = s
        assert ag.step(0, 1, "go", run_model=False) == (None, 0, None)
    assert ag.output_ids is None and ag.past_key_values is None and ag.time_ids == []
    ag.step_id = 4
    ag.step(0, 2, "go", run_model=True)
    assert m.calls[1]["images"].shape[1] == 3 and m.calls[1]["time_ids"] == [[4]]


def test_parse_actions():
    assert parse_actions("↑↑←→STOP") == [1, 1, 2, 3, 0]
    assert parse_actions("nothing here") == []


def test_weight_generator_is_pure_and_matches_spec():
    assert W.fnv1a64("") == 0xCBF29CE484222325 and W.fnv1a64("a") == 0xAF63DC4C8601EC8C
    spec = [s for s in W.tensor_specs(TINY) if s.name == "model.layers.1.mlp.up_proj.weight"][0]
    a = W.synth_tensor(spec, 1234)
    seed_t = W.tensor_seed(1234, spec.name)
    part = W.round_to_bf16(W.synth_flat(seed_t, 1000, 50, spec.half_width, spec.base))
    assert np.array_equal(a.reshape(-1)[1000:1050], part)
    # scalar re-derivation of one element
    idx = 12345
    z = W.splitmix64_int(seed_t + idx * 0x9E3779B97F4A7C15)
    v = (np.float32(z >> 40) - np.float32(8388608.0)) * (np.float32(spec.half_width) / np.float32(8388608.0))
    assert a.reshape(-1)[idx] == W.round_to_bf16(np.array([v], dtype=np.float32))[0]
    assert np.abs(a).max() <= spec.half_width * 1.004 and abs(float(a.mean())) < 0.01      # bf16 rounding can exceed a by 2^-9
    assert np.array_equal(W.round_to_bf16(a), a)                                  # values are bf16-representable
    n_true = sum(s.numel for s in W.tensor_specs(TRUE))
    assert 7.9e9 < n_true < 8.1e9                                                  # 7.07 B LLM + 0.55 B embed + ViT + projector
    gains = [s for s in W.tensor_specs(TINY) if s.name.endswith("layernorm.weight")]
    assert all(s.base == 1.0 for s in gains)


def test_image_processor_equals_oracle_preprocess():
    frame = synthetic_frame(0, 3)
    proc = SigLipImageProcessor()
    a = proc.preprocess_array(frame).numpy()
    assert np.array_equal(a, O.siglip_preprocess(frame))
    from PIL import Image
    pv = proc.preprocess(images=Image.fromarray(frame), return_tensors="pt")["pixel_values"][0]
    assert np.array_equal(pv.numpy(), a) and proc.crop_size == {"height": 384, "width": 384}


def test_library_loads_and_exports_every_declared_symbol():
    lib = _lib.load()
    header = open(os.path.join(ROOT, "include", "streamvln_hip.h")).read()
    declared = set(re.findall(r"\b(svln_[a-z0-9_]+)\s*\(", header))
    assert declared == set(_lib.SIGNATURES), declared ^ set(_lib.SIGNATURES)
    nm = subprocess.run(["nm", "-D", "--defined-only", _lib.LIB_PATH], capture_output=True, text=True, check=True).stdout
    exported = set(re.findall(r" T (svln_[a-z0-9_]+)", nm))
    assert declared <= exported, declared - exported
    for name in declared:
        assert getattr(lib, name) is not None


def test_engine_fails_loudly_without_a_gpu():
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    from streamvln_amd.model import StreamVLNForCausalLM
    with pytest.raises(_lib.SvlnError):
        StreamVLNForCausalLM(TINY)


class StubTokenizer:
    """whitespace tokenizer with the methods preprocess_qwen uses"""

    def __init__(self):
        self.vocab, self.chat_template = {}, None

    def _id(self, w):
        return self.vocab.setdefault(w, 100 + len(self.vocab))

    def add_tokens(self, toks, special_tokens=False):
        for t in toks:
            self._id(t)

    def convert_tokens_to_ids(self, t):
        return self._id(t)

    def apply_chat_template(self, msgs):
        out = []
        for m in msgs:
            text = "<|im_start|> " + m["role"] + " \n " + m["content"].replace("<image>", " <image> ").replace("<memory>", " <memory> ") + " <|im_end|> \n"
            out += [self._id(w) for w in text.split(" ") if w]
        return out


def test_qwen_prompt_encoder_follows_preprocess_qwen():
    from streamvln_amd.prompt import QwenPromptEncoder
    tok = StubTokenizer()
    enc = QwenPromptEncoder(tok, flavour="agent")
    inv = lambda ids: [k for t in ids for k, v in enc.tok.vocab.items() if v == t]
    first = enc(True, False, "walk to the kitchen")
    assert first.count(IMAGE_TOKEN_INDEX) == 1 and first.count(MEMORY_TOKEN_INDEX) == 0
    words = inv([t for t in first if t >= 0])
    assert words[:3] == ["<|im_start|>", "system", "\n"] and "kitchen" in words and "<video>" not in words
    assert first[-8:].count(IMAGE_TOKEN_INDEX) == 0 and words[-5:] == ["<|im_start|>", "assistant", "\n", "<|im_end|>", "\n"]
    mem = enc(True, True, "walk")
    assert mem.count(MEMORY_TOKEN_INDEX) == 1 and mem.index(MEMORY_TOKEN_INDEX) < mem.index(IMAGE_TOKEN_INDEX)
    assert "visited" in inv([t for t in mem if t >= 0])
    later = enc(False, False)
    assert later.count(IMAGE_TOKEN_INDEX) == 1 and inv([t for t in later if t >= 0])[:2] == ["<|im_start|>", "user"]   # no system turn
    ev = QwenPromptEncoder(tok, flavour="eval")
    ev_words = [k for t in ev(True, True, "x") if t >= 0 for k, v in ev.tok.vocab.items() if v == t]
    assert "historical" in ev_words and "visited" not in ev_words
    ag = StreamingAgent(FakeModel(), enc, num_frames=8, preprocess=lambda rgb: torch.zeros(3, 4, 4))
    ag.act(0, "go left")
    assert ag.model.calls[0]["inputs"].shape[1] == len(enc(True, False, "go left"))


def test_eval_flavour_prompts_equal_the_reference_eval_loop():
    """tests/golden/eval_prompts.npz = what the REFERENCE's Habitat loop assembles and tokenises for the three situations of a window
    (oracle/make_eval_prompts.py: the `if output_ids is None` statement of `VLNEvaluator.eval_action`, streamvln_eval.py:291-302, executed
    as it stands, then the reference's `preprocess_qwen` :393-469 under `random.seed(k)` with the stub tokenizer).
    QwenPromptEncoder(flavour="eval") must produce the same text and the same ids -- including WHICH conjunction `random.choice` picks
    for a given seed, the "These are your historical observations <memory>." sentence and the system turn only when the reference adds it."""
    import random
    from stub_tokenizer import StubTokenizer                   # the deterministic tokenizer the fixture was generated with
    from streamvln_amd.agent import CONJUNCTIONS
    from streamvln_amd.prompt import QwenPromptEncoder
    g = np.load(os.path.join(os.path.dirname(__file__), "golden", "eval_prompts.npz"))
    assert list(g["conjunctions"]) == list(CONJUNCTIONS)
    instruction = str(g["instruction"])
    picked = set()
    for name, first_turn, with_memory in (("first", True, False), ("memory", True, True), ("later", False, False)):
        for seed in g["seeds"].tolist():
            enc = QwenPromptEncoder(StubTokenizer(), flavour="eval", rng=random.Random(seed))
            key = f"{name}_s{seed}"
            gold = g[key + "_ids"].tolist()
            assert bool(g[key + "_add_system"]) == first_turn
            if first_turn:
                assert enc.first_turn_text(with_memory, instruction) == str(g[key + "_text"])
            else:
                assert str(g[key + "_text"]) == ""
            ids = enc(first_turn, with_memory, instruction)
            assert ids == gold, (key, ids, gold)
            assert gold.count(MEMORY_TOKEN_INDEX) == int(with_memory) and gold.count(IMAGE_TOKEN_INDEX) == 1
            picked.add(tuple(gold[-12:]))
    assert len(picked) >= 3                                    # the seeds really select different conjunctions


def test_config_from_hf_checkpoint_config():
    """config.json / AutoConfig of a StreamVLN checkpoint -> StreamVLNConfig (streamvln_eval.py:521-527); unsupported structure raises."""
    from types import SimpleNamespace
    from streamvln_amd.config import TRUE, config_from_hf
    qwen7b = dict(hidden_size=3584, num_hidden_layers=28, num_attention_heads=28, num_key_value_heads=4, intermediate_size=18944,
                  vocab_size=152064, rope_theta=1000000.0, rms_norm_eps=1e-06, rope_scaling=None, mm_spatial_pool_mode="bilinear",
                  mm_projector_type="mlp2x_gelu", use_sliding_window=False)
    c = config_from_hf(qwen7b)
    assert c.to_dict() == dict(TRUE.to_dict(), name="from_checkpoint")
    c2 = config_from_hf(SimpleNamespace(**dict(qwen7b, num_hidden_layers=2, vocab_size=4096)))     # attribute-style (AutoConfig)
    assert (c2.layers, c2.vocab, c2.hidden) == (2, 4096, 3584)
    for bad in (dict(num_attention_heads=56), dict(rope_scaling={"type": "linear", "factor": 2.0}), dict(mm_spatial_pool_mode="average"),
                dict(mm_projector_type="linear"), dict(use_sliding_window=True), dict(num_key_value_heads=3)):
        with pytest.raises(ValueError):
            config_from_hf(dict(qwen7b, **bad))


def test_agent_reproduces_the_reference_callers_generate_calls():
    """tests/golden/agent_calls.npz = every generate(**kwargs) the REFERENCE's VLNEvaluator.step issued over 44 env steps driven as
    its HTTP server drives it (oracle/make_agent_calls.py; streamvln_agent.py:169-258, http_realworld_server.py:95-112).
    StreamingAgent.step + QwenPromptEncoder(flavour="agent") must issue the same calls: ids incl. sentinels, views (which frames,
    by pixel sum), time_ids, past_key_values None / not, flags, reset points and returned action sequences."""
    from stub_tokenizer import RecordingModel, StubTokenizer
    from streamvln_amd.prompt import QwenPromptEncoder
    g = np.load(os.path.join(ROOT, "tests", "golden", "agent_calls.npz"))
    num_frames, nfs, nh = (int(v) for v in g["config"])
    tok = StubTokenizer()
    model = RecordingModel(SigLipImageProcessor())
    enc = QwenPromptEncoder(tok, flavour="agent")
    ag = StreamingAgent(model, enc, num_frames=num_frames, num_future_steps=nfs, num_history=nh, image_dtype=torch.bfloat16,
                        decode_actions=lambda seq: parse_actions(tok.batch_decode(seq)[0].strip()))
    n_resets0 = len(model.resets)                       # the constructor's reset_memory (the reference does not reset at construction)
    returned = []
    instruction = str(g["instruction"])
    for s in range(int(g["n_steps"])):
        acts, _, _ = ag.step(0, synthetic_frame(0, s), instruction, run_model=(ag.step_id % 4 == 0))
        returned.append([] if acts is None else list(acts))
        ag.step_id += 1
    assert len(model.calls) == int(g["n_calls"])
    for k, c in enumerate(model.calls):
        for name, v in c.items():
            exp = g[f"c{k}_{name}"]
            assert np.array_equal(np.asarray(v), exp), (k, name, v, exp)
    assert np.array_equal(np.asarray(model.resets[n_resets0:], dtype=np.int64).reshape(-1, 2), g["resets"])
    assert [len(r) for r in returned] == g["returned_len"].tolist()
    assert [a for r in returned for a in r] == g["returned_flat"].tolist()
    assert (g["c8_inputs"] == MEMORY_TOKEN_INDEX).sum() == 1 and int(g["c8_views"]) == 1 + nh     # the window-restart call is in the fixture


def test_run_sharded_resumes_by_skipping_finished_episodes(tmp_path):
    """resume contract of streamvln_eval.py:203-224,365-377: result.json is append-only; a restarted run skips the
    (scene, episode, instruction) triples already present, rank 0 reloads their metrics, the summary covers all episodes."""
    import json
    from types import SimpleNamespace
    from streamvln_amd.eval_harness import load_done, run_sharded
    eps = {sc: [SimpleNamespace(episode_id=f"{sc}-{i}", instruction_text=f"go {i}") for i in range(3)] for sc in ("b_scene", "a_scene")}
    path = str(tmp_path / "out" / "result.json")
    ran = []

    class Interrupted(RuntimeError):
        pass

    def run_episode(scene, ep, limit=[4]):
        if len(ran) >= limit[0]:
            raise Interrupted()
        ran.append(ep.episode_id)
        i = int(ep.episode_id[-1])
        return {"success": float(i % 2), "spl": 0.25 * i, "os": 1.0, "ne": 1.0 + i, "steps": 10 + i}

    with pytest.raises(Interrupted):                                   # first run dies after 4 of 6 episodes
        run_sharded(eps, run_episode, result_path=path)
    done, metrics = load_done(path)
    assert [d[1] for d in done] == ["a_scene-0", "a_scene-1", "a_scene-2", "b_scene-0"] and len(metrics) == 4     # scenes in sorted order
    first = list(ran)
    ran.clear()
    summary = run_sharded(eps, lambda s, e: run_episode(s, e, limit=[99]), result_path=path)
    assert ran == ["b_scene-1", "b_scene-2"] and not set(ran) & set(first)                                        # only the unfinished ones ran
    assert summary["length"] == 6 and abs(summary["sucs_all"] - 2 / 6) < 1e-12 and abs(summary["ones_all"] - 2.0) < 1e-12
    lines = [json.loads(l) for l in open(path)]
    assert len(lines) == 7 and "scene_id" not in lines[-1] and lines[-1]["length"] == 6                           # 6 episode records + summary
    ran.clear()
    again = run_sharded(eps, lambda s, e: run_episode(s, e, limit=[99]), result_path=path)                         # nothing left to do
    assert ran == [] and again["length"] == 6


def test_oracle_is_test_infrastructure_only():
    """`oracle/` may be imported by tests/, by `__graft_entry__.smoke()` and by bench.py's `cpu_baseline*` functions -- nowhere in the product
    package, not in the timed part of bench.py, not in tools/ (checked on the syntax tree: an import inside any other function or at module
    level fails)."""
    import ast
    import glob
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

    def oracle_imports(path):
        tree = ast.parse(open(path).read(), path)
        hits = []

        def visit(node, fn):
            for child in ast.iter_child_nodes(node):
                name = fn
                if isinstance(child, (ast.FunctionDef, ast.AsyncFunctionDef)):
                    name = child.name if fn is None else fn            # the outermost enclosing function
                if isinstance(child, ast.Import) and any(a.name.split(".")[0] == "oracle" for a in child.names):
                    hits.append(fn)
                if isinstance(child, ast.ImportFrom) and (child.module or "").split(".")[0] == "oracle":
                    hits.append(fn)
                visit(child, name)
        visit(tree, None)
        return hits

    for path in glob.glob(os.path.join(root, "streamvln_amd", "**", "*.py"), recursive=True) + glob.glob(os.path.join(root, "tools", "*.py")):
        assert oracle_imports(path) == [], path
    assert all(fn and fn.startswith("cpu_baseline") for fn in oracle_imports(os.path.join(root, "bench.py")))
    assert all(fn == "smoke" for fn in oracle_imports(os.path.join(root, "__graft_entry__.py")))
    assert oracle_imports(os.path.join(root, "bench.py")) and oracle_imports(os.path.join(root, "__graft_entry__.py"))
