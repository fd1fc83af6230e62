"""Pin the Habitat-loop prompt flavour with the REFERENCE's own code (missing #5).

`streamvln/streamvln_eval.py` builds the prompt of a turn inside `VLNEvaluator.eval_action` (:290-304: the conversation template, the
"These are your historical observations <memory>." sentence, the instruction) and tokenises it in `VLNEvaluator.preprocess_qwen`
(:393-469: chat template, `random.choice(self.conjunctions) + <image>`, sentinel ids).  The module imports the Habitat simulator at the
top and the constructor needs a Habitat config, neither of which exists in this image and neither of which the prompt code touches, so:

  * the absent modules are pre-registered as empty stubs (as `qformer` / `quaternion` are for the other fixtures) and the module is imported;
  * an instance is made WITHOUT the constructor; `conversation` and `conjunctions` are read from the constructor's own assignments
    (parsed from the reference source at generation time, literal values only);
  * the `if output_ids is None: ... else: ...` statement of `eval_action` is compiled from the reference's syntax tree and executed as it
    stands for the three situations of a window (first turn, first turn of a later window with <memory>, later turn), then the
    reference's `preprocess_qwen` is called with a deterministic stub tokenizer under `random.seed(k)`.

Written to tests/golden/eval_prompts.npz: per case the text the reference assembled, `add_system`, and the ids (with -200 / -300).  Data only.
TEST INFRASTRUCTURE ONLY.  Run here (needs /root/reference):  python -m oracle.make_eval_prompts
"""
from __future__ import annotations

import ast
import contextlib
import copy
import io
import os
import random
import sys
import types

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

from oracle import ref_harness as RH                           # noqa: E402
from oracle.make_agent_calls import import_evaluator           # noqa: E402  (registers the quaternion / omegaconf / depth stubs)
from stub_tokenizer import StubTokenizer                        # noqa: E402

REF_EVAL = "/root/reference/streamvln/streamvln_eval.py"
INSTRUCTION = "walk past the sofa and stop at the door"
SEEDS = (0, 1, 2, 3, 4, 5, 6)
# (name, step_id, a previous turn's output exists)
CASES = (("first", 0, False), ("memory", 32, False), ("later", 4, True))


class _Anything(types.ModuleType):
    """a stub module whose every attribute is another stub (the simulator names the eval module imports and the prompt code never uses)"""
    def __getattr__(self, name):
        if name.startswith("__"):
            raise AttributeError(name)
        return type(name, (), {})


def import_eval_module():
    import_evaluator()                                         # reference on sys.path + the stubs the agent fixture already uses
    for name in ("habitat", "habitat.config", "habitat.config.default", "habitat.config.default_structured_configs", "habitat.utils",
                 "habitat.utils.visualizations", "habitat.utils.visualizations.utils", "habitat_extensions", "habitat_baselines",
                 "habitat_baselines.config", "habitat_baselines.config.default"):
        if name not in sys.modules:
            sys.modules[name] = _Anything(name)
    import streamvln_eval                                      # /root/reference/streamvln/streamvln_eval.py
    return streamvln_eval


def constructor_literals(tree):
    """`self.conjunctions = [...]` and the `prompt = f"..."` behind `self.conversation` from VLNEvaluator.__init__ (values only)"""
    cls = next(n for n in tree.body if isinstance(n, ast.ClassDef) and n.name == "VLNEvaluator")
    init = next(n for n in cls.body if isinstance(n, ast.FunctionDef) and n.name == "__init__")
    conj = prompt = None
    for node in ast.walk(init):
        if isinstance(node, ast.Assign) and len(node.targets) == 1:
            t = node.targets[0]
            if isinstance(t, ast.Attribute) and t.attr == "conjunctions":
                conj = ast.literal_eval(node.value)
            if isinstance(t, ast.Name) and t.id == "prompt":
                v = node.value
                assert isinstance(v, ast.JoinedStr) and all(isinstance(p, ast.Constant) for p in v.values), "prompt has placeholders"
                prompt = "".join(p.value for p in v.values)
    assert conj and prompt
    return conj, prompt


def prompt_statement(tree):
    """the `if output_ids is None: ... else: ...` statement of eval_action (streamvln_eval.py:291-304), compiled as it stands"""
    cls = next(n for n in tree.body if isinstance(n, ast.ClassDef) and n.name == "VLNEvaluator")
    found = []
    for fn in cls.body:
        if isinstance(fn, ast.FunctionDef):
            for node in ast.walk(fn):
                if isinstance(node, ast.If) and isinstance(node.test, ast.Compare) and isinstance(node.test.left, ast.Name) and \
                        node.test.left.id == "output_ids" and isinstance(node.test.ops[0], ast.Is) and node.orelse and \
                        any(isinstance(s, ast.Assign) and getattr(s.targets[0], "id", "") == "add_system" for s in node.body):
                    found.append(node)
    assert len(found) == 1, len(found)
    mod = ast.Module(body=[found[0]], type_ignores=[])
    ast.fix_missing_locations(mod)
    return compile(mod, REF_EVAL, "exec"), (found[0].lineno, found[0].end_lineno)


def main():
    ev_mod = import_eval_module()
    tree = ast.parse(open(REF_EVAL).read())
    conjunctions, prompt = constructor_literals(tree)
    code, span = prompt_statement(tree)
    VE = ev_mod.VLNEvaluator
    ev = VE.__new__(VE)                                        # no Habitat config: the constructor is not run
    ev.conjunctions = conjunctions
    ev.conversation = [{"from": "human", "value": prompt}, {"from": "gpt", "value": ""}]       # :103-105
    episode = types.SimpleNamespace(instruction=types.SimpleNamespace(instruction_text=INSTRUCTION))
    fx = {"instruction": np.asarray(INSTRUCTION), "seeds": np.asarray(SEEDS, dtype=np.int64), "conjunctions": np.asarray(conjunctions),
          "statement_lines": np.asarray(span, dtype=np.int64)}
    for name, step_id, has_prev in CASES:
        for seed in SEEDS:
            ns = {"copy": copy, "self": ev, "step_id": step_id, "episode": episode, "output_ids": (object() if has_prev else None)}
            ns.update({k: getattr(ev_mod, k) for k in ("DEFAULT_MEMORY_TOKEN", "DEFAULT_VIDEO_TOKEN", "DEFAULT_IMAGE_TOKEN")})
            with contextlib.redirect_stdout(io.StringIO()):
                exec(code, ns)                                 # the reference's own statement: builds `sources`, `add_system`
                text = ns["sources"][0]["value"]
                random.seed(seed)
                ids, _ = VE.preprocess_qwen(ev, [ns["sources"]], StubTokenizer(), True, add_system=ns["add_system"])
            key = f"{name}_s{seed}"
            fx[key + "_text"] = np.asarray(text)
            fx[key + "_add_system"] = np.int64(bool(ns["add_system"]))
            fx[key + "_ids"] = ids[0].numpy().astype(np.int64)
        print(f"{name}: add_system {bool(ns['add_system'])}, {ids.shape[1]} ids, text {text[:60]!r}...")
    out = os.path.join(ROOT, "tests", "golden", "eval_prompts.npz")
    np.savez_compressed(out, **fx)
    print(f"wrote {out}: {len(CASES) * len(SEEDS)} prompts from streamvln_eval.py:{span[0]}-{span[1]} + preprocess_qwen")


if __name__ == "__main__":
    main()
