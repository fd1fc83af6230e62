"""Pin the caller protocol with the REFERENCE's own caller: drive `VLNEvaluator.step`
(streamvln/streamvln_agent.py:169-258) exactly as its HTTP server does (http_realworld_server.py:95-112:
`run_model = step_id % 4 == 0`, the caller increments `step_id`) for 44 env steps across the window reset, with a
recording stand-in for the model and a deterministic stub tokenizer, and write every `generate(**kwargs)` the
reference issued to tests/golden/agent_calls.npz (ids incl. -200/-300 sentinels, views + per-view pixel sums,
time_ids, past_key_values None / not, flags, reset points, returned action sequences).  Data only.

TEST INFRASTRUCTURE ONLY.  Run here (needs /root/reference):  python -m oracle.make_agent_calls
`streamvln_agent.py` imports three modules this image lacks and the path never touches (`quaternion`, `omegaconf`,
`depth_camera_filtering`): they are pre-registered as empty stubs, like `qformer` in ref_harness.py.  The evaluator's device is
set to "cpu" (`dict_to_cuda` is a plain `.to(device)`, utils/utils.py:161-170).
"""
from __future__ import annotations

import argparse
import contextlib
import io
import os
import sys
import types

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

from oracle import ref_harness as RH                           # noqa: E402
from stub_tokenizer import RecordingModel, StubTokenizer        # noqa: E402
from streamvln_amd.synthetic import synthetic_frame             # noqa: E402

N_STEPS, NUM_FRAMES, NUM_FUTURE, NUM_HISTORY = 44, 32, 4, 8
INSTRUCTION = "walk past the sofa and stop at the door"
INTRINSIC = np.array([[192.0, 0, 191.42857143, 0], [0, 192.0, 191.42857143, 0], [0, 0, 1, 0], [0, 0, 0, 1]])


def import_evaluator():
    RH.import_reference()                                      # sys.path + qformer stub
    for name, attrs in (("quaternion", {}), ("omegaconf", {"OmegaConf": type("OmegaConf", (), {})}),
                        ("depth_camera_filtering", {"filter_depth": lambda *a, **k: None})):
        if name not in sys.modules:
            m = types.ModuleType(name)
            for k, v in attrs.items():
                setattr(m, k, v)
            sys.modules[name] = m
    from streamvln_agent import VLNEvaluator                    # /root/reference/streamvln/streamvln_agent.py
    return VLNEvaluator


def main():
    VLNEvaluator = import_evaluator()
    _, _, _, _, Proc = RH.import_reference()
    model = RecordingModel(Proc())                              # the reference's own SigLipImageProcessor
    args = argparse.Namespace(num_frames=NUM_FRAMES, num_future_steps=NUM_FUTURE, num_history=NUM_HISTORY)
    sink = io.StringIO()
    with contextlib.redirect_stdout(sink):
        ev = VLNEvaluator({"camera_intrinsic": INTRINSIC}, model=model, tokenizer=StubTokenizer(), args=args)
        ev.device = "cpu"
        returned = []
        for s in range(N_STEPS):                                # http_realworld_server.py:95-112
            run_model = ev.step_id % 4 == 0
            acts, _, text = ev.step(0, synthetic_frame(0, s), INSTRUCTION, run_model=run_model)
            returned.append([] if acts is None else list(acts))
            ev.step_id += 1
    fx = {"n_steps": np.int64(N_STEPS), "n_calls": np.int64(len(model.calls)), "instruction": np.asarray(INSTRUCTION),
          "config": np.asarray([NUM_FRAMES, NUM_FUTURE, NUM_HISTORY], dtype=np.int64),
          "resets": np.asarray(model.resets, dtype=np.int64).reshape(-1, 2),
          "returned_len": np.asarray([len(r) for r in returned], dtype=np.int64),
          "returned_flat": np.asarray([a for r in returned for a in r], dtype=np.int64)}
    for k, c in enumerate(model.calls):
        for name, v in c.items():
            fx[f"c{k}_{name}"] = v
        print(f"call {k}: n_inputs {c['inputs'].size} views {int(c['views'])} time_ids[0] {int(c['time_ids'][0])} "
              f"len(time_ids) {c['time_ids'].size} pkv {int(c['pkv'])} sentinels {(c['inputs'] == -300).sum()}x<memory> "
              f"{(c['inputs'] == -200).sum()}x<image>")
    out = os.path.join(ROOT, "tests", "golden", "agent_calls.npz")
    np.savez_compressed(out, **fx)
    print(f"wrote {out}: {len(model.calls)} generate calls, resets at {model.resets}")


if __name__ == "__main__":
    main()
