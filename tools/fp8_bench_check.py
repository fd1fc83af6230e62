"""Diagnostic (round 4): bench.py's fp8 `vs_bf16` comparison step by step at the benchmarked size -- which turn / row / mode produces the
relative L2 error the round-3 bench line reported (1.2472)."""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench                                                  # noqa: E402
from streamvln_amd.config import CONFIGS                      # noqa: E402
from streamvln_amd.model import StreamVLNForCausalLM          # noqa: E402

cfg = CONFIGS["streamvln_qwen2_7b"]
model = StreamVLNForCausalLM(cfg, dtype=torch.bfloat16, device=0, max_envs=1, max_frames=9)
model.load_synthetic(1234)
model.model.num_history = 8
model.set_decode_graph(True)
run = bench.Runner(model, cfg, 0)


def short_episode(n=3):
    run.agent.reset_memory(); run.step = 0
    if os.environ.get('SAME_PROMPTS', '1') == '1':
        run.agent.prompt_encoder.reset()
    out = []
    for _ in range(n):
        run.turn()
        out.append((run.agent.turn_log[-1]["out"].sequences[0].tolist(), model.last_hidden()))
    return out


def rel(a, b):
    return float(np.linalg.norm(a - b) / np.linalg.norm(b))


ref = short_episode()
ref2 = short_episode()
print("bf16 twice: ids equal", [a[0] == b[0] for a, b in zip(ref, ref2)], "hidden equal", [np.array_equal(a[1], b[1]) for a, b in zip(ref, ref2)])
for name, dec, gemm in (("decode", 1, 0), ("gemm", 0, 1), ("both", 1, 1)):
    model.set_fp8_decode(bool(dec)); model.set_fp8_gemm(bool(gemm))
    got = short_episode()
    model.set_fp8_decode(False); model.set_fp8_gemm(False)
    for t, ((ia, ha), (ib, hb)) in enumerate(zip(ref, got)):
        print(name, "turn", t, "ids bf16", ia, "fp8", ib, "rel per row", [round(rel(hb[j], ha[j]), 4) for j in range(min(len(ha), len(hb)))],
              "|h| bf16", [round(float(np.linalg.norm(ha[j])), 2) for j in range(len(ha))])
model.close()
