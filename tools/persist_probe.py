"""Phase timeline of ONE persistent decode-layer launch at the benchmarked size (svln_probe_decode_layer): where the launch's time goes,
per workgroup, in microseconds since the earliest stamp."""
import ctypes as C
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench                                                  # noqa: E402
from streamvln_amd import _lib                                # noqa: E402
from streamvln_amd.config import CONFIGS                      # noqa: E402
from streamvln_amd.model import StreamVLNForCausalLM          # noqa: E402

cfg = CONFIGS["streamvln_qwen2_7b"]
model = StreamVLNForCausalLM(cfg, dtype=torch.bfloat16, device=0, max_envs=1, max_frames=9)
model.load_synthetic(1234)
model.model.num_history = 8
model.set_decode_graph(True)
model.set_decode_persistent(True)
run = bench.Runner(model, cfg, 0)
for _ in range(4):
    run.turn()
names = ["merge", "edge0", "o_proj", "edge1", "gate_up", "edge2", "down", "edge3", "qkv", "-", "L.start", "L.o", "L.gu", "L.down", "L.end"]
for rep in range(3):
    out = np.zeros((256, 16), dtype=np.uint64)
    n = C.c_int32()
    _lib.check(model._lib.svln_probe_decode_layer(model._h, 5, out.ctypes.data_as(C.POINTER(C.c_uint64)), 256, C.byref(n)))
    t = out[: n.value].astype(np.float64)
    t0 = t[t > 0].min()
    us = (t - t0) / 100.0
    print(f"rep {rep}: per stamp, microseconds since the first stamp: median over workgroups [min .. max]")
    for k, nm in enumerate(names):
        if nm == "-":
            continue
        col = us[:, k]
        print(f"  {nm:8s} {np.median(col):8.2f}  [{col.min():8.2f} .. {col.max():8.2f}]")
model.close()
