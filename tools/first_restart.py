"""Where the first window-restart turn of a process spends more time than the second (run on the GPU box)."""
import collections, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ctypes as C
import torch
import bench
from streamvln_amd.config import CONFIGS
from streamvln_amd.model import StreamVLNForCausalLM

cfg = CONFIGS["streamvln_qwen2_7b"]
model = StreamVLNForCausalLM(cfg, dtype=torch.bfloat16, device=0, max_envs=1, max_frames=9)
model.load_synthetic(1234); model.model.num_history = 8; model.set_decode_graph(True)
run = bench.Runner(model, cfg, 0)
acc = collections.defaultdict(float)

def wrap(obj, name, key):
    f = getattr(obj, name)
    def g(*a, **k):
        t = time.perf_counter(); r = f(*a, **k); acc[key] += time.perf_counter() - t; return r
    setattr(obj, name, g)

class LibProxy:
    def __init__(self, lib): self._lib = lib
    def __getattr__(self, n):
        f = getattr(self._lib, n)
        def g(*a):
            t = time.perf_counter(); r = f(*a); acc["lib." + n] += time.perf_counter() - t; return r
        return g
lib0, h = model._lib, model._h
model._lib = LibProxy(model._lib)
model.get_vision_tower().image_processor._engine = (model._lib, model._h)
wrap(run.agent, "_build_request", "agent._build_request")
wrap(model, "_parse_call", "model._parse_call")
wrap(model, "generate", "model.generate")
d3 = [C.c_double() for _ in range(3)]
for i in range(40):
    acc.clear()
    lib0.svln_phase_times(h, C.byref(d3[0]), C.byref(d3[1]), C.byref(d3[2]), 1)
    t = time.perf_counter(); run.turn(); model.sync(); w = (time.perf_counter() - t) * 1e3
    lib0.svln_phase_times(h, C.byref(d3[0]), C.byref(d3[1]), C.byref(d3[2]), 0)
    if w > 24 or i in (7,):
        print(f"turn {i}: wall {w:.2f} ms  gpu phases v/p/d {d3[0].value:.2f} {d3[1].value:.2f} {d3[2].value:.2f}  " +
              "  ".join(f"{k} {v * 1e3:.2f}" for k, v in sorted(acc.items(), key=lambda kv: -kv[1])[:7]))
