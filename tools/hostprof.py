"""Where a steady turn's wall time goes on the host side (per-call perf_counter; run on the GPU box)."""
import collections, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from streamvln_amd.config import CONFIGS
from streamvln_amd.model import StreamVLNForCausalLM
import streamvln_amd.agent as A

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
model._lib = LibProxy(model._lib)
model.get_vision_tower().image_processor._engine = (model._lib, model._h)
wrap(run.agent, "_build_request", "agent._build_request")
wrap(run.agent, "_consume", "agent._consume")
wrap(model, "generate", "model.generate")
wrap(model, "_parse_call", "model._parse_call")
wrap(model, "_result", "model._result")
wrap(run, "preprocess", "runner.preprocess")
for _ in range(17): run.turn()          # into the second episode: steady turns 1..7
model.sync(); acc.clear()
N = 6
t0 = time.perf_counter()
for _ in range(N): run.turn()
model.sync()
wall = (time.perf_counter() - t0) / N * 1e3
print(f"steady turn wall {wall:.3f} ms")
for k, v in sorted(acc.items(), key=lambda kv: -kv[1]):
    print(f"  {k:32s} {v / N * 1e3:8.3f} ms/turn")
import ctypes as C
d = [C.c_double() for _ in range(3)]
