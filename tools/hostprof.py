"""Where a steady turn's wall time goes on the host side (per-call perf_counter; run on the GPU box).
  python tools/hostprof.py [--pageable-frames] [--own-torch-stream]"""
import collections, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from streamvln_amd.config import CONFIGS
from streamvln_amd.model import StreamVLNForCausalLM

cfg = CONFIGS["streamvln_qwen2_7b"]
model = StreamVLNForCausalLM(cfg, dtype=torch.bfloat16, device=0, max_envs=1, max_frames=9)
model.load_synthetic(1234); model.model.num_history = 8; model.set_decode_graph(True)
run = bench.Runner(model, cfg, 0, frame_ring="--pageable-frames" not in sys.argv)
if "--own-torch-stream" not in sys.argv:
    torch.cuda.set_stream(model.torch_stream)
acc = collections.defaultdict(float)
marks = []

def wrap(obj, name, key):
    f = getattr(obj, name)
    def g(*a, **k):
        t = time.perf_counter(); r = f(*a, **k); t1 = time.perf_counter(); acc[key] += t1 - t; marks.append((key, t, t1)); return r
    setattr(obj, name, g)

class LibProxy:
    def __init__(self, lib): self._lib = lib
    def __getattr__(self, n):
        f = getattr(self._lib, n)
        def g(*a):
            t = time.perf_counter(); r = f(*a); t1 = time.perf_counter(); acc["lib." + n] += t1 - t; marks.append(("lib." + n, t, t1)); return r
        return g
model._lib = LibProxy(model._lib)
model.get_vision_tower().image_processor._engine = (model._lib, model._h)
wrap(run.agent, "_build_request", "agent._build_request")
wrap(run.agent, "_consume", "agent._consume")
wrap(model, "generate", "model.generate")
wrap(model, "_parse_call", "model._parse_call")
wrap(run, "preprocess", "runner.preprocess")
for _ in range(17): run.turn()          # into the second episode: steady turns 1..7
model.sync(); acc.clear(); marks.clear()
N = 6
t0 = time.perf_counter()
for _ in range(N): run.turn()
model.sync()
wall = (time.perf_counter() - t0) / N * 1e3
print(f"steady turn wall {wall:.3f} ms")
for k, v in sorted(acc.items(), key=lambda kv: -kv[1]):
    print(f"  {k:32s} {v / N * 1e3:8.3f} ms/turn")
# timeline of the host between two svln_turn calls (the GPU is idle from the end of one to the first launch of the next)
turns = [m for m in marks if m[0] == "lib.svln_turn"]
gaps = [(b[1] - a[2]) * 1e3 for a, b in zip(turns, turns[1:])]
print("host time between svln_turn return and the next svln_turn entry (GPU idle), ms:", [round(g, 3) for g in gaps])
a, b = turns[1], turns[2]
print("one such gap in detail (us since the return):")
for k, t, t1 in marks:
    if a[2] <= t <= b[1] and k != "lib.svln_turn":
        print(f"   {k:34s} start {1e6 * (t - a[2]):8.1f}  dur {1e6 * (t1 - t):7.1f}")
