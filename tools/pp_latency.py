"""Host latency of the image processor call in different stream states (run on the GPU box)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from streamvln_amd.config import TINY
from streamvln_amd.model import StreamVLNForCausalLM

m = StreamVLNForCausalLM(TINY, dtype=torch.bfloat16, max_envs=1, max_frames=1, max_positions=256)
proc = m.get_vision_tower().image_processor
frame = np.random.default_rng(0).integers(0, 256, (480, 640, 3), dtype=np.uint8)
for _ in range(5): proc.preprocess_array(frame)
torch.cuda.synchronize()

def run(tag, between, n=40):
    ts = []
    for _ in range(n):
        between()
        t = time.perf_counter(); o = proc.preprocess_array(frame); ts.append(time.perf_counter() - t)
    torch.cuda.synchronize()
    ts = np.array(ts) * 1e6
    print(f"{tag:46s} median {np.median(ts):7.1f} us  p90 {np.percentile(ts, 90):7.1f}  max {ts.max():7.1f}")

x = torch.zeros(1 << 20, device="cuda")
run("back to back", lambda: None)
run("device idle (synchronize before)", lambda: torch.cuda.synchronize())
run("sleep 200 us before", lambda: time.sleep(2e-4))
run("a torch kernel pending on the current stream", lambda: x.add_(1.0))
run("torch.stack of the previous outputs pending", lambda: torch.stack([x, x]))
m.close()
