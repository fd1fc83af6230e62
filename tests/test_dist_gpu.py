"""The N-rank code path of bench.py on the GPU box every round (no 8-GPU node is ours to launch): two ranks started by
`torch.distributed.run` as a fresh child process, both on GPU 0, rendezvous over gloo -- the same launch line the driver uses for
N > 1 (RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* from the environment), the same barrier + max-over-ranks timing, the same metric
all-reduce; only the backend (gloo instead of RCCL) and the device mapping (--single-device) differ."""
import json
import os
import socket
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


BENCH_ARGS = ["--gpus", "2", "--steps", "3", "--warmup", "1", "--config", "tiny", "--backend", "gloo", "--single-device", "--no-cpu-baseline",
              "--no-feature-cache-pass", "--no-fp8-pass", "--no-prune-pass", "--batched-envs", "2"]


@pytest.mark.gpu
@pytest.mark.parametrize("launcher", ["torch.distributed.run", "self"])
def test_bench_two_ranks_child_process(launcher):
    """launcher = the driver's line (`python -m torch.distributed.run ... bench.py --gpus 2`), or the plain command `python bench.py --gpus 2`
    with no RANK in the environment, which must start the same launcher itself as a child process (scripts/streamvln_eval_multi_gpu.sh:7)."""
    if launcher == "self":
        cmd = [sys.executable, os.path.join(ROOT, "bench.py")] + BENCH_ARGS
    else:
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
               "--master-port", str(_free_port()), os.path.join(ROOT, "bench.py")] + BENCH_ARGS
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK")}
    env.update(HSA_ENABLE_IPC_MODE_LEGACY="0", MASTER_ADDR="127.0.0.1")
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=900, cwd=ROOT, env=env)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]                       # rank 0 prints ONE JSON line
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["steps"] == 3 and d["warmup"] == 1 and d["scaling"] == "weak"
    assert d["metric_allreduce_check"]["length"] == 2              # one synthetic episode per rank went through the all-reduce
    assert abs(d["metric_allreduce_check"]["ones_all"] - 0.5) < 1e-12      # mean of rank ids 0 and 1
    assert d["value"] > 0 and abs(d["per_gpu"] * 2 - d["value"]) < 0.02 * d["value"]
    assert d["batched_envs"]["envs_per_gpu"] == 2 and d["roofline"]["launches_timed"] > 0
    assert "cpu_baseline" not in d                                 # rank 0 at N = 1 only


@pytest.mark.gpu
def test_bench_rccl_branch_one_rank():
    """RCCL itself (backend "nccl") on the one GPU of the box: a 1-rank process group takes the same branch as the driver's N-rank
    runs -- torch.cuda.set_device(LOCAL_RANK), init_process_group("nccl"), barrier, the max-over-ranks all-reduce and the 5-scalar
    metric all-reduce on CUDA tensors (bench.py: timed_pass / reduce_metrics)."""
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "2", "--warmup", "1", "--config", "tiny", "--backend", "nccl",
           "--no-cpu-baseline", "--no-feature-cache-pass", "--no-fp8-pass", "--no-prune-pass", "--no-batched-pass"]
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(_free_port()), RANK="0", WORLD_SIZE="1",
               LOCAL_RANK="0")
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=900, cwd=ROOT, env=env)
    assert r.returncode == 0, r.stderr[-3000:]
    d = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][0])
    assert d["n_gpus"] == 1 and d["metric_allreduce_check"]["length"] == 1 and d["value"] > 0
    assert d["config"]["dist_backend"] == "nccl"
