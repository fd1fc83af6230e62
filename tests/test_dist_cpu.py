"""CPU, world_size 2 over gloo: episode sharding + the single metric exchange of the path (streamvln_eval.py:213-219,553-581)."""
import json
import os
import subprocess
import sys
import tempfile

from streamvln_amd.eval_harness import reduce_metrics, shard_episodes

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

WORKER = r'''
import json, os, sys
sys.path.insert(0, sys.argv[1])
from streamvln_amd.dist import init_distributed_mode
from streamvln_amd.eval_harness import run_sharded
rank, world, local = init_distributed_mode(backend="gloo")
scenes = {"b/scene2/x.glb": list(range(7)), "a/scene1/x.glb": list(range(10, 15))}
def run_episode(scene, ep):
    return {"success": float(ep % 2), "spl": ep / 100.0, "os": 1.0, "ne": float(ep), "steps": ep}
out = {}
for mode in ("all_reduce", "all_gather"):
    out[mode] = run_sharded(scenes, run_episode, result_path=None, mode=mode)
if rank == 0:
    print("RESULT " + json.dumps(out), flush=True)
import torch.distributed as dist
dist.barrier(); dist.destroy_process_group()
'''


def test_shard_episodes_matches_reference_slicing():
    scenes = {"b": list(range(7)), "a": list(range(10, 15))}
    s0, s1 = shard_episodes(scenes, 0, 2), shard_episodes(scenes, 1, 2)
    assert s0 == [("a", 10), ("a", 12), ("a", 14), ("b", 0), ("b", 2), ("b", 4), ("b", 6)]
    assert s1 == [("a", 11), ("a", 13), ("b", 1), ("b", 3), ("b", 5)]
    assert sorted(s0 + s1) == sorted((k, e) for k, v in scenes.items() for e in v)


def test_single_process_summary():
    s = reduce_metrics([{"success": 1, "spl": 0.5, "os": 1, "ne": 2.0}, {"success": 0, "spl": 0.0, "os": 1, "ne": 4.0}])
    assert s == {"sucs_all": 0.5, "spls_all": 0.25, "oss_all": 1.0, "ones_all": 3.0, "length": 2}


def test_world_size_2_gloo():
    with tempfile.TemporaryDirectory() as d:
        w = os.path.join(d, "worker.py")
        open(w, "w").write(WORKER)
        env = dict(os.environ, MASTER_ADDR="127.0.0.1")
        r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr",
                            "127.0.0.1", "--master-port", "29571", w, ROOT], capture_output=True, text=True, timeout=240, env=env)
        assert r.returncode == 0, r.stderr[-2000:]
        line = [l for l in r.stdout.splitlines() if l.startswith("RESULT ")][0]
        out = json.loads(line[len("RESULT "):])
    eps = list(range(7)) + list(range(10, 15))
    exp = {"sucs_all": sum(e % 2 for e in eps) / 12, "spls_all": sum(e / 100.0 for e in eps) / 12, "oss_all": 1.0,
           "ones_all": sum(eps) / 12, "length": 12}
    for mode in ("all_reduce", "all_gather"):
        for k, v in exp.items():
            assert abs(out[mode][k] - v) < 1e-12, (mode, k, out[mode][k], v)


def test_bench_starts_its_own_ranks_without_a_launcher():
    """`python bench.py --gpus 2` with no RANK in the environment starts `torch.distributed.run` as a child process and passes its
    return code through (the GPU box runs the same command to completion: tests/test_dist_gpu.py).  Without a GPU both ranks get past
    the rendezvous and the WORLD_SIZE check and stop at the 'needs a GPU' assertion; the parent must report that failure, not
    'WORLD_SIZE 1'."""
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK")}
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--backend", "gloo", "--single-device", "--config",
                        "tiny", "--steps", "2", "--warmup", "1"], capture_output=True, text=True, timeout=600, env=env, cwd=ROOT)
    import torch
    if torch.cuda.is_available():       # on a GPU box the run completes
        assert r.returncode == 0, r.stderr[-2000:]
        return
    assert r.returncode != 0
    assert "WORLD_SIZE 1" not in r.stderr
    assert r.stderr.count("bench.py needs a GPU") >= 2, r.stderr[-2000:]          # both ranks were started and initialised gloo
