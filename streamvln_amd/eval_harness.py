"""Episode-parallel evaluation: the only way the path is partitioned across GPUs.

Reference: streamvln/streamvln_eval.py:213-226 (within each scene, sorted, rank r takes
`episodes[r::world_size]`; one env + one model replica per GPU; no communication during rollouts) and
:553-581 (metrics combined once at the end: `all_gather` of the per-rank episode count, then of the four
per-episode metric vectors between two barriers; rank 0 appends the summary to result.json).

Here the default exchange is ONE RCCL `all_reduce(SUM)` of five fp64 scalars
[sum success, sum spl, sum os, sum ne, n] over xGMI (same means up to fp summation order);
`mode="all_gather"` reproduces the reference's per-episode gather for exact parity of the summary.
The resume contract is kept: result.json is append-only JSON lines and finished
(scene, episode, instruction) triples are skipped on restart (streamvln_eval.py:203-224,365-377).
"""
from __future__ import annotations

import json
import os
from typing import Dict, Iterable, List, Sequence

import torch
import torch.distributed as dist

from .dist import get_rank, get_world_size, is_dist_avail_and_initialized

METRIC_KEYS = ("success", "spl", "os", "ne")


def shard_episodes(scene_episodes: Dict[str, Sequence], rank: int, world_size: int):
    """[(scene, episode)] for this rank: scenes in sorted order, episodes[rank::world_size] inside each."""
    out = []
    for scene in sorted(scene_episodes.keys()):
        for ep in scene_episodes[scene][rank::world_size]:
            out.append((scene, ep))
    return out


def load_done(result_path: str):
    """Finished episodes + their metrics from an existing result.json (resume-by-skip)."""
    done, metrics = [], []
    if os.path.exists(result_path):
        with open(result_path) as f:
            for line in f:
                line = line.strip()
                if not line:
                    continue
                res = json.loads(line)
                if "scene_id" not in res:       # the trailing summary line
                    continue
                done.append([res["scene_id"], res["episode_id"], res["episode_instruction"]])
                metrics.append({k: res[k] for k in METRIC_KEYS})
    return done, metrics


def append_result(result_path: str, record: dict):
    os.makedirs(os.path.dirname(result_path) or ".", exist_ok=True)
    with open(result_path, "a") as f:
        f.write(json.dumps(record) + "\n")


def reduce_metrics(per_episode: List[Dict[str, float]], device: torch.device | str = "cpu", mode: str = "all_reduce"):
    """Combine per-rank episode metrics into the reference's summary dict
    {sucs_all, spls_all, oss_all, ones_all, length} (streamvln_eval.py:570-576)."""
    n = len(per_episode)
    cols = {k: [float(m[k]) for m in per_episode] for k in METRIC_KEYS}
    if not is_dist_avail_and_initialized():
        sums = {k: sum(v) for k, v in cols.items()}
        total = n
    elif mode == "all_reduce":
        t = torch.tensor([sum(cols["success"]), sum(cols["spl"]), sum(cols["os"]), sum(cols["ne"]), float(n)],
                         dtype=torch.float64, device=device)
        dist.all_reduce(t, op=dist.ReduceOp.SUM)
        sums = dict(zip(METRIC_KEYS, t[:4].tolist()))
        total = int(round(t[4].item()))
    elif mode == "all_gather":
        world = get_world_size()
        cnt = torch.tensor(n, device=device)
        cnts = [torch.zeros_like(cnt) for _ in range(world)]
        dist.all_gather(cnts, cnt)
        dist.barrier()
        sums, total = {}, int(sum(int(c) for c in cnts))
        mx = max(int(c) for c in cnts)
        for k in METRIC_KEYS:
            mine = torch.zeros(mx, dtype=torch.float64, device=device)      # equal-sized buffers (gloo/RCCL safe)
            mine[:n] = torch.tensor(cols[k], dtype=torch.float64, device=device) if n else mine[:0]
            bufs = [torch.zeros(mx, dtype=torch.float64, device=device) for _ in range(world)]
            dist.all_gather(bufs, mine)
            cat = torch.cat([b[: int(c)] for b, c in zip(bufs, cnts)])
            sums[k] = float(sum(cat.tolist()))                                # rank-order concatenation, python sum as in the reference
        dist.barrier()
    else:
        raise ValueError(mode)
    if total == 0:
        return {"sucs_all": 0.0, "spls_all": 0.0, "oss_all": 0.0, "ones_all": 0.0, "length": 0}
    return {"sucs_all": sums["success"] / total, "spls_all": sums["spl"] / total, "oss_all": sums["os"] / total,
            "ones_all": sums["ne"] / total, "length": total}


def run_sharded(scene_episodes: Dict[str, Sequence], run_episode, result_path: str | None = None,
                device: torch.device | str = "cpu", mode: str = "all_reduce"):
    """Evaluate this rank's shard with `run_episode(scene, episode) -> {success, spl, os, ne, ...}` and
    return the global summary.  `episode` needs `.episode_id` and `.instruction_text` when resuming."""
    rank, world = get_rank(), get_world_size()
    done, prior = load_done(result_path) if result_path else ([], [])
    mine = prior if rank == 0 else []                                       # rank 0 reloads finished episodes (:208-212)
    mine = list(mine)
    for scene, ep in shard_episodes(scene_episodes, rank, world):
        key = [scene, getattr(ep, "episode_id", ep), getattr(ep, "instruction_text", "")]
        if key in done:
            continue
        m = run_episode(scene, ep)
        mine.append({k: m[k] for k in METRIC_KEYS})
        if result_path:
            rec = {"scene_id": key[0], "episode_id": key[1], "episode_instruction": key[2]}
            rec.update({k: m[k] for k in METRIC_KEYS})
            rec["steps"] = m.get("steps", 0)
            append_result(result_path, rec)
    summary = reduce_metrics(mine, device=device, mode=mode)
    if rank == 0 and result_path:
        append_result(result_path, summary)
    return summary
