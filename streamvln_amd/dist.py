"""One process per GPU; `torch.distributed` backend "nccl" is RCCL on ROCm (xGMI between the 8 GPUs of a node).

Mirrors streamvln/utils/dist.py:48-89 (`init_distributed_mode`: env:// rendezvous from torchrun, barrier,
rank-0-only printing) without the CUDA hard-coding: falls back to gloo on hosts without a GPU so the
multi-process path is testable on CPU.
"""
from __future__ import annotations

import builtins
import datetime
import os

import torch
import torch.distributed as dist


def is_dist_avail_and_initialized() -> bool:
    return dist.is_available() and dist.is_initialized()


def get_world_size() -> int:
    return dist.get_world_size() if is_dist_avail_and_initialized() else 1


def get_rank() -> int:
    return dist.get_rank() if is_dist_avail_and_initialized() else 0


def setup_for_distributed(is_master: bool):
    """rank-0-only print (utils/dist.py:10-23); pass force=True to print from any rank."""
    if getattr(builtins.print, "_svln_patched", False):
        return
    builtin_print = builtins.print

    def print(*args, **kwargs):
        force = kwargs.pop("force", False)
        if is_master or force:
            builtin_print(*args, **kwargs)
    print._svln_patched = True
    builtins.print = print


def init_distributed_mode(backend: str | None = None, timeout_s: int = 7200, quiet_workers: bool = False):
    """Returns (rank, world_size, local_rank).  No-op when RANK/WORLD_SIZE are absent."""
    if "RANK" not in os.environ or "WORLD_SIZE" not in os.environ:
        return 0, 1, 0
    rank, world, local = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"]), int(os.environ.get("LOCAL_RANK", 0))
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    if backend is None:
        backend = "nccl" if torch.cuda.is_available() else "gloo"
    if backend == "nccl":
        torch.cuda.set_device(local)
    os.environ.setdefault("MASTER_PORT", "29500")
    if not dist.is_initialized():
        dist.init_process_group(backend=backend, init_method="env://", world_size=world, rank=rank,
                                timeout=datetime.timedelta(seconds=timeout_s))
    dist.barrier()
    if quiet_workers:
        setup_for_distributed(rank == 0)
    return rank, world, local
