"""Yardstick only (never on the product path): what the vendor GEMM (torch.matmul -> hipBLASLt / rocBLAS) reaches on this box for
the products of the path, HBM-cold weights (rotated through > 1 GB), bf16.  usage: python tools/blaslt_yardstick.py"""
import torch

shapes = [  # (name, M, N, K)
    ("steady gate/up", 212, 37888, 3584), ("steady qkv", 212, 4608, 3584), ("steady o", 212, 3584, 3584), ("steady down", 212, 3584, 18944),
    ("restart gate/up", 1952, 37888, 3584), ("restart qkv", 1952, 4608, 3584), ("restart o", 1952, 3584, 3584), ("restart down", 1952, 3584, 18944),
    ("vit1 qkv", 729, 3456, 1152), ("vit1 out", 729, 1152, 1152), ("vit1 fc1", 729, 4304, 1152), ("vit1 fc2", 729, 1152, 4304),
    ("vit9 qkv", 6561, 3456, 1152), ("vit9 out", 6561, 1152, 1152), ("vit9 fc1", 6561, 4304, 1152), ("vit9 fc2", 6561, 1152, 4304),
]
dev = torch.device("cuda:0")
for name, M, N, K in shapes:
    copies = max(2, int(1.2e9 // (N * K * 2)) + 1)
    copies = min(copies, 64)
    W = torch.randn(copies, N, K, device=dev, dtype=torch.bfloat16)
    A = torch.randn(M, K, device=dev, dtype=torch.bfloat16)
    for i in range(3):
        torch.matmul(A, W[i % copies].t())
    torch.cuda.synchronize()
    reps = 20
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for i in range(reps):
        torch.matmul(A, W[i % copies].t())
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1e3 / reps
    print(f"{name:16s} M {M:5d} N {N:6d} K {K:6d}  {us:8.1f} us  {2.0 * M * N * K / us / 1e6:7.1f} TF/s  W {N * K * 2 / us / 1e6:5.2f} TB/s", flush=True)
    del W, A
