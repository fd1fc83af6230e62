#!/usr/bin/env python
"""Kernel micro-bench driver: calls single-kernel C-ABI entry points on true-size shapes so rocprofv3
(--kernel-trace / --pmc) can attribute time and counters per kernel.  Usage:
    rocprofv3 --kernel-trace --output-format csv -d out -- python tools/kbench.py gemm [reps]
"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from streamvln_amd import _lib
from streamvln_amd.config import TINY
from streamvln_amd.model import StreamVLNForCausalLM


def ptr(t):
    import ctypes as C
    return C.c_void_p(t.data_ptr()) if t is not None else None


def main():
    what = sys.argv[1] if len(sys.argv) > 1 else "gemm"
    reps = int(sys.argv[2]) if len(sys.argv) > 2 else 10
    m = StreamVLNForCausalLM(TINY, dtype=torch.bfloat16, max_envs=1, max_frames=9, max_positions=4096)
    lib, h = m._lib, m._h
    dt = torch.bfloat16
    if what == "gemm":
        from streamvln_amd.config import TRUE1
        m.close()       # true-width engine: its split-K workspace holds the slabs of the true-width products
        m = StreamVLNForCausalLM(TRUE1, dtype=torch.bfloat16, max_envs=1, max_frames=9, max_positions=4096)
        lib, h = m._lib, m._h
        # (M, N, K, epi, force_cfg, force_split)  -- steady prefill and ViT shapes
        shapes = [(212, 37888, 3584, _lib.EPI_SWIGLU, 0, 0), (212, 4608, 3584, 0, 0, 0), (212, 3584, 3584, 0, 0, 0),
                  (212, 3584, 18944, 0, 0, 0), (729, 3456, 1152, 0, 0, 0), (729, 1152, 1152, 0, 0, 0),
                  (729, 4304, 1152, _lib.EPI_GELU_TANH, 0, 0), (729, 1152, 4304, 0, 0, 0), (1952, 37888, 3584, _lib.EPI_SWIGLU, 0, 0)]
        extra = os.environ.get("KBENCH_SHAPES")
        if extra:
            shapes = [tuple(int(v) for v in s.split(",")) for s in extra.split(";")]
        for (M, N, K, epi, fc, fs) in shapes:
            A = (torch.rand(M, K, device="cuda") - 0.5).to(dt)
            # rotate over enough weight copies to exceed the 256 MB infinity cache: in the pipeline every layer's weights
            # come cold from HBM (KBENCH_HOT=1 keeps one copy)
            nw = 1 if os.environ.get("KBENCH_HOT") else max(1, min(8, -(-768 * 2**20 // (N * K * 2))))
            Ws = [((torch.rand(N, K, device="cuda") - 0.5) * 0.05).to(dt) for _ in range(nw)]
            Cn = N // 2 if epi == _lib.EPI_SWIGLU else N
            out = torch.zeros(M, Cn, device="cuda", dtype=dt)
            torch.cuda.synchronize()
            for i in range(reps):
                W = Ws[i % nw]
                _lib.check(lib.svln_op_gemm(h, ptr(A), K, ptr(W), K, ptr(out), Cn, None, None, 0, 0, M, N, K, epi, fc, fs))
    elif what == "attn":
        # ViT attention on one frame (true dims) and LLM prefill attention T=212 over a 1700-key context (true dims, 1 layer)
        from streamvln_amd.config import TRUE1
        m.close()
        m = StreamVLNForCausalLM(TRUE1, dtype=torch.bfloat16, max_envs=1, max_frames=9, max_positions=4096)
        lib, h = m._lib, m._h
        cfg = TRUE1
        F = int(os.environ.get("KBENCH_FRAMES", "1"))
        qkv = (torch.rand(F * 729, 3 * cfg.v_hidden, device="cuda") - 0.5).to(dt)
        out = torch.zeros(F * 729, cfg.v_hidden, device="cuda", dtype=dt)
        torch.cuda.synchronize()
        for i in range(reps):
            _lib.check(lib.svln_op_attention_vit(h, ptr(qkv), 3 * cfg.v_hidden, F, ptr(out), cfg.v_hidden))
        T, P = int(os.environ.get("KBENCH_T", "212")), int(os.environ.get("KBENCH_P", "1700"))
        qd = (cfg.q_heads + 2 * cfg.kv_heads) * 128
        ctx = (torch.rand(P, qd, device="cuda") - 0.5).to(dt)
        new = (torch.rand(T, qd, device="cuda") - 0.5).to(dt)
        o2 = torch.zeros(T, cfg.q_heads * 128, device="cuda", dtype=dt)
        torch.cuda.synchronize()
        for i in range(reps):
            _lib.check(lib.svln_op_attention_llm(h, ptr(new.clone()), qd, T, P, ptr(ctx.clone()), P, ptr(o2), cfg.q_heads * 128, 1))
    elif what == "gemv":
        for (N, K, norm, epi) in [(37888, 3584, True, _lib.EPI_SWIGLU), (4608, 3584, True, 0), (3584, 3584, False, 0),
                                  (3584, 18944, False, 0), (152064, 3584, False, _lib.EPI_ARGMAX)]:
            nw = 1 if os.environ.get("KBENCH_HOT") else max(1, min(8, -(-768 * 2**20 // (N * K * 2))))
            Ws = [((torch.rand(N, K, device="cuda") - 0.5) * 0.05).to(dt) for _ in range(nw)]
            x = (torch.rand(K, device="cuda") - 0.5).to(dt)
            g = torch.ones(K, device="cuda", dtype=dt) if norm else None
            y = torch.zeros(N, device="cuda", dtype=dt)
            import ctypes as C
            tok = C.c_int32()
            torch.cuda.synchronize()
            for i in range(reps):
                W = Ws[i % nw]
                _lib.check(lib.svln_op_gemv(h, ptr(W), K, ptr(x), ptr(g), 1e-6, None, None, ptr(y), N, K, epi, C.byref(tok)))
    m.close()


if __name__ == "__main__":
    main()
