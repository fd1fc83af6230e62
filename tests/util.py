"""Shared helpers for the parity tests (test infrastructure)."""
import ctypes as C
import os

import numpy as np
import torch

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def ptr(t):
    return C.c_void_p(t.data_ptr()) if t is not None else C.c_void_p(0)


def rnd(shape, seed, scale=1.0):
    g = torch.Generator().manual_seed(seed)
    return (torch.rand(shape, generator=g) * 2 - 1) * scale


def q(t, dtype):
    """round to the engine dtype and come back to fp32 (so oracle and kernel see identical inputs)"""
    return t.to(dtype).to(torch.float32)


def tol(dtype):
    # fp32 parity mode: north-star tolerance 1e-3 on hidden states; kernels are far inside it.
    # bf16 shipping mode: one bf16 rounding of the output (2^-9 relative) + fp32-accumulate reorder.
    return (2e-4, 2e-4) if dtype == torch.float32 else (2e-2, 2e-2)


def assert_close(got, exp, dtype, what=""):
    rt, at = tol(dtype)
    got, exp = got.detach().float().cpu(), exp.detach().float().cpu()
    err = (got - exp).abs()
    bound = at * max(1.0, float(exp.abs().max())) + rt * exp.abs()
    bad = err > bound
    assert not bool(bad.any()), f"{what}: max err {float(err.max()):.3e} (|exp| max {float(exp.abs().max()):.3e}), {int(bad.sum())} bad of {bad.numel()}"


def load_golden(name):
    return np.load(os.path.join(GOLD, name + ".npz"))
