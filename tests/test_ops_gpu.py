"""Op-level parity: every HIP kernel, called through the C ABI, against the CPU oracle on the same
seeded inputs, in both engine dtypes (fp32 parity mode: tight; bf16 shipping mode: bf16-rounding bound)."""
import ctypes as C
import math
import os

import numpy as np
import pytest
import torch

from streamvln_amd import _lib
from streamvln_amd.config import TINY, TRUE1
from streamvln_amd.model import StreamVLNForCausalLM
from streamvln_amd import weights as W
from oracle import streamvln_oracle as O
from util import assert_close, ptr, q, rnd

pytestmark = pytest.mark.gpu
DTYPES = [torch.float32, torch.bfloat16]
_engines = {}


def engine(cfg, dtype):
    key = (cfg.name, dtype)
    if key not in _engines:
        # true width: room for the window-restart shapes (9-frame ViT batch, T = 1952 prefill)
        _engines[key] = StreamVLNForCausalLM(cfg, dtype=dtype, max_envs=1, max_frames=9 if cfg is TRUE1 else 3, max_positions=2048)
    return _engines[key]


def chk(rc):
    _lib.check(rc)


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("cfgsplit", [(0, 0), (128, 0), (0, 1), (0, 3), (256, 0), (256 | 0x4000, 0), (256 | 0x8000, 0), (258, 0), (0x10000, 0), (128 | 0x10000, 0), (256 | 0x10000, 0), (32, 0), (32, 4), (64, 0), (129, 0), (264, 0), (264, 3)])   # 264 = 256x64 tiles (M <= 256), 8 waves stacked along M; 256 = 256x256 tiles (bf16: the 8-phase schedule; | 0x8000: its 32x32x16 form; | 0x4000: the stage-ring kernel); 258 = 256x256 tiles, two K slices on the 8-phase kernel; | 0x10000 = column tiles fastest in the workgroup order; 32 = 32x128 tiles (M <= 32 only); 64 = 64x64 tiles; 129 = 128x128 tiles, two in-workgroup K groups
@pytest.mark.parametrize("M,N,K,epi,bias,res", [
    (212, 512, 3584, _lib.EPI_NONE, True, False),        # qkv-like, ragged M
    (300, 384, 1152, _lib.EPI_NONE, True, True),         # out_proj + residual
    (729, 4304, 1152, _lib.EPI_GELU_TANH, True, False),  # ViT fc1: N not a multiple of 128
    (130, 1152, 4304, _lib.EPI_NONE, True, True),        # ViT fc2: K not a multiple of the stage
    (64, 256, 592, _lib.EPI_GELU_ERF, True, False),      # patch-embed K
    (1, 128, 64, _lib.EPI_NONE, False, False),           # degenerate
    (729, 1152, 1152, _lib.EPI_NONE, True, True),        # ViT out_proj at one frame
    (1300, 384, 320, _lib.EPI_NONE, True, True),         # more rows than columns, several row tiles: column tiles fastest in the workgroup order
])
def test_gemm(dtype, M, N, K, epi, bias, res, cfgsplit):
    if cfgsplit[0] == 32 and M > 32:
        M = 8 if M % 2 else 32                        # the 32-row config is chosen for M <= 32 only: rerun the shape with few rows
    m = engine(TINY, dtype)
    A, Wt = q(rnd((M, K), 1), dtype), q(rnd((N, K), 2, 1.0 / math.sqrt(K)), dtype)
    b = q(rnd((N,), 3, 0.1), dtype) if bias else None
    r = q(rnd((M, N), 4), dtype) if res else None
    exp = A @ Wt.t()
    if b is not None:
        exp = exp + b
    if epi == _lib.EPI_GELU_TANH:
        exp = O.gelu_tanh(exp)
    if epi == _lib.EPI_GELU_ERF:
        exp = O.gelu_erf(exp)
    if r is not None:
        exp = exp + r
    dA, dW = A.to(dtype).cuda(), Wt.to(dtype).cuda()
    db = b.to(dtype).cuda() if bias else None
    dr = r.to(dtype).cuda() if res else None
    out = torch.zeros((M, N), dtype=dtype, device="cuda")
    torch.cuda.synchronize()
    chk(m._lib.svln_op_gemm(m._h, ptr(dA), K, ptr(dW), K, ptr(out), N, ptr(db), ptr(dr), N, 0, M, N, K, epi, *cfgsplit))
    assert_close(out, exp, dtype, f"gemm {M}x{N}x{K} epi{epi} cfg{cfgsplit}")


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("M,N,K,split,kind,expect_fused", [
    (212, 512, 1024, 0, "rms", True),       # steady prefill rows: heuristic K split -> fused reduce + norm
    (37, 3584, 3584, 4, "rms", True),       # o_proj at true width, ragged M
    (212, 3584, 18944 // 8, 3, "rms", True),    # down_proj-like
    (1, 512, 512, 2, "rms", True),
    (1400, 4096, 512, 0, "rms", False),     # a full round of 128x128 tiles -> unsplit: the caller must run the norm itself
    (700, 2048, 1024, 0, "rms", True),      # less than a round of them -> K-split product, fused
    (729, 1152, 1152, 0, "ln", True),       # SigLIP out_proj, one frame: K-split product, the reduce emits layer_norm2
    (729, 1152, 1152, 9, "ln", True),       # ... also with a forced split count
    (729, 1152, 4304, 8, "ln", True),       # SigLIP fc2 -> next layer_norm1
    (50, 144, 288, 2, "ln", True),
    (1952, 3584, 8192, 0, "rms", "bf16"),   # down_proj-like at the window-restart rows: 8 x 14 tiles of 256x256, two K slices on the 8-phase kernel
])
def test_gemm_fused_norm(dtype, M, N, K, split, kind, expect_fused):
    """o_proj / down_proj (+ residual) with the following RMSNorm, and SigLIP out_proj / fc2 (+ bias + residual) with the following
    LayerNorm, emitted by the split-K reduce (modeling_qwen2.py:269-299, siglip_encoder.py:269-305)."""
    import ctypes as C
    if expect_fused == "bf16":                          # (the 8-phase kernel takes bf16 operands: the fp32 engine runs this product unsplit)
        expect_fused = dtype == torch.bfloat16
    m = engine(TINY, dtype)
    ln = kind == "ln"
    A, Wt = q(rnd((M, K), 11), dtype), q(rnd((N, K), 12, 1.0 / math.sqrt(K)), dtype)
    r, g = q(rnd((M, N), 13), dtype), q(1.0 + rnd((N,), 14, 0.2), dtype)
    b = q(rnd((N,), 15, 0.1), dtype) if ln else None
    nb = q(rnd((N,), 16, 0.1), dtype) if ln else None
    full = A @ Wt.t() + r + (b if ln else 0.0)
    norm = (lambda t: O.layer_norm(t, g, nb, 1e-6)) if ln else (lambda t: O.rms_norm(t, g, 1e-6))
    dA, dW, dg = A.to(dtype).cuda(), Wt.to(dtype).cuda(), g.to(dtype).cuda()
    db = b.to(dtype).cuda() if ln else None
    dnb = nb.to(dtype).cuda() if ln else None
    x = r.to(dtype).cuda()                             # in place: C == res, like the engine's residual stream
    xn = torch.full((M, N), 7.0, dtype=dtype, device="cuda")
    fused = C.c_int32(-1)
    torch.cuda.synchronize()
    chk(m._lib.svln_op_gemm_norm(m._h, ptr(dA), K, ptr(dW), K, ptr(x), N, ptr(db), ptr(x), N, ptr(dg), ptr(dnb), ptr(xn), 1e-6, M, N, K, split,
                                 C.byref(fused)))
    assert fused.value == int(expect_fused)
    assert_close(x, full, dtype, f"gemm+res {M}x{N}x{K}")
    if expect_fused:
        # the residual stream is stored in the engine dtype and the norm reads the stored rows
        exp_norm = norm(x.float().cpu()) if dtype == torch.bfloat16 else norm(full)
        assert_close(xn, exp_norm, dtype, f"fused {kind} norm {M}x{N}x{K}")
    else:
        assert float(xn.float().min()) == 7.0 and float(xn.float().max()) == 7.0


@pytest.mark.parametrize("M,N,K,epi", [
    (1952, 1024, 3584, _lib.EPI_NONE),         # window-restart rows (7.6 row tiles), 56 K tiles
    (700, 768, 192, _lib.EPI_NONE),            # 3 K tiles (odd: the second half of the last iteration is skipped)
    (513, 512, 320, _lib.EPI_GELU_TANH),       # 5 K tiles
    (300, 300, 64, _lib.EPI_NONE),             # one K tile; ragged M and N
    (729, 1152, 4304, _lib.EPI_NONE),          # 67.25 K tiles: ragged last tile
    (1000, 640, 128, _lib.EPI_NONE),           # two K tiles: prologue only
    (1952, 2048, 1024, _lib.EPI_SWIGLU),       # gate/up epilogue
    (6561, 3456, 1152, _lib.EPI_NONE),         # nine-frame ViT qkv: 26 x 14 tiles, the last column tile half empty, 18 K tiles
    (1000, 4304, 1152, _lib.EPI_GELU_TANH),    # fc1-shaped: GELU epilogue, the last column tile 208 wide
])
def test_gemm_8phase_equals_stage_ring(M, N, K, epi):
    """The 8-phase schedule of the 256x256 tile (gemm.hip: p8_mainloop).  Its 32x32x16 form (force_cfg 256 | 0x8000) accumulates every output
    over K in the same order as the stage-ring kernel of the same tile (256 | 0x4000), so the two must agree BIT FOR BIT; the shipping
    16x16x32 form (256) sums K in groups of 32, so it is held to the fp32 product within the bf16 bound -- and to ITSELF bit for bit over
    repeated launches: a race in the half-tile staging (a read before its DMA landed, a re-stage before the last read) shows as a tile
    that differs between launches on the same operands."""
    m = engine(TINY, torch.bfloat16)
    A = q(rnd((M, K), 71), torch.bfloat16).to(torch.bfloat16).cuda()
    W = q(rnd((N, K), 72, 1.0 / math.sqrt(K)), torch.bfloat16).to(torch.bfloat16).cuda()
    b = q(rnd((N,), 73, 0.1), torch.bfloat16).to(torch.bfloat16).cuda()
    n_out = N // 2 if epi == _lib.EPI_SWIGLU else N
    bias = None if epi == _lib.EPI_SWIGLU else ptr(b)

    def run(cfg, fill):
        out = torch.full((M, n_out), fill, dtype=torch.bfloat16, device="cuda")
        torch.cuda.synchronize()
        chk(m._lib.svln_op_gemm(m._h, ptr(A), K, ptr(W), K, ptr(out), n_out, bias, None, 0, 0, M, N, K, epi, cfg, 0))
        torch.cuda.synchronize()
        return out

    ring = run(256 | 0x4000, 0.0)
    if epi == _lib.EPI_SWIGLU:                   # (the gate / up packing itself is covered by test_gemm_swiglu_and_posmod)
        acc = A.float().cpu() @ W.float().cpu().t()
        acc = acc.view(M, N // 64, 2, 32)
        exp = (O.silu(acc[:, :, 0]) * acc[:, :, 1]).reshape(M, n_out)
    else:
        exp = A.float().cpu() @ W.float().cpu().t() + b.float().cpu()
        if epi == _lib.EPI_GELU_TANH:
            exp = O.gelu_tanh(exp)
    assert_close(ring, exp, torch.bfloat16, f"stage ring {M}x{N}x{K}")
    first16 = run(256, 5.0)
    assert_close(first16, exp, torch.bfloat16, f"8-phase 16x16x32 {M}x{N}x{K}")
    for rep in range(int(os.environ.get("SVLN_RACE_REPS", "10"))):         # (SVLN_RACE_REPS=400 for a long race screen of a schedule edit)
        out = run(256 | 0x8000, 3.0)
        assert torch.equal(out.view(torch.int16), ring.view(torch.int16)), ("32x32x16 form", rep, int((out.view(torch.int16) != ring.view(torch.int16)).sum()))
        out = run(256, 3.0)
        assert torch.equal(out.view(torch.int16), first16.view(torch.int16)), ("16x16x32 form", rep, int((out.view(torch.int16) != first16.view(torch.int16)).sum()))
    # the LDS-staged stores of the 16x16x32 form (16-byte aligned output rows) against its direct 2-byte stores (| 0x20000): the same bits
    direct = run(256 | 0x20000, 7.0)
    assert torch.equal(direct.view(torch.int16), first16.view(torch.int16)), ("staged vs direct epilogue", int((direct.view(torch.int16) != first16.view(torch.int16)).sum()))


@pytest.mark.parametrize("epi", [_lib.EPI_NONE, _lib.EPI_GELU_TANH, _lib.EPI_SWIGLU])
def test_gemm_8phase_staged_epilogue_ragged_rows_and_residual(epi):
    """p8_epilogue16_staged: output rows wider than N (ldc > N), N not a multiple of 8 (the last 16-byte chunk of a row is stored element
    by element and the padding columns keep their fill), M not a multiple of 64, bias and a residual with res_mod -- bit for bit against the
    direct epilogue."""
    m = engine(TINY, torch.bfloat16)
    M, K = 700, 192
    N = 1100 if epi != _lib.EPI_SWIGLU else 1088          # SwiGLU: whole [gate 32 | up 32] blocks
    n_out = N // 2 if epi == _lib.EPI_SWIGLU else N
    ldc = (n_out + 15) // 8 * 8                            # 16-byte aligned rows, a few padding columns
    A = q(rnd((M, K), 81), torch.bfloat16).to(torch.bfloat16).cuda()
    W = q(rnd((N, K), 82, 1.0 / math.sqrt(K)), torch.bfloat16).to(torch.bfloat16).cuda()
    b = q(rnd((N,), 83, 0.1), torch.bfloat16).to(torch.bfloat16).cuda()
    R = q(rnd((100, n_out), 84), torch.bfloat16).to(torch.bfloat16).cuda()
    glu = epi == _lib.EPI_SWIGLU

    def run(cfg):
        out = torch.full((M, ldc), 9.0, dtype=torch.bfloat16, device="cuda")
        torch.cuda.synchronize()
        chk(m._lib.svln_op_gemm(m._h, ptr(A), K, ptr(W), K, ptr(out), ldc, None if glu else ptr(b), None if glu else ptr(R), n_out, 0 if glu else 100,
                                M, N, K, epi, cfg, 0))
        torch.cuda.synchronize()
        return out
    staged, direct = run(256), run(256 | 0x20000)
    assert torch.equal(staged.view(torch.int16), direct.view(torch.int16)), int((staged.view(torch.int16) != direct.view(torch.int16)).sum())
    assert bool((staged[:, n_out:] == 9.0).all())
    acc = A.float().cpu() @ W.float().cpu().t()
    if glu:
        acc = acc.view(M, N // 64, 2, 32)
        exp = (O.silu(acc[:, :, 0]) * acc[:, :, 1]).reshape(M, n_out)
    else:
        exp = acc + b.float().cpu()
        if epi == _lib.EPI_GELU_TANH:
            exp = O.gelu_tanh(exp)
        exp = exp + R.float().cpu()[torch.arange(M) % 100]
    assert_close(staged[:, :n_out], exp, torch.bfloat16, f"staged epilogue epi {epi}")


@pytest.mark.parametrize("dtype", DTYPES)
def test_gemm_swiglu_and_posmod(dtype):
    m = engine(TINY, dtype)
    M, I, K = 150, 256, 512
    A = q(rnd((M, K), 5), dtype)
    gate, up = q(rnd((I, K), 6, 0.08), dtype), q(rnd((I, K), 7, 0.08), dtype)
    packed = torch.zeros((2 * I, K))
    for j in range(I):                                   # 32-row blocks: [gate 32 | up 32]
        packed[(j // 32) * 64 + j % 32] = gate[j]
        packed[(j // 32) * 64 + 32 + j % 32] = up[j]
    exp = O.silu(A @ gate.t()) * (A @ up.t())
    out = torch.zeros((M, I), dtype=dtype, device="cuda")
    dA, dW = A.to(dtype).cuda(), packed.to(dtype).cuda()
    torch.cuda.synchronize()
    chk(m._lib.svln_op_gemm(m._h, ptr(dA), K, ptr(dW), K, ptr(out), I, None, None, 0, 0, M, 2 * I, K, _lib.EPI_SWIGLU, 0, 0))
    assert_close(out, exp, dtype, "gemm swiglu")
    for cs in [(128, 0), (0, 1), (0, 4), (256, 0)]:
        out.zero_()
        torch.cuda.synchronize()
        chk(m._lib.svln_op_gemm(m._h, ptr(dA), K, ptr(dW), K, ptr(out), I, None, None, 0, 0, M, 2 * I, K, _lib.EPI_SWIGLU, *cs))
        assert_close(out, exp, dtype, f"gemm swiglu {cs}")
    # residual with row modulo (position embedding add of the patch GEMM)
    S, N = 50, 128
    Wt, pos, b = q(rnd((N, K), 8, 0.05), dtype), q(rnd((S, N), 9), dtype), q(rnd((N,), 10), dtype)
    exp = A @ Wt.t() + b + pos[torch.arange(M) % S]
    out = torch.zeros((M, N), dtype=dtype, device="cuda")
    dW, dp, db = Wt.to(dtype).cuda(), pos.to(dtype).cuda(), b.to(dtype).cuda()
    torch.cuda.synchronize()
    chk(m._lib.svln_op_gemm(m._h, ptr(dA), K, ptr(dW), K, ptr(out), N, ptr(db), ptr(dp), N, S, M, N, K, _lib.EPI_NONE, 0, 2))
    assert_close(out, exp, dtype, "gemm res_mod")


@pytest.mark.parametrize("dtype", DTYPES)
def test_gemm_tail_tiles(dtype):
    """one row tile, 296 column tiles (gate/up at T <= 256): first 256 tiles unsplit + K-split tail launch"""
    m = engine(TINY, dtype)
    M, I, K = 70, 18944, 256
    A = q(rnd((M, K), 61), dtype)
    gate, up = q(rnd((I, K), 62, 0.1), dtype), q(rnd((I, K), 63, 0.1), dtype)
    packed = torch.zeros((2 * I, K))
    idx = torch.arange(I)
    packed[(idx // 32) * 64 + idx % 32] = gate
    packed[(idx // 32) * 64 + 32 + idx % 32] = up
    dA, dW = A.to(dtype).cuda(), packed.to(dtype).cuda()
    out = torch.zeros((M, I), dtype=dtype, device="cuda")
    torch.cuda.synchronize()
    chk(m._lib.svln_op_gemm(m._h, ptr(dA), K, ptr(dW), K, ptr(out), I, None, None, 0, 0, M, 2 * I, K, _lib.EPI_SWIGLU, 0, 0))
    assert_close(out, O.silu(A @ gate.t()) * (A @ up.t()), dtype, "gemm swiglu tail tiles")
    b, r = q(rnd((2 * I,), 64, 0.1), dtype), q(rnd((M, 2 * I), 65), dtype)
    out2 = torch.zeros((M, 2 * I), dtype=dtype, device="cuda")
    db, dr = b.to(dtype).cuda(), r.to(dtype).cuda()
    torch.cuda.synchronize()
    chk(m._lib.svln_op_gemm(m._h, ptr(dA), K, ptr(dW), K, ptr(out2), 2 * I, ptr(db), ptr(dr), 2 * I, 0, M, 2 * I, K, _lib.EPI_NONE, 0, 0))
    assert_close(out2, A @ packed.t() + b + r, dtype, "gemm bias+res tail tiles")


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("N,K,norm,bias,res", [(4608, 3584, True, True, False), (3584, 18944, False, False, True),
                                               (515, 512, True, False, True), (7, 64, False, True, False)])
def test_gemv(dtype, N, K, norm, bias, res):
    m = engine(TINY, dtype)
    Wt, x = q(rnd((N, K), 11, 1.0 / math.sqrt(K)), dtype), q(rnd((K,), 12), dtype)
    g = q(1 + rnd((K,), 13, 0.1), dtype) if norm else None
    b = q(rnd((N,), 14, 0.1), dtype) if bias else None
    r = q(rnd((N,), 15), dtype) if res else None
    xe = O.rms_norm(x, g, 1e-6) if norm else x
    exp = Wt @ xe
    if b is not None:
        exp = exp + b
    if r is not None:
        exp = exp + r
    d = lambda t: t.to(dtype).cuda() if t is not None else None
    dW, dx, dg, db, dr = d(Wt), d(x), d(g), d(b), d(r)
    y = torch.zeros((N,), dtype=dtype, device="cuda")
    torch.cuda.synchronize()
    chk(m._lib.svln_op_gemv(m._h, ptr(dW), K, ptr(dx), ptr(dg), 1e-6, ptr(db), ptr(dr), ptr(y), N, K, _lib.EPI_NONE, None))
    assert_close(y, exp, dtype, f"gemv {N}x{K}")


@pytest.mark.parametrize("dtype", DTYPES)
def test_gemv_swiglu_and_argmax(dtype):
    m = engine(TINY, dtype)
    I, K = 1024, 512
    x = q(rnd((K,), 16), dtype)
    g = q(1 + rnd((K,), 17, 0.1), dtype)
    gate, up = q(rnd((I, K), 18, 0.08), dtype), q(rnd((I, K), 19, 0.08), dtype)
    packed = torch.zeros((2 * I, K))
    idx = torch.arange(I)
    packed[(idx // 32) * 64 + idx % 32] = gate
    packed[(idx // 32) * 64 + 32 + idx % 32] = up
    xe = O.rms_norm(x, g, 1e-6)
    exp = O.silu(gate @ xe) * (up @ xe)
    y = torch.zeros((I,), dtype=dtype, device="cuda")
    dW, dx, dg = packed.to(dtype).cuda(), x.to(dtype).cuda(), g.to(dtype).cuda()
    torch.cuda.synchronize()
    chk(m._lib.svln_op_gemv(m._h, ptr(dW), K, ptr(dx), ptr(dg), 1e-6, None, None, ptr(y), 2 * I, K, _lib.EPI_SWIGLU, None))
    assert_close(y, exp, dtype, "gemv swiglu")
    # arg-max with an exact tie between two rows: the lowest index must win (torch.argmax on CPU)
    V = 5000
    Wv = q(rnd((V, K), 20, 0.05), dtype)
    Wv[77] = q(x * 0.02, dtype)
    Wv[4321] = Wv[77]
    logits = Wv @ x
    assert int(torch.argmax(logits)) == 77 and float(logits[77]) == float(logits[4321])
    tok = C.c_int32(-1)
    dWv = Wv.to(dtype).cuda()
    torch.cuda.synchronize()
    chk(m._lib.svln_op_gemv(m._h, ptr(dWv), K, ptr(dx), None, 1e-6, None, None, None, V, K, _lib.EPI_ARGMAX, C.byref(tok)))
    assert tok.value == 77, tok.value
    # non-finite logits (a NaN activation, or every logit -inf) must come back as -1 -- never as an out-of-range row index that
    # the next step's embedding gather would dereference -- and finite maxima still win over NaN rows
    xn = x.clone(); xn[3] = float("nan")
    dxn = xn.to(dtype).cuda()
    torch.cuda.synchronize()
    chk(m._lib.svln_op_gemv(m._h, ptr(dWv), K, ptr(dxn), None, 1e-6, None, None, None, V, K, _lib.EPI_ARGMAX, C.byref(tok)))
    assert tok.value == -1, tok.value
    Wn = Wv.clone(); Wn[10:20] = float("nan")
    dWn = Wn.to(dtype).cuda()
    torch.cuda.synchronize()
    chk(m._lib.svln_op_gemv(m._h, ptr(dWn), K, ptr(dx), None, 1e-6, None, None, None, V, K, _lib.EPI_ARGMAX, C.byref(tok)))
    assert tok.value == 77, tok.value


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("B", [1, 2, 4, 8])
def test_gemv_batched_true_width(dtype, B):
    """gemv_batched_kernel (decode step of <= 2 lockstep envs, lm_head arg-max at every batch size) at the true hidden width 3584:
    plain + bias + residual, fused RMSNorm, SwiGLU over the [gate 32 | up 32] packing, per-env arg-max with ties -> lowest index.
    The bf16 instantiation takes the packed dot2 path (v_dot2_f32_bf16), which has no fp32 twin."""
    m = engine(TINY, dtype)
    K, N = 3584, 4608
    x = q(rnd((B, K), 130 + B), dtype)
    g = q(1 + rnd((K,), 31, 0.1), dtype)
    W = q(rnd((N, K), 32, 0.03), dtype)
    bias, res = q(rnd((N,), 33, 0.1), dtype), q(rnd((B, N), 34), dtype)
    dW, dx, dg, db, dr = (t.to(dtype).cuda() for t in (W, x, g, bias, res))
    y = torch.zeros((B, N), dtype=dtype, device="cuda")
    torch.cuda.synchronize()
    chk(m._lib.svln_op_gemv_batched(m._h, ptr(dW), K, ptr(dx), K, None, 1e-6, ptr(db), ptr(dr), N, ptr(y), N, N, K, _lib.EPI_NONE, B, None))
    assert_close(y, x @ W.t() + bias + res, dtype, f"gemv_batched B={B}")
    y.zero_()
    chk(m._lib.svln_op_gemv_batched(m._h, ptr(dW), K, ptr(dx), K, ptr(dg), 1e-6, None, None, 0, ptr(y), N, N, K, _lib.EPI_NONE, B, None))
    xe = torch.stack([O.rms_norm(x[b], g, 1e-6) for b in range(B)])
    assert_close(y, xe @ W.t(), dtype, f"gemv_batched norm B={B}")
    # SwiGLU: N = 2 I packed rows -> I outputs
    I = N // 2
    idx = torch.arange(I)
    gate, up = W[(idx // 32) * 64 + idx % 32], W[(idx // 32) * 64 + 32 + idx % 32]
    ys = torch.zeros((B, I), dtype=dtype, device="cuda")
    chk(m._lib.svln_op_gemv_batched(m._h, ptr(dW), K, ptr(dx), K, ptr(dg), 1e-6, None, None, 0, ptr(ys), I, N, K, _lib.EPI_SWIGLU, B, None))
    assert_close(ys, O.silu(xe @ gate.t()) * (xe @ up.t()), dtype, f"gemv_batched swiglu B={B}")
    # arg-max per env; env 0 has an exact tie between rows 77 and 4000 (lowest index wins), ragged N
    V = 4607
    Wv = W[:V].clone()
    Wv[77] = q(x[0] * 0.02, dtype)
    Wv[4000] = Wv[77]
    logits = x @ Wv.t()
    assert int(torch.argmax(logits[0])) == 77 and float(logits[0, 77]) == float(logits[0, 4000])
    toks = (C.c_int32 * 8)()
    dWv = Wv.to(dtype).cuda()
    torch.cuda.synchronize()
    chk(m._lib.svln_op_gemv_batched(m._h, ptr(dWv), K, ptr(dx), K, None, 1e-6, None, None, 0, None, 0, V, K, _lib.EPI_ARGMAX, B, toks))
    margins = torch.topk(logits, 2, dim=1).values
    for b in range(B):
        if b == 0 or float(margins[b, 0] - margins[b, 1]) > (1e-3 if dtype == torch.float32 else 5e-2):
            assert toks[b] == int(torch.argmax(logits[b])), (b, toks[b])


def test_forced_split_is_bounded_by_the_workspace():
    """svln_op_gemm(force_split=S) must not write S fp32 slabs past the split-K workspace: rejected with an error"""
    m = engine(TINY, torch.bfloat16)
    M, N, K = 256, 4096, 512
    A = torch.zeros((M, K), dtype=torch.bfloat16, device="cuda")
    W = torch.zeros((N, K), dtype=torch.bfloat16, device="cuda")
    Cm = torch.zeros((M, N), dtype=torch.bfloat16, device="cuda")
    torch.cuda.synchronize()
    rc = m._lib.svln_op_gemm(m._h, ptr(A), K, ptr(W), K, ptr(Cm), N, None, None, 0, 0, M, N, K, _lib.EPI_NONE, 0, 4096)
    assert rc != 0 and b"workspace" in m._lib.svln_last_error()
    rc = m._lib.svln_op_gemm(m._h, ptr(A), K, ptr(W), K, ptr(Cm), N - 2, None, None, 0, 0, M, N - 2, K, _lib.EPI_NONE, 0, 2)
    assert rc != 0 and b"multiple of 4" in m._lib.svln_last_error()


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("N,keep", [(588, 32), (392, 40), (70, 69), (5, 1)])
def test_memory_prune_selection(dtype, N, keep):
    """svln_op_memory_prune (opt-in extension, no reference counterpart) == oracle.prune_memory_tokens on the same tokens."""
    m = engine(TINY, dtype)
    H = TINY.hidden                                          # rows are the engine's hidden width; N <= max_frames * 196
    mem = q(rnd((N, H), 71) + 0.7 * rnd((1, H), 72) * (1.0 + rnd((N, 1), 73).abs()), dtype)      # a shared component of varying weight
    mem[3 % N] = mem[0]                                      # an exact tie: the lower index must rank first
    idx, score = O.prune_memory_tokens(mem, keep)
    dm = mem.to(dtype).cuda()
    out_idx = (C.c_int32 * keep)()
    out_score = np.zeros(N, dtype=np.float32)
    torch.cuda.synchronize()
    chk(m._lib.svln_op_memory_prune(m._h, ptr(dm), N, keep, out_idx, out_score.ctypes.data_as(C.POINTER(C.c_float))))
    got = list(out_idx)
    assert np.abs(out_score - score.numpy()).max() < 2e-5
    assert got == sorted(got) and len(set(got)) == keep
    if got != idx.tolist():                                  # only a near-tie at the cut may differ (summation order)
        cut = sorted(score.tolist())[keep - 1]
        for i in set(got) ^ set(idx.tolist()):
            assert abs(float(score[i]) - cut) < 2e-5, (i, float(score[i]), cut)


def _quant_ref(W):
    """per-row e4m3 quantisation as the engine does it: scale = amax / 448, q = e4m3(clamp(w * (1 / scale)))"""
    amax = W.abs().amax(dim=1)
    scale = torch.where(amax > 0, amax / 448.0, torch.ones_like(amax)).to(torch.float32)
    inv = (1.0 / scale).to(torch.float32)
    qf = (W * inv[:, None]).clamp(-448.0, 448.0).to(torch.float8_e4m3fn)
    return qf, scale


def test_fused_norm_quantisation_equals_the_quantise_kernel():
    """the split-K reduce that emits the RMSNorm can also emit its e4m3 copy (opt-in fp8 products): bytes and scales equal torch's
    float8_e4m3fn conversion of the bf16 norm rows = what svln_op_quant_fp8 produces from them"""
    m = engine(TINY, torch.bfloat16)
    dtype = torch.bfloat16
    for M, N, K, split in [(212, 3584, 1024, 4), (8, 3584, 2368, 3), (37, 512, 512, 2)]:
        A, Wt = q(rnd((M, K), 71), dtype), q(rnd((N, K), 72, 1.0 / math.sqrt(K)), dtype)
        r, g = q(rnd((M, N), 73), dtype), q(1.0 + rnd((N,), 74, 0.2), dtype)
        dA, dW, dg = A.to(dtype).cuda(), Wt.to(dtype).cuda(), g.to(dtype).cuda()
        x = r.to(dtype).cuda()
        xn = torch.zeros((M, N), dtype=dtype, device="cuda")
        q8 = torch.zeros((M, N), dtype=torch.uint8, device="cuda")
        sc = torch.zeros((M,), dtype=torch.float32, device="cuda")
        fused = C.c_int32(-1)
        torch.cuda.synchronize()
        chk(m._lib.svln_op_gemm_norm_q8(m._h, ptr(dA), K, ptr(dW), K, ptr(x), N, ptr(x), N, ptr(dg), ptr(xn), 1e-6, M, N, K, split, ptr(q8), ptr(sc),
                                        C.byref(fused)))
        assert fused.value == 1
        qf, scale = _quant_ref(xn.float().cpu())
        assert torch.equal(sc.cpu(), scale), (M, N, K)
        assert torch.equal(q8.cpu(), qf.view(torch.uint8)), (M, N, K)


def test_fp8_row_quantisation_matches_torch_e4m3():
    """svln_op_quant_fp8 (SURVEY 8f-2 extension): OCP e4m3 bytes and per-row scales equal torch's float8_e4m3fn conversion."""
    m = engine(TINY, torch.bfloat16)
    for rows, cols, sd in [(515, 3584, 0.02), (64, 18944, 0.5), (3, 16, 1.0)]:
        W = q(rnd((rows, cols), 31, sd), torch.bfloat16)
        W[0] = 0.0                                               # all-zero row: scale 1, bytes 0
        qf, scale = _quant_ref(W)
        dW = W.to(torch.bfloat16).cuda()
        w8 = torch.zeros((rows, cols), dtype=torch.uint8, device="cuda")
        sc = torch.zeros((rows,), dtype=torch.float32, device="cuda")
        torch.cuda.synchronize()
        chk(m._lib.svln_op_quant_fp8(m._h, ptr(dW), rows, cols, ptr(w8), ptr(sc)))
        assert torch.equal(sc.cpu(), scale), float((sc.cpu() - scale).abs().max())
        got, exp = w8.cpu(), qf.view(torch.uint8)
        got = torch.where(got == 0x80, torch.zeros_like(got), got)      # -0 == +0
        exp = torch.where(exp == 0x80, torch.zeros_like(exp), exp)
        assert torch.equal(got, exp), f"{int((got != exp).sum())} of {got.numel()} bytes differ ({rows}x{cols})"


@pytest.mark.parametrize("N,K,norm,bias,res", [(4608, 3584, True, True, False), (3584, 18944, False, False, True),
                                               (515, 512, True, False, True), (7, 64, False, True, False), (9000, 3584, False, False, False)])
def test_gemv_fp8_weights(N, K, norm, bias, res):
    """fp8 weight-only GEMV == the same product over the dequantised weights (fp32 accumulate both sides)."""
    dtype = torch.bfloat16
    m = engine(TINY, dtype)
    Wt, x = q(rnd((N, K), 41, 1.0 / math.sqrt(K)), dtype), q(rnd((K,), 42), dtype)
    g = q(1 + rnd((K,), 43, 0.1), dtype) if norm else None
    b = q(rnd((N,), 44, 0.1), dtype) if bias else None
    r = q(rnd((N,), 45), dtype) if res else None
    qf, scale = _quant_ref(Wt)
    Wd = qf.to(torch.float32) * scale[:, None]
    xe = O.rms_norm(x, g, 1e-6) if norm else x
    exp = Wd @ xe
    if b is not None:
        exp = exp + b
    if r is not None:
        exp = exp + r
    d = lambda t: t.to(dtype).cuda() if t is not None else None
    dx, dg, db, dr = d(x), d(g), d(b), d(r)
    w8, sc = qf.view(torch.uint8).cuda(), scale.cuda()
    y = torch.zeros((N,), dtype=dtype, device="cuda")
    torch.cuda.synchronize()
    chk(m._lib.svln_op_gemv_fp8(m._h, ptr(w8), ptr(sc), K, ptr(dx), ptr(dg), 1e-6, ptr(db), ptr(dr), ptr(y), N, K, _lib.EPI_NONE, None))
    assert_close(y, exp, dtype, f"gemv fp8 {N}x{K}")
    # the quantisation error itself stays at the e4m3 level (3 mantissa bits -> ~3 % per weight, averaged down by the dot product)
    full = Wt @ xe + (b if b is not None else 0) + (r if r is not None else 0)
    rel = float((exp - full).norm() / full.norm())
    assert rel < 0.05, rel


@pytest.mark.parametrize("M,N,K,epi,bias,res,cfgsplit", [
    (212, 512, 3584, _lib.EPI_NONE, True, True, (0, 0)),        # steady prefill rows: 256x128 tiles, heuristic K split
    (212, 1024, 3584, _lib.EPI_SWIGLU, False, False, (0, 0)),   # gate/up with the SwiGLU epilogue
    (212, 3584, 18944 // 4, _lib.EPI_NONE, False, True, (0, 3)),  # down_proj-like, forced 3-way split
    (8, 4608, 3584, _lib.EPI_NONE, True, False, (0, 0)),        # 8 lockstep envs: 32x128 tiles
    (5, 2048, 1024, _lib.EPI_SWIGLU, False, False, (32, 2)),
    (600, 2048, 1024, _lib.EPI_NONE, True, False, (0, 0)),      # 128x128 tiles
    (700, 1024, 512, _lib.EPI_NONE, False, False, (256, 0)),    # 256x256 tiles
    (300, 384, 1152, _lib.EPI_NONE, True, True, (64, 0)),       # 64x64 tiles
])
def test_gemm_fp8_mfma(M, N, K, epi, bias, res, cfgsplit):
    """e4m3 x e4m3 MFMA product with per-row scales on both operands (svln_set_fp8_gemm) against the fp32 product of the
    DEQUANTISED operands (quantisation error is the model's business, not the kernel's)."""
    dtype = torch.bfloat16
    m = engine(TINY, dtype)
    A = q(rnd((M, K), 41), dtype)
    W = q(rnd((N, K), 42, 1.0 / math.sqrt(K)), dtype)
    dA, dW = A.to(dtype).cuda(), W.to(dtype).cuda()
    A8, W8 = torch.zeros((M, K), dtype=torch.uint8, device="cuda"), torch.zeros((N, K), dtype=torch.uint8, device="cuda")
    sa, sw = torch.zeros((M,), dtype=torch.float32, device="cuda"), torch.zeros((N,), dtype=torch.float32, device="cuda")
    torch.cuda.synchronize()
    chk(m._lib.svln_op_quant_fp8(m._h, ptr(dA), M, K, ptr(A8), ptr(sa)))
    chk(m._lib.svln_op_quant_fp8(m._h, ptr(dW), N, K, ptr(W8), ptr(sw)))
    Ad = A8.view(torch.float8_e4m3fn).float().cpu() * sa.cpu()[:, None]
    Wd = W8.view(torch.float8_e4m3fn).float().cpu() * sw.cpu()[:, None]
    assert float((Ad - A).abs().max()) <= float(A.abs().max()) * 2.0 ** -4            # e4m3: 3 mantissa bits
    exp = Ad @ Wd.t()
    b = q(rnd((N,), 43, 0.1), dtype) if bias else None
    r = q(rnd((M, N), 44), dtype) if res else None
    if b is not None:
        exp = exp + b
    n_out = N
    if epi == _lib.EPI_SWIGLU:
        idx = torch.arange(N // 2)
        exp = O.silu(exp[:, (idx // 32) * 64 + idx % 32]) * exp[:, (idx // 32) * 64 + 32 + idx % 32]
        n_out = N // 2
    if r is not None:
        exp = exp + r
    db = b.to(dtype).cuda() if bias else None
    dr = r.to(dtype).cuda() if res else None
    out = torch.zeros((M, n_out), dtype=dtype, device="cuda")
    torch.cuda.synchronize()
    chk(m._lib.svln_op_gemm_fp8(m._h, ptr(A8), ptr(sa), K, ptr(W8), ptr(sw), K, ptr(out), n_out, ptr(db), ptr(dr), n_out, M, N, K, epi, *cfgsplit))
    assert_close(out, exp, dtype, f"fp8 gemm {M}x{N}x{K} epi{epi} cfg{cfgsplit}")


def test_gemv_fp8_swiglu_and_argmax():
    dtype = torch.bfloat16
    m = engine(TINY, dtype)
    I, K = 1024, 512
    x = q(rnd((K,), 46), dtype)
    g = q(1 + rnd((K,), 47, 0.1), dtype)
    gate, up = q(rnd((I, K), 48, 0.08), dtype), q(rnd((I, K), 49, 0.08), dtype)
    packed = torch.zeros((2 * I, K))
    idx = torch.arange(I)
    packed[(idx // 32) * 64 + idx % 32] = gate
    packed[(idx // 32) * 64 + 32 + idx % 32] = up
    qf, scale = _quant_ref(packed)
    Wd = qf.to(torch.float32) * scale[:, None]
    xe = O.rms_norm(x, g, 1e-6)
    gd, ud = Wd[(idx // 32) * 64 + idx % 32], Wd[(idx // 32) * 64 + 32 + idx % 32]
    exp = O.silu(gd @ xe) * (ud @ xe)
    y = torch.zeros((I,), dtype=dtype, device="cuda")
    w8, sc, dx, dg = qf.view(torch.uint8).cuda(), scale.cuda(), x.to(dtype).cuda(), g.to(dtype).cuda()
    torch.cuda.synchronize()
    chk(m._lib.svln_op_gemv_fp8(m._h, ptr(w8), ptr(sc), K, ptr(dx), ptr(dg), 1e-6, None, None, ptr(y), 2 * I, K, _lib.EPI_SWIGLU, None))
    assert_close(y, exp, dtype, "gemv fp8 swiglu")
    V = 5000
    Wv = q(rnd((V, K), 50, 0.05), dtype)
    Wv[77] = q(x * 0.02, dtype)
    qv, sv = _quant_ref(Wv)
    logits = (qv.to(torch.float32) * sv[:, None]) @ x
    tok = C.c_int32(-1)
    torch.cuda.synchronize()
    chk(m._lib.svln_op_gemv_fp8(m._h, ptr(qv.view(torch.uint8).cuda()), ptr(sv.cuda()), K, ptr(dx), None, 1e-6, None, None, None, V, K,
                                _lib.EPI_ARGMAX, C.byref(tok)))
    assert tok.value == int(torch.argmax(logits)) == 77, tok.value
    # fp32 engines refuse fp8 weights
    m32 = engine(TINY, torch.float32)
    rc = m32._lib.svln_op_gemv_fp8(m32._h, ptr(w8), ptr(sc), K, ptr(dx), None, 1e-6, None, None, ptr(y), 2 * I, K, _lib.EPI_NONE, None)
    assert rc != 0


@pytest.mark.parametrize("dtype", DTYPES)
def test_norms(dtype):
    m = engine(TINY, dtype)
    for rows, n in [(213, 3584), (729, 1152), (5, 144)]:
        x = q(rnd((rows, n), 21, 3.0) + 0.5, dtype)
        g, b = q(1 + rnd((n,), 22, 0.1), dtype), q(rnd((n,), 23, 0.1), dtype)
        dx, dg, db = x.to(dtype).cuda(), g.to(dtype).cuda(), b.to(dtype).cuda()
        y = torch.zeros_like(dx)
        torch.cuda.synchronize()
        chk(m._lib.svln_op_rmsnorm(m._h, ptr(dx), ptr(dg), ptr(y), rows, n, 1e-6))
        assert_close(y, O.rms_norm(x, g, 1e-6), dtype, f"rmsnorm {rows}x{n}")
        chk(m._lib.svln_op_layernorm(m._h, ptr(dx), ptr(dg), ptr(db), ptr(y), rows, n, 1e-6))
        assert_close(y, O.layer_norm(x, g, b, 1e-6), dtype, f"layernorm {rows}x{n}")


def _llm_attention_oracle(cfg, qkv_ctx, qkv_new, P):
    nq, nkv, hd = cfg.q_heads, cfg.kv_heads, cfg.head_dim
    allr = torch.cat([qkv_ctx, qkv_new], 0) if P > 0 else qkv_new
    L = allr.shape[0]
    pos = torch.arange(L)
    cos, sin = O.rope_cos_sin(pos, hd, cfg.rope_theta)
    qq = allr[:, : nq * hd].view(L, nq, hd)
    kk = allr[:, nq * hd: (nq + nkv) * hd].view(L, nkv, hd)
    vv = allr[:, (nq + nkv) * hd:].view(L, nkv, hd)
    qq = qq * cos[:, None] + O.rotate_half(qq) * sin[:, None]
    kk = kk * cos[:, None] + O.rotate_half(kk) * sin[:, None]
    g = nq // nkv
    T = qkv_new.shape[0]
    qn = qq[P:].transpose(0, 1).reshape(nkv, g, T, hd)
    s = torch.einsum("kgtd,ksd->kgts", qn, kk.transpose(0, 1)) * hd ** -0.5
    mask = torch.arange(L)[None, :] > (P + torch.arange(T))[:, None]
    s = s.masked_fill(mask[None, None], float("-inf"))
    p = torch.softmax(s, -1)
    o = torch.einsum("kgts,ksd->kgtd", p, vv.transpose(0, 1))
    return o.reshape(nq, T, hd).transpose(0, 1).reshape(T, nq * hd)


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("cfg,T,P,nsplit", [(TINY, 212, 0, 1), (TINY, 37, 300, 1), (TINY, 1, 777, 8), (TINY, 3, 130, 8),
                                            (TRUE1, 212, 800, 1), (TRUE1, 1, 1500, 8), (TRUE1, 376, 0, 1),
                                            (TRUE1, 1952, 0, 1), (TRUE1, 212, 1740, 1)])
def test_attention_llm(dtype, cfg, T, P, nsplit):
    """prefill (causal, bottom-right aligned over a cached context) and split-KV decode, incl. RoPE + paged KV append.
    (TRUE1, 1952, 0): the window-restart turn -- 107 row blocks per kv head, unsplit keys, no cached context;
    (TRUE1, 212, 1740): a steady turn late in the window (split-KV over 31 key pages)."""
    m = engine(cfg, dtype)
    ld = (cfg.q_heads + 2 * cfg.kv_heads) * cfg.head_dim
    ctx = q(rnd((max(P, 1), ld), 31), dtype)
    new = q(rnd((T, ld), 32), dtype)
    exp = _llm_attention_oracle(cfg, ctx[:P], new, P)
    dctx, dnew = ctx.to(dtype).cuda(), new.to(dtype).cuda()
    out = torch.zeros((T, cfg.q_heads * cfg.head_dim), dtype=dtype, device="cuda")
    torch.cuda.synchronize()
    chk(m._lib.svln_op_attention_llm(m._h, ptr(dnew), ld, T, P, ptr(dctx), P, ptr(out), cfg.q_heads * cfg.head_dim, nsplit))
    assert_close(out, exp, dtype, f"attention llm T{T} P{P} split{nsplit}")


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("cfg,F", [(TINY, 1), (TINY, 3), (TRUE1, 1), (TRUE1, 2), (TRUE1, 9)])      # F = 9: the window-restart batch (unsplit keys)
def test_attention_vit(dtype, cfg, F):
    m = engine(cfg, dtype)
    S, Hv, nh, hd = cfg.v_tokens, cfg.v_hidden, cfg.v_heads, cfg.v_head_dim
    qkv = q(rnd((F * S, 3 * Hv), 41), dtype)
    x = qkv.view(F, S, 3, nh, hd)
    qq, kk, vv = (x[:, :, i].transpose(1, 2) for i in range(3))
    p = torch.softmax((qq @ kk.transpose(2, 3)) * hd ** -0.5, -1)
    exp = (p @ vv).transpose(1, 2).reshape(F * S, Hv)
    dq = qkv.to(dtype).cuda()
    out = torch.zeros((F * S, Hv), dtype=dtype, device="cuda")
    torch.cuda.synchronize()
    chk(m._lib.svln_op_attention_vit(m._h, ptr(dq), 3 * Hv, F, ptr(out), Hv))
    assert_close(out, exp, dtype, f"attention vit F{F}")


@pytest.mark.parametrize("dtype", DTYPES)
def test_pool_and_patchify(dtype):
    m = engine(TINY, dtype)
    cfg = TINY
    F = 2
    feat = q(rnd((F, cfg.v_tokens, cfg.hidden), 51), dtype)
    exp = O.pool_bilinear(cfg, feat)
    ref = torch.nn.functional.interpolate(feat.view(F, 27, 27, -1).permute(0, 3, 1, 2), size=[14, 14], mode="bilinear")
    assert torch.allclose(exp, ref.permute(0, 2, 3, 1).reshape(F, 196, -1), atol=1e-5)
    din = feat.to(dtype).cuda()
    out = torch.zeros((F, cfg.pool_tokens, cfg.hidden), dtype=dtype, device="cuda")
    torch.cuda.synchronize()
    chk(m._lib.svln_op_pool(m._h, ptr(din), ptr(out), F))
    assert_close(out, exp, dtype, "pool")
    pix = rnd((F, 3, 384, 384), 52)
    kp = 592
    p_, side = cfg.v_patch, cfg.v_side
    e = pix[:, :, : side * p_, : side * p_].reshape(F, 3, side, p_, side, p_).permute(0, 2, 4, 1, 3, 5).reshape(F * side * side, 588)
    e = torch.cat([e, torch.zeros(e.shape[0], kp - 588)], 1)
    dpix = pix.cuda()
    out = torch.ones((F * side * side, kp), dtype=dtype, device="cuda")
    torch.cuda.synchronize()
    chk(m._lib.svln_op_patchify(m._h, ptr(dpix), ptr(out), F))
    assert_close(out, q(e, dtype), dtype, "patchify")


@pytest.mark.parametrize("dtype", DTYPES)
def test_synth_weights_bit_exact(dtype):
    """device generator == numpy generator, element for element, incl. packed (qkv, gate/up) layouts"""
    m = engine(TINY, dtype)
    m.load_synthetic(99)
    names = ["model.layers.1.mlp.gate_proj.weight", "model.layers.0.mlp.up_proj.weight", "model.layers.1.self_attn.k_proj.weight",
             "model.layers.0.self_attn.v_proj.bias", "model.norm.weight", "lm_head.weight",
             W.VT + "embeddings.patch_embedding.weight", W.VT + "encoder.layers.1.self_attn.v_proj.weight"]
    specs = {s.name: s for s in W.tensor_specs(TINY)}
    for n in names:
        exp = W.synth_tensor(specs[n], 99)
        got = m.get_tensor(n)
        assert np.array_equal(got, exp), n
