"""Build-time guard for the inline-asm LDS reads of the attention kernels.

attention.hip issues the Vt fragment reads as `asm volatile("ds_read_b64 %0, ...")` with an "=v" output and retires them later with a
separate asm `s_waitcnt lgkmcnt(N)` naming the destinations.  hipcc does not know the read is still in flight: if register allocation
ever copied, spilled or reused a destination between the read and its wait, the MFMA would consume stale registers with no diagnostic.
This test compiles attention.hip to gfx950 assembly (hipcc cross-compiles without a GPU) and checks, for every attention kernel, that no
instruction touches a ds_read_b64 destination between the read and the wait that retires it (LDS operations retire in order, so a
`lgkmcnt(k)` retires everything but the k youngest)."""
import os
import re
import shutil
import subprocess
import tempfile

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "streamvln_amd", "csrc", "attention.hip")
REG = re.compile(r"\bv\[(\d+):(\d+)\]|\bv(\d+)\b")


def _regs(text):
    out = set()
    for m in REG.finditer(text):
        if m.group(3) is not None:
            out.add(int(m.group(3)))
        else:
            out.update(range(int(m.group(1)), int(m.group(2)) + 1))
    return out


def _check_function(name, lines):
    """returns (asm ds_read_b64 count, violations)"""
    pending = []                      # LDS operations in flight, oldest first: set of destination VGPRs (empty for writes / compiler reads)
    n_reads, bad = 0, []
    in_asm = False
    for ln in lines:
        t = ln.strip()
        if t.startswith(";;#ASMSTART"):
            in_asm = True
            continue
        if t.startswith(";;#ASMEND"):
            in_asm = False
            continue
        if not t or t.startswith(";") or t.startswith(".") or t.endswith(":"):
            continue
        code = t.split(";")[0]
        op = code.split()[0]
        m = re.search(r"lgkmcnt\((\d+)\)", code)
        if op == "s_waitcnt" and m:
            k = int(m.group(1))
            pending = pending[len(pending) - k:] if k else []
            continue
        if op == "s_waitcnt" and "lgkmcnt" not in code and "vmcnt" not in code and "expcnt" not in code:
            pending = []              # s_waitcnt <imm>: treat as a full drain
            continue
        if op.startswith("ds_"):
            if in_asm and op == "ds_read_b64":
                dst = _regs(code.split(",")[0])
                n_reads += 1
                # the address register may not be one of the in-flight destinations either
                touched = _regs(code.split(",", 1)[1]) if "," in code else set()
                for p in pending:
                    if p & (touched | dst):
                        bad.append((name, code))
                pending.append(dst)
            else:
                touched = _regs(code)
                for p in pending:
                    if p & touched:
                        bad.append((name, code))
                pending.append(set())
            continue
        touched = _regs(code)
        for p in pending:
            if p & touched:
                bad.append((name, code))
    return n_reads, bad


@pytest.mark.skipif(shutil.which("hipcc") is None, reason="needs hipcc")
def test_attention_asm_lds_reads_are_not_touched_before_their_wait():
    with tempfile.TemporaryDirectory() as d:
        out = os.path.join(d, "attn.s")
        r = subprocess.run(["hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "--cuda-device-only", "-S", "-o", out, SRC],
                           capture_output=True, text=True, timeout=600)
        assert r.returncode == 0, r.stderr[-2000:]
        text = open(out).read().splitlines()
    funcs, cur = {}, None
    for ln in text:
        m = re.match(r"^(_ZN\S*attn_kernel\S*):", ln)
        if m:
            cur = m.group(1)
            funcs[cur] = []
        elif ln.startswith("\t.end_amdhsa_kernel") or ln.startswith(".Lfunc_end"):
            cur = None
        elif cur is not None:
            funcs[cur].append(ln)
    assert len(funcs) >= 8, list(funcs)
    total = 0
    for name, lines in funcs.items():
        n, bad = _check_function(name, lines)
        total += n
        assert not bad, bad[:5]
        spill = [ln for ln in lines if "scratch_" in ln or "buffer_store_dword" in ln and "offen" in ln]
        assert not spill, (name, spill[:3])            # no scratch traffic in the attention kernels
    assert total >= 32                                  # the bf16 instantiations carry the asm reads


GEMM_SRC = os.path.join(ROOT, "streamvln_amd", "csrc", "gemm.hip")


@pytest.mark.skipif(shutil.which("hipcc") is None, reason="needs hipcc")
def test_gemm_register_staged_weights_are_not_touched_between_load_and_lds_store():
    """gemm_glds_kernel<..., WR = true> (M <= 256 products): the weight chunks are loaded HBM -> VGPR by inline-asm `global_load_dwordx4`
    several stages ahead and copied to LDS by an inline-asm `ds_write_b128`; hipcc does not know the load is in flight.  Between a load
    and the ds_write that consumes its destination there must be no other instruction that reads or writes those registers (an earlier
    form of the kernel, whose waits carried the registers as "+v" operands, got copies of them placed BEFORE the wait), and at least one
    asm `s_waitcnt vmcnt`."""
    with tempfile.TemporaryDirectory() as d:
        out = os.path.join(d, "gemm.s")
        r = subprocess.run(["hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "--cuda-device-only", "-S", "-o", out, GEMM_SRC],
                           capture_output=True, text=True, timeout=1200)
        assert r.returncode == 0, r.stderr[-2000:]
        text = open(out).read().splitlines()
    funcs, cur = {}, None
    for ln in text:
        m = re.match(r"^(_ZN\S*gemm_glds_kernel\S*):", ln)
        if m:
            cur = m.group(1)
            funcs[cur] = []
        elif ln.startswith("\t.end_amdhsa_kernel") or ln.startswith(".Lfunc_end"):
            cur = None
        elif cur is not None:
            funcs[cur].append(ln)
    checked = 0
    for name, lines in funcs.items():
        loads = [ln for ln in lines if re.search(r"^\s*global_load_dwordx4\s+v\[", ln)]
        if not loads:
            continue                                    # not a register-staged instantiation
        checked += 1
        wregs = set()
        for ln in loads:
            wregs |= _regs(ln.split(",")[0])
        assert len(wregs) == 32, (name, sorted(wregs))  # 4 stages x 2 chunks x 4 dwords: the loads always land in the same registers
        # Inside the stage loop (hipcc annotates the blocks of a loop) a staged chunk lives in its registers from the asm load to the asm
        # ds_write; between the ds_write and the next load of the slot the registers are dead and hipcc may use them as scratch (it
        # computes the next load's address in them).  What must never appear in the loop is a COPY of one of them (v_mov / v_accvgpr /
        # scratch store with such a source): that is how a value in flight gets duplicated before it has landed.
        in_loop = in_asm = False
        n_store = 0
        for ln in lines:
            t = ln.strip()
            if t.startswith(";;#ASMSTART"):
                in_asm = True; continue
            if t.startswith(";;#ASMEND"):
                in_asm = False; continue
            if re.match(r"^(\.LBB\w+:|; %bb\.\d+:)", t):
                in_loop = "in Loop:" in t or "Loop Header" in t
                continue
            if not t or t.startswith(";") or t.startswith("."):
                continue
            code = t.split(";")[0]
            op = code.split()[0]
            assert not op.startswith("scratch_"), (name, code)
            if not in_loop:
                continue
            if in_asm:
                if op == "ds_write_b128":
                    assert _regs(code.split(",", 1)[1]) <= wregs, (name, code)
                    n_store += 1
                continue
            if op.startswith("v_mov") or op.startswith("v_accvgpr") or op.startswith("buffer_store") or op.startswith("global_store"):
                srcs = _regs(code.split(",", 1)[1]) if "," in code else set()
                assert not (srcs & wregs), (name, code)
        assert n_store >= 8, (name, n_store)
    assert checked >= 4, checked
