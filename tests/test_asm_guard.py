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
