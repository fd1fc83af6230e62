"""Build-time guards on the generated gfx950 assembly (hipcc cross-compiles without a GPU).

1. The inline-asm LDS reads of the attention kernels, below.
2. The 8-phase GEMM schedule (gemm.hip: p8_mainloop), at the end of this file: its synchronisation is placed by COUNT (half-tiles of
   LDS-DMA in flight behind `s_waitcnt vmcnt(6)`, fragment reads behind `lgkmcnt`), so the test pins the counts the derivation assumes.


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
def test_gemm_8phase_loop_has_the_instruction_counts_its_waits_assume():
    """p8_mainloop orders LDS-DMA against the fragment reads with COUNTED waits: per wave two DMA instructions per phase, `vmcnt(6)` at
    phases 4 and 8 (three half-tiles stay in flight), 12 / 4 / 8 / 0 asm `ds_read_b128` per phase with `lgkmcnt(8)` retiring the four B reads
    of phases 1 and 5 before the barrier.  A compiler that adds, merges or re-times any of these silently breaks the derivation, so the
    generated code of every 8-phase instantiation is checked: no scratch, <= 256 VGPRs (two waves per SIMD), 48 fragment reads up to the last MFMA, 19 barriers
    (1 + the stagger + 16 + the closing one; one more in front of an LDS-staged epilogue), the MFMA count of the form, two `lgkmcnt(8)`, 30-32 DMA instructions, and 6 and 0 as the only
    `vmcnt` immediates (hipcc adds no drain of its own in front of the asm fragment reads)."""
    with tempfile.TemporaryDirectory() as d:
        out = os.path.join(d, "gemm.s")
        r = subprocess.run(["hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "--cuda-device-only", "-S", "-o", out, GEMM_SRC],
                           capture_output=True, text=True, timeout=900)
        assert r.returncode == 0, r.stderr[-2000:]
        text = open(out).read().splitlines()
    funcs, cur = {}, None
    for ln in text:
        m = re.match(r"^(_ZN\S*gemm_glds_kernel\S*TileCfgILi256ELi256ELi2ELi4ELi128ELb0ELi2ELb[01]ELi1ELb1EEE\S*):", ln)
        if m:
            cur = m.group(1)
            funcs[cur] = []
        elif cur is not None:
            funcs[cur].append(ln)
            if ln.startswith("\t.end_amdhsa_kernel"):
                cur = None
    assert len(funcs) >= 12, list(funcs)                      # 4 epilogues x (16x16x32, its two-slice form, 32x32x16)
    for name, lines in funcs.items():
        code = [ln.split(";")[0].strip() for ln in lines]
        code = [c for c in code if c and not c.startswith(".") and not c.endswith(":")]
        form32 = "ELb1ELi1ELb1EEE" in name                     # TileCfg<..., ILV = true, 1, P8 = true> = the 32x32x16 form
        meta = "\n".join(lines)
        assert re.search(r"\.amdhsa_private_segment_fixed_size 0\b", meta), name
        assert int(re.search(r"\.amdhsa_next_free_vgpr (\d+)", meta).group(1)) <= 256, name
        assert not [c for c in code if c.startswith("scratch_")], name
        n_mfma = len([c for c in code if c.startswith("v_mfma_f32_32x32x16_bf16" if form32 else "v_mfma_f32_16x16x32_bf16")])
        assert n_mfma == (64 if form32 else 128), (name, n_mfma)
        # the main loop ends at the last MFMA; behind it the unsplit 16x16x32 kernels stage their stores through LDS (one more barrier, then
        # plain ds_write_b16 / ds_read_b128 of the wave's own tile: 16 reads, 8 for the SwiGLU epilogue's 32 output columns)
        mfma = "v_mfma_f32_32x32x16_bf16" if form32 else "v_mfma_f32_16x16x32_bf16"
        last = max(i for i, c in enumerate(code) if c.startswith(mfma))
        loop, tail = code[:last + 1], code[last + 1:]
        assert len([c for c in loop if c.startswith("ds_read_b128")]) == 48, name
        staged_reads = len([c for c in tail if c.startswith("ds_read_b128")])
        assert staged_reads in (0, 8, 16), name
        assert len([c for c in code if c.startswith("s_barrier")]) == 19 + (1 if staged_reads else 0), name
        vm = [int(x) for c in code for x in re.findall(r"s_waitcnt vmcnt\((\d+)\)", c)]
        assert set(vm) <= {0, 6} and vm.count(6) >= 3, (name, vm)
        assert len([c for c in code if c.startswith("s_waitcnt lgkmcnt(8)")]) == 2, name
        assert 30 <= len([c for c in code if c.startswith("global_load_lds_dwordx4")]) <= 32, name
