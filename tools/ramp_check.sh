# does a kernel run slower right after the between-turn host gap?  per-occurrence durations of the one-frame ViT qkv GEMM (27 per turn)
set -e
OUT=$GRAFT_REPO_ROOT/gpurun_out/ramp
rm -rf $OUT && mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --output-format csv -d $OUT -- python3 bench.py --steps 12 --warmup 4 --no-cpu-baseline --no-feature-cache-pass --no-prune-pass --no-fp8-pass --no-batched-pass > gpurun_out/ramp_bench.json 2> gpurun_out/ramp_bench.err
python3 - <<PY
import csv, glob
rows = list(csv.DictReader(open(glob.glob("$OUT/*/*kernel_trace.csv")[0])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
idx = [i for i, r in enumerate(rows) if "patchify" in r["Kernel_Name"]]
for turn in (6, 7, 9):
    seg = rows[idx[turn]:idx[turn + 1]]
    q = [(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3 for r in seg if "gemm_glds" in r["Kernel_Name"] and r["Grid_Size_X"] == str(162 * 256)]
    print("turn", turn, "ViT qkv GEMM per layer:", " ".join(f"{x:.1f}" for x in q))
    g = [(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3 for r in seg if "gemv_kernel" in r["Kernel_Name"] and "Li3E" in r["Kernel_Name"]]
    print("   decode gate/up GEMV first 12:", " ".join(f"{x:.1f}" for x in g[:12]), " last 4:", " ".join(f"{x:.1f}" for x in g[-4:]))
    p = [(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3 for r in seg if "Li256ELi128" in r["Kernel_Name"] and r["Grid_Size_X"] == str(256 * 512)]
    print("   prefill gate/up main:", " ".join(f"{x:.1f}" for x in p))
PY
rm -rf $OUT
