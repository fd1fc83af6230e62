# MFMA utilisation of the GEMM products of the path (one kbench run per product and pass; gfx950 has no derived-counter formulas in ROCm 7.2):
#   pass 1  rocprofv3 --kernel-trace            -> durations
#   pass 2  rocprofv3 --pmc <SQ / GRBM counters> -> MFMA-busy cycles, wave-cycle split
# writes gpurun_out/<round>_gemm_mfma_pmc.json  (copy into profiles/)
set -e
R=${ROUND:-r03}
OUT=$GRAFT_REPO_ROOT/gpurun_out/pmc_gemm
rm -rf $OUT && mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
i=0
for SH in "212,37888,3584,3,0,0" "212,4608,3584,0,0,0" "212,3584,3584,0,0,0" "212,3584,18944,0,0,0" "729,3456,1152,0,0,0" "729,1152,1152,0,0,0" "729,4304,1152,1,0,0" "729,1152,4304,0,0,0" "1952,37888,3584,3,0,0" "1952,3584,18944,0,0,0" "6561,4304,1152,1,0,0"; do
  export KBENCH_SHAPES="$SH"
  rocprofv3 --kernel-trace --output-format csv -d $OUT/t$i -- python3 tools/kbench.py gemm 6 > $OUT/t$i.log 2>&1
  rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE --output-format csv -d $OUT/p$i -- python3 tools/kbench.py gemm 6 > $OUT/p$i.log 2>&1
  echo "$SH" > $OUT/shape$i.txt
  echo "done $SH"
  i=$((i+1))
done
python3 - <<PY
import csv, glob, json, collections
OUT, R, n = "$OUT", "$R", $i
prods = []
for i in range(n):
    M, N, K, epi, _, _ = [int(v) for v in open(f"{OUT}/shape{i}.txt").read().strip().split(",")]
    tr = list(csv.DictReader(open(glob.glob(f"{OUT}/t{i}/*/*kernel_trace.csv")[0])))
    ours = [r for r in tr if "gemm_glds" in r["Kernel_Name"] or "splitk" in r["Kernel_Name"]]
    reps = 6
    us = sum(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in ours) / 1e3 / reps
    names = collections.Counter(r["Kernel_Name"].split("GLOBAL__N_1")[-1][:70] + " grid " + r["Grid_Size_X"] for r in ours)
    pm = list(csv.DictReader(open(glob.glob(f"{OUT}/p{i}/*/*counter_collection.csv")[0])))
    c = collections.defaultdict(float)
    for r in pm:
        if "gemm_glds" in r["Kernel_Name"]:
            c[r["Counter_Name"]] += float(r["Counter_Value"]) / reps
    wc = c["SQ_WAVE_CYCLES"] or 1.0
    flops = 2.0 * M * N * K
    prods.append({"shape": f"{M}x{N}x{K}", "epi": epi, "launches_per_product": {k: v // reps for k, v in names.items()},
                  "us_per_product": round(us, 1), "tflops": round(flops / us / 1e6, 1), "frac_of_2p5_PF": round(flops / us / 1e6 / 2500, 3),
                  "weight_TBps": round(N * K * 2 / us / 1e6, 2),
                  "mfma_busy_frac": round(c["SQ_VALU_MFMA_BUSY_CYCLES"] / (c["GRBM_GUI_ACTIVE"] / 8 * 256 * 4), 3) if c["GRBM_GUI_ACTIVE"] else None,
                  "wave_cycles_split": {"wait_any": round(c["SQ_WAIT_ANY"] / wc, 3), "wait_inst_any": round(c["SQ_WAIT_INST_ANY"] / wc, 3),
                                        "active_inst_any": round(c["SQ_ACTIVE_INST_ANY"] / wc, 3)}})
d = {"command": "per product: rocprofv3 --kernel-trace -- python3 tools/kbench.py gemm 6, then rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE -- the same (tools/pmc_gemm.sh); weights rotated through > 768 MB (HBM-cold); isolated launches with a host sync in between, so durations run 10-20 % above the same kernels inside a turn",
     "note": "mfma_busy_frac = SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE / 8 XCDs * 256 CUs * 4 SIMDs), counters of the main gemm_glds launches; us_per_product includes the split-K reduce launches; frac_of_2p5_PF = 2MNK / time / 2.5 PF",
     "products": prods}
json.dump(d, open(f"$GRAFT_REPO_ROOT/gpurun_out/{R}_gemm_mfma_pmc.json", "w"), indent=1)
for p in prods: print(p["shape"], p["us_per_product"], p["tflops"], p["mfma_busy_frac"], p["wave_cycles_split"])
PY
rm -rf $OUT
