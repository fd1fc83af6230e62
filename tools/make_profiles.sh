# One GPU-box call that regenerates the round's committed profile artefacts under gpurun_out/<round>/ (copy into profiles/ afterwards):
#   bench plain (incl. the measured CPU baseline), bench under rocprofv3 --kernel-trace --stats, by-shape + per-phase summaries,
#   and the HBM-traffic PMC passes (FETCH_SIZE / WRITE_SIZE in separate runs, as the microarch guide prescribes) of the roofline kernel.
set -e
R=${ROUND:-r03}
OUT=$GRAFT_REPO_ROOT/gpurun_out/$R
rm -rf $OUT && mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
echo "== bench plain"; python3 bench.py --steps 20 --warmup 5 > $OUT/${R}_bench_plain.json 2> $OUT/bench_plain.err
echo "== bench under rocprofv3"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline > $OUT/${R}_bench_under_rocprof.json 2> $OUT/bench_prof.err
T=$(ls $OUT/trace/*/*kernel_trace.csv | head -1)
python3 profiles/analyze_trace.py $T 90 > $OUT/${R}_bench_kernel_trace_by_shape.txt
python3 profiles/turn_breakdown.py $T 6 > $OUT/${R}_steady_turn_phases.txt
python3 profiles/turn_breakdown.py $T 8 >> $OUT/${R}_steady_turn_phases.txt
cp $(ls $OUT/trace/*/*kernel_stats.csv | head -1) $OUT/${R}_bench_kernel_stats.csv
rm -rf $OUT/trace
for C in FETCH_SIZE WRITE_SIZE; do
  echo "== pmc $C"
  rocprofv3 --pmc $C --output-format csv -d $OUT/pmc_$C -- python3 tools/kbench.py gemv 5 > $OUT/pmc_$C.log 2>&1
  F=$(ls $OUT/pmc_$C/*/*counter_collection.csv | head -1)
  head -1 $F > $OUT/${R}_gemv_pmc_$C.csv; grep "gemv_kernel" $F >> $OUT/${R}_gemv_pmc_$C.csv
  rm -rf $OUT/pmc_$C
done
python3 - <<PY
import csv, json, os
R, OUT = "$R", "$OUT"
def mean(c):
    rows = [r for r in csv.DictReader(open(f"{OUT}/{R}_gemv_pmc_{c}.csv")) if "Li3E" in r["Kernel_Name"] and r["Grid_Size"] == str(1184 * 256)]
    return sum(float(r["Counter_Value"]) for r in rows) / max(len(rows), 1), len(rows)
f, nf = mean("FETCH_SIZE"); w, nw = mean("WRITE_SIZE")
alg = 2 * 18944 * 3584 * 2
d = {"kernel": "gemv_kernel<bf16, EPI_SWIGLU> N=37888 K=3584 (decode gate/up projection)",
     "command": "rocprofv3 --pmc FETCH_SIZE|WRITE_SIZE (separate passes) -- python tools/kbench.py gemv 5",
     "launches": [nf, nw], "FETCH_SIZE_KB_per_launch": f, "WRITE_SIZE_KB_per_launch": w,
     "correction": "gfx950: FETCH_SIZE reports exactly half of a wide coalesced 16 B/lane stream (MI355X_MICROARCH.md, HBM) -> x2; WRITE_SIZE exact",
     "traffic_bytes_per_launch": 2 * f * 1024 + w * 1024, "algorithmic_bytes_per_launch": alg,
     "ratio": (2 * f * 1024 + w * 1024) / alg}
json.dump(d, open(f"{OUT}/{R}_gemv_swiglu_pmc.json", "w"), indent=1)
print(json.dumps(d))
PY
ls -la $OUT; cat $OUT/${R}_steady_turn_phases.txt; python3 -c "
import json; d=json.load(open('$OUT/${R}_bench_plain.json')); print({k: d[k] for k in ['value','ms_per_step','p50_ms_per_turn','phase_ms_per_turn','generate_boundary','roofline','roofline_prefill_gemm','cpu_baseline']})"
