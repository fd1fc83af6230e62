# kernels of ONE turn of a short bench run, grouped by (kernel, grid), prefill phase only:  bash tools/turn_kernels.sh <turn index>
set -e
TURN=${1:-16}
PHASE=${2:-prefill}
OUT=$GRAFT_REPO_ROOT/gpurun_out/tk
rm -rf $OUT && mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --output-format csv -d $OUT -- python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-feature-cache-pass --no-prune-pass --no-fp8-pass --no-batched-pass > $OUT/bench.json 2> $OUT/bench.err
python3 - <<PY
import csv, glob, collections
PHASE = "$PHASE"
rows = list(csv.DictReader(open(glob.glob("$OUT/*/*kernel_trace.csv")[0])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
idx = [i for i, r in enumerate(rows) if "patchify" in r["Kernel_Name"]]
seg = rows[idx[$TURN]:idx[$TURN + 1]]
ph, agg = "vision", collections.OrderedDict()
for r in seg:
    nm = r["Kernel_Name"]
    if ph == "vision" and "gather_rows" in nm: ph = "prefill"
    if ph == "prefill" and "gemv" in nm: ph = "decode"
    if ph != PHASE: continue
    k = (nm.split("GLOBAL__N_1")[-1][:62], r["Grid_Size_X"], r["Workgroup_Size_X"])
    a = agg.setdefault(k, [0, 0.0]); a[0] += 1; a[1] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
tot = sum(v[1] for v in agg.values())
print(f"turn $TURN {PHASE}: {tot / 1e3:.3f} ms in {sum(v[0] for v in agg.values())} kernels")
for k, v in sorted(agg.items(), key=lambda kv: -kv[1][1])[:16]:
    print(f"  {k[0]:64s} wgs {int(k[1]) // int(k[2]):5d} x{k[2]:>4s}  calls {v[0]:3d}  avg {v[1] / v[0]:8.2f} us  tot {v[1] / 1e3:7.3f} ms")
PY
rm -rf $OUT
