# A/B of builds / knobs of the engine on ONE box, alternating (methodology rule: never rank builds across boxes):
#   bash tools/ab_lib.sh <rounds> <spec> <spec> ...      spec = path/to/lib.so[,ENV=value[,ENV=value]]
# build_ab/libA.so = a baseline built from another commit with tools/build_ref_lib.sh
set -e
R=$1; shift
cd $GRAFT_REPO_ROOT
for i in $(seq 1 $R); do
  for SPEC in "$@"; do
    L=${SPEC%%,*}
    ENVS=$(echo "$SPEC" | cut -s -d, -f2- | tr ',' ' ')
    env SVLN_LIB=$GRAFT_REPO_ROOT/$L $ENVS python3 bench.py --steps ${AB_STEPS:-20} --warmup 5 --no-cpu-baseline --no-feature-cache-pass --no-prune-pass --no-fp8-pass --no-batched-pass > gpurun_out/ab_tmp.json 2> gpurun_out/ab_tmp.err
    python3 - <<PY
import json
d = json.load(open("gpurun_out/ab_tmp.json"))
t = d["turn_ms"]
steady = sorted(x for x in t if x < 23.5)
print("$SPEC", "value", d["value"], "p50", d["p50_ms_per_turn"], "phases", d["phase_ms_per_turn"], "steady median", steady[len(steady) // 2] if steady else None,
      "restart", [x for x in t if x > 40], flush=True)
PY
  done
done
