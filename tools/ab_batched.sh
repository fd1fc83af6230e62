# A/B of engine knobs on the 8-env batched pass of bench.py (one box, alternating):  bash tools/ab_batched.sh <rounds> <spec> <spec> ...
set -e
R=$1; shift
cd $GRAFT_REPO_ROOT
for i in $(seq 1 $R); do
  for SPEC in "$@"; do
    L=${SPEC%%,*}
    ENVS=$(echo "$SPEC" | cut -s -d, -f2- | tr ',' ' ')
    env SVLN_LIB=$GRAFT_REPO_ROOT/$L $ENVS python3 bench.py --steps 16 --warmup 4 --no-cpu-baseline --no-feature-cache-pass --no-prune-pass --no-fp8-pass > gpurun_out/ab_tmp.json 2> gpurun_out/ab_tmp.err
    python3 -c "
import json
d = json.load(open('gpurun_out/ab_tmp.json'))
print('$SPEC', 'value', d['value'], 'batched', d['batched_envs']['value'], d['batched_envs']['ms_per_lockstep_turn'], flush=True)"
  done
done
