# A/B of an environment knob inside the turn: bash tools/ab_env.sh VAR v1 v2 ...   (kernel-trace of a short bench per value)
set -e
VAR=$1; shift
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
for V in "$@"; do
  export $VAR=$V
  OUT=$GRAFT_REPO_ROOT/gpurun_out/ab_$V
  rm -rf $OUT && mkdir -p $OUT
  rocprofv3 --kernel-trace --output-format csv -d $OUT -- python3 bench.py --steps ${AB_STEPS:-12} --warmup 4 --no-cpu-baseline --no-feature-cache-pass --no-prune-pass --no-fp8-pass --no-batched-pass > $OUT/bench.json 2> $OUT/bench.err
  T=$(ls $OUT/*/*kernel_trace.csv | head -1)
  echo "== $VAR=$V  $(python3 -c "import json; d=json.load(open('$OUT/bench.json')); print(d['value'], d['p50_ms_per_turn'])")"
  python3 profiles/turn_breakdown.py $T 6 | head -3
  python3 profiles/turn_breakdown.py $T 8 | head -3
  rm -rf $OUT
done
