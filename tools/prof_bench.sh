# rocprofv3 kernel trace of a short bench run -> by-shape summary + per-phase breakdown (run on the GPU box)
set -e
OUT=$GRAFT_REPO_ROOT/gpurun_out/prof
rm -rf $OUT && mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -- python3 bench.py --steps 16 --warmup 4 --no-cpu-baseline --no-feature-cache-pass --no-prune-pass --no-fp8-pass --no-batched-pass > gpurun_out/prof_bench.json 2> gpurun_out/prof_bench.err
T=$(ls $OUT/*/*kernel_trace.csv | head -1)
python3 profiles/analyze_trace.py $T 70 > gpurun_out/prof_by_shape.txt
python3 profiles/turn_breakdown.py $T 6 gaps > gpurun_out/prof_phases.txt
python3 profiles/turn_breakdown.py $T 8 >> gpurun_out/prof_phases.txt
cp $(ls $OUT/*/*kernel_stats.csv | head -1) gpurun_out/prof_kernel_stats.csv
rm -rf $OUT
cat gpurun_out/prof_phases.txt; head -50 gpurun_out/prof_by_shape.txt
