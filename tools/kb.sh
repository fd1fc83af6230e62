set -e
cd /tmp && export TMPDIR=/tmp
rm -rf $GRAFT_REPO_ROOT/gpurun_out/kb && mkdir -p $GRAFT_REPO_ROOT/gpurun_out/kb
cd $GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --output-format csv -d gpurun_out/kb -- python3 tools/kbench.py gemm 12 > gpurun_out/kb.log 2>&1
python3 profiles/analyze_trace.py $(ls gpurun_out/kb/*/*kernel_trace.csv | head -1) 60
