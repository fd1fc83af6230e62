# GEMV kernels with HBM-cold weights (rotated copies) and with one hot copy (<= 256 MB products then sit in the Infinity Cache)
set -e
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
for mode in cold hot; do
  rm -rf gpurun_out/kbv_$mode && mkdir -p gpurun_out/kbv_$mode
  if [ $mode = hot ]; then export KBENCH_HOT=1; fi
  rocprofv3 --kernel-trace --output-format csv -d gpurun_out/kbv_$mode -- python3 tools/kbench.py gemv 12 > gpurun_out/kbv_$mode.log 2>&1
  echo "== $mode"; python3 profiles/analyze_trace.py $(ls gpurun_out/kbv_$mode/*/*kernel_trace.csv | head -1) 12 | grep -i "gemv"
  rm -rf gpurun_out/kbv_$mode
done
