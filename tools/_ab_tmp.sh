set -e
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
for V in 0 4 104 6 106; do
  echo "=== SVLN_AB_VITGROUPS=$V"
  export SVLN_AB_VITGROUPS=$V
  timeout -k 10 300 python3 -m pytest tests/test_ops_gpu.py -q -k "attention_vit" 2>&1 | tail -3
  rm -rf gpurun_out/kbv && mkdir -p gpurun_out/kbv
  timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d gpurun_out/kbv -- python3 tools/kbench.py attn 40 > gpurun_out/kbv.log 2>&1
  python3 profiles/analyze_trace.py $(ls gpurun_out/kbv/*/*kernel_trace.csv | head -1) 12 | grep -i "attn\|vit_kv" || true
done
