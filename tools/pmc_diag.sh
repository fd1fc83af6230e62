# Where a GEMM product spends its cycles: memory-path counters of the main gemm_glds launch (one kbench run per counter group; rocprofv3
# --pmc alone, as the pool requires).  Two TA / TD groups of the first version are left out:
#   "TA_BUSY_avr TA_FLAT_READ_LDS_WAVEFRONTS_sum TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum"
#   "TD_TD_BUSY_sum TD_TC_STALL_sum TA_TA_BUSY_sum TA_FLAT_READ_WAVEFRONTS_sum"
# rocprofv3 refused them when it built the counter configuration -- rocprofiler_create_counter_config: "error code 38: Request exceeds
# the capabilities of the hardware to collect" -- and aborted the process (signal 6) while the engine was being created; the pass did
# not hang, it died before any kernel ran (log: profiles/r03_pmc_diag_TA_group_refused.log).  Four TA counters (plus a derived _avr
# one) in ONE pass are more than the TA block's counter slots on gfx950 take; such a group has to be split into passes of one or two
# counters.  Not done: the TCP / TCC groups below answered the question, and the refused groups were not run again.  bash tools/pmc_diag.sh "<M,N,K,epi,force_cfg,force_split>" [...]   -> gpurun_out/pmc_diag.txt
set -e
OUT=$GRAFT_REPO_ROOT/gpurun_out/pmc_diag
rm -rf $OUT && mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
CGRP=(
 "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE"
 "SQ_INST_CYCLES_VMEM_RD SQ_INSTS_VMEM_RD SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_DATA_FIFO_FULL SQ_LDS_CMD_FIFO_FULL SQ_INSTS_LDS SQ_BUSY_CYCLES"
 "TCP_TCC_READ_REQ_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_PENDING_STALL_CYCLES_sum TCP_TCR_TCP_STALL_CYCLES_sum"
 "TCP_TOTAL_ACCESSES_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum TCP_READ_TAGCONFLICT_STALL_CYCLES_sum TCP_GATE_EN1_sum"
 "TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum TCC_REQ_sum"
 "TCC_EA0_RDREQ_LEVEL_sum TCC_CYCLE_sum TCC_EA0_RDREQ_DRAM_CREDIT_STALL_sum TCC_TAG_STALL_sum"
)
: > $GRAFT_REPO_ROOT/gpurun_out/pmc_diag.txt
for SH in "$@"; do
  export KBENCH_SHAPES="$SH"
  g=0
  for G in "${CGRP[@]}"; do
    rocprofv3 --pmc $G --output-format csv -d $OUT/g$g -- python3 tools/kbench.py gemm 6 > $OUT/g$g.log 2>&1 || echo "group $g failed ($G)" >> $GRAFT_REPO_ROOT/gpurun_out/pmc_diag.txt
    g=$((g+1))
  done
  python3 - <<PY >> $GRAFT_REPO_ROOT/gpurun_out/pmc_diag.txt
import csv, glob, collections
print("== shape $SH")
for d in sorted(glob.glob("$OUT/g*/")):
    fs = glob.glob(d + "*/*counter_collection.csv")
    if not fs: continue
    agg = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter()
    for r in csv.DictReader(open(fs[0])):
        if "gemm_glds" not in r["Kernel_Name"]: continue
        key = r["Kernel_Name"].split("GLOBAL__N_1")[-1][:60] + " grid " + r["Grid_Size"]
        agg[key][r["Counter_Name"]] += float(r["Counter_Value"]); cnt[(key, r["Counter_Name"])] += 1
    for key, cs in agg.items():
        print("  ", key)
        for c, v in cs.items():
            print(f"      {c:44s} {v / cnt[(key, c)]:16.1f} per launch")
PY
  rm -rf $OUT/g*
done
cat $GRAFT_REPO_ROOT/gpurun_out/pmc_diag.txt
