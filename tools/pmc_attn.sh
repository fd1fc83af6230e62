set -e
OUT=$GRAFT_REPO_ROOT/gpurun_out/pmc_attn
rm -rf $OUT && mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_LDS --output-format csv -d $OUT -- python3 tools/kbench.py attn 6 > gpurun_out/pmc_attn.log 2>&1
python3 - <<'PY'
import csv, glob, collections, os
f = glob.glob(os.environ["GRAFT_REPO_ROOT"] + "/gpurun_out/pmc_attn/*/*counter_collection.csv")[0]
agg = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter()
for r in csv.DictReader(open(f)):
    k = r["Kernel_Name"]
    if "attn" not in k and "rope" not in k: continue
    key = (k.split("svln12_GLOBAL__N_1")[-1][:40], r.get("Grid_Size", ""), r.get("Workgroup_Size", ""))
    agg[key][r["Counter_Name"]] += float(r["Counter_Value"]); 
    if r["Counter_Name"] == "SQ_WAVE_CYCLES": cnt[key] += 1
for key, c in agg.items():
    n = max(cnt[key], 1)
    wc = c["SQ_WAVE_CYCLES"] or 1
    print(key, "calls", n, {k: round(v / n) for k, v in c.items()})
    print("    frac of wave cycles: wait_any %.2f wait_inst %.2f active %.2f lds_wait %.2f | lds conflict/idx %.2f | mfma_busy_cycles/ (4*wave_cycles) %.3f" % (
        c["SQ_WAIT_ANY"] / wc, c["SQ_WAIT_INST_ANY"] / wc, c["SQ_ACTIVE_INST_ANY"] / wc, c["SQ_WAIT_INST_LDS"] / wc,
        c["SQ_LDS_BANK_CONFLICT"] / max(c["SQ_LDS_IDX_ACTIVE"], 1), c["SQ_VALU_MFMA_BUSY_CYCLES"] / (4 * wc)))
PY
rm -rf $OUT
