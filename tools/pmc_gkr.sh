#!/bin/bash
# PMC counters of the GKR bookkeeping / wiring kernels (depth 8, width 2^20): separate rocprofv3 --pmc passes, --kernel-trace only.
set -u
export TMPDIR=/tmp
OUT=$PWD/gpurun_out/pmc_gkr
rm -rf $OUT; mkdir -p $OUT
for set in "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU" "SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS SQ_INSTS_SALU" "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_VMEM" "GRBM_GUI_ACTIVE" "TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_PENDING_STALL_CYCLES_sum" "TCC_HIT_sum TCC_MISS_sum TCC_EA_RDREQ_sum" "FETCH_SIZE"; do
  tag=$(echo $set | tr ' ' '_' | cut -c1-40)
  timeout -k 10 150 rocprofv3 --kernel-trace --pmc $set --output-format csv -d $OUT/$tag -- python3 tools/prof_gkr.py 20 8 > $OUT/$tag.log 2>&1 || { echo "pass $set failed"; tail -3 $OUT/$tag.log; }
done
python3 - <<'PY'
import csv, glob, collections, os
out = os.path.join(os.environ.get("GRAFT_REPO_ROOT", "."), "gpurun_out", "pmc_gkr")
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(out + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        kn = r["Kernel_Name"]
        if any(t in kn for t in ("k_gkr_phase", "k_gkr_wiring_eval", "k_gkr_forward")):
            name = kn.split("(")[0].replace("void zk::", "").replace("zk::", "")
            agg[name][r["Counter_Name"]].append(float(r["Counter_Value"]))
            agg[name]["VGPR"].append(float(r.get("VGPR_Count", 0) or 0))
            agg[name]["LDS"].append(float(r.get("LDS_Block_Size", 0) or 0))
for name, d in sorted(agg.items()):
    print(name)
    for k, v in sorted(d.items()):
        print(f"   {k:32s} avg {sum(v)/len(v):.5g}  (n={len(v)})")
PY
