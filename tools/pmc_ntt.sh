#!/bin/bash
# PMC counters of the 2^24-point NTT passes (k_ntt_pass): separate rocprofv3 --pmc passes with --kernel-trace only.
set -u
export TMPDIR=/tmp
OUT=$PWD/gpurun_out/pmc_ntt
rm -rf $OUT; mkdir -p $OUT
cat > /tmp/ntt_run.py <<'PY'
import sys, os
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import zk_amd
ctx = zk_amd.Context(zk_amd.BN254_FR, 0)
x = zk_amd.MultiLinearPolynomial.random(ctx, 24, 5, 0); y = zk_amd.MultiLinearPolynomial.alloc(ctx, 24)
print("ntt ms", zk_amd.bench_ntt(ctx, x, y, False, 3))
PY
for set in "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU" "SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS" "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR" "GRBM_GUI_ACTIVE" "FETCH_SIZE" "WRITE_SIZE"; do
  tag=$(echo $set | tr ' ' '_' | cut -c1-40)
  timeout -k 10 120 rocprofv3 --kernel-trace --pmc $set --output-format csv -d $OUT/$tag -- python3 /tmp/ntt_run.py > $OUT/$tag.log 2>&1 || { echo "pass $set failed"; tail -3 $OUT/$tag.log; }
done
python3 - <<'PY'
import csv, glob, collections, os
out = os.path.join(os.environ.get("GRAFT_REPO_ROOT", "."), "gpurun_out", "pmc_ntt")
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(out + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "k_ntt_pass" in r["Kernel_Name"]:
            name = r["Kernel_Name"].split("(")[0].replace("void zk::", "")
            agg[name][r["Counter_Name"]].append(float(r["Counter_Value"]))
            agg[name]["VGPR"].append(float(r.get("VGPR_Count", 0) or 0))
            agg[name]["LDS"].append(float(r.get("LDS_Block_Size", 0) or 0))
for name, d in sorted(agg.items()):
    print(name)
    for k, v in sorted(d.items()):
        print(f"   {k:28s} avg {sum(v)/len(v):.4g}  (n={len(v)})")
PY
