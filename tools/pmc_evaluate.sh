#!/bin/bash
# evaluate (k_eval_stream + k_eval_low) under rocprofv3: kernel stats, then VALU / wait counters and FETCH_SIZE in passes of their own.
set -u
export TMPDIR=/tmp
OUT=$PWD/gpurun_out/pmc_eval
rm -rf $OUT; mkdir -p $OUT
timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 tools/prof_evaluate.py > $OUT/trace.log 2>&1 || { echo "trace failed"; tail -3 $OUT/trace.log; exit 1; }
cp $(find $OUT/trace -name "*kernel_stats.csv" | head -1) $OUT/evaluate_kernel_stats.csv
cat $OUT/evaluate_kernel_stats.csv
grep "^evaluate" $OUT/trace.log
for set in "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU" "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_LDS" "FETCH_SIZE" "WRITE_SIZE"; do
  tag=$(echo $set | tr ' ' '_' | cut -c1-40)
  timeout -k 10 120 rocprofv3 --kernel-trace --pmc $set --output-format csv -d $OUT/$tag -- python3 tools/prof_evaluate.py > $OUT/$tag.log 2>&1 || { echo "pass $set failed"; tail -3 $OUT/$tag.log; }
done
python3 - <<'PY'
import csv, glob, collections, os
out = os.path.join(os.environ.get("GRAFT_REPO_ROOT", "."), "gpurun_out", "pmc_eval")
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(out + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "k_eval" in r["Kernel_Name"]:
            name = r["Kernel_Name"].split("(")[0].replace("void zk::", "") + " grid=" + r.get("Grid_Size", "?")
            agg[name][r["Counter_Name"]].append(float(r["Counter_Value"]))
for name, d in sorted(agg.items()):
    print(name)
    for k, v in sorted(d.items()):
        print(f"   {k:28s} avg {sum(v)/len(v):.6g}  (n={len(v)})")
PY
