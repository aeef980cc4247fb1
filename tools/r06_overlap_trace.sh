#!/bin/bash
# NOTE: measures the three-stream overlapped schedule (ZK_SHARD_OVERLAP) as it existed at commit 9312777; it lost to the serial loop up to A = 40 us
# and was removed together with k_round_mid (profiles/r06_shard_overlap_ab.log, r06_shard_overlap_trace.log).  Check that commit out to re-run.
# kernel timeline of the one-rank sharded prover (RCCL), overlapped schedule with an injected 15-us all-reduce: start / end of every kernel of the
# last proof relative to its first launch, with the stream (queue) it ran on
set -u
export TMPDIR=/tmp
cd /tmp
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/prof_overlap
rm -rf $OUT; mkdir -p $OUT
export ZK_SHARD_OVERLAP=${ZK_SHARD_OVERLAP:-1} ZK_SHARD_FAKE_ALLREDUCE_US=${ZK_SHARD_FAKE_ALLREDUCE_US:-15}
timeout -k 10 200 rocprofv3 --kernel-trace --output-format csv -d $OUT/trace -- python3 $R/tools/prof_shard.py 13 3 21 > $OUT/run.log 2>&1 || { echo failed; tail -5 $OUT/run.log; exit 1; }
python3 - <<P
import csv, glob
f = glob.glob("$OUT/trace/**/*kernel_trace.csv", recursive=True)[0]
rows = sorted((r for r in csv.DictReader(open(f))), key=lambda r: int(r["Start_Timestamp"]))
idx = max(i for i, r in enumerate(rows) if "k_store_sponge" in r["Kernel_Name"])
t0 = int(rows[idx]["Start_Timestamp"])
for r in rows[idx:idx + 80]:
    n = r["Kernel_Name"].split("(")[0].replace("void zk::", "").replace("zk::", "")[:44]
    st, en = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    print(f"{n:46s} queue {r.get('Queue_Id', '?'):>3s}  start {(st - t0) / 1e3:8.1f}  end {(en - t0) / 1e3:8.1f}  dur {(en - st) / 1e3:7.1f} us")
P
rm -rf $OUT/trace
tail -1 $OUT/run.log
