#!/bin/bash
# tools/pmc_zeta.sh -- to_evaluation_form at 2^24 on the GPU box: same-box A/B of the LDS-tiled passes against round 4's global
# passes (wall clock), per-kernel stats, then HBM traffic from FETCH_SIZE / WRITE_SIZE in PMC passes of their own.
# Output: gpurun_out/r05_zeta.log
set -u
cd "${GRAFT_REPO_ROOT:-.}"
OUT=$PWD/gpurun_out
mkdir -p "$OUT"
LOG=$OUT/r05_zeta.log
export TMPDIR=/tmp
{
  echo "## wall clock, same box, alternating"
  for r in 1 2 3; do
    python3 tools/prof_zeta.py 24 9
    ZK_ZETA_GLOBAL=1 python3 tools/prof_zeta.py 24 9
  done
  for n in 12 16 20 22 26; do python3 tools/prof_zeta.py $n 9; ZK_ZETA_GLOBAL=1 python3 tools/prof_zeta.py $n 9; done
  echo "## dense term list (2^16 terms)"
  python3 tools/prof_zeta.py 24 5 65536
  echo "## kernel stats (rocprofv3 --kernel-trace --stats)"
  rm -rf /tmp/zeta_stats && rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/zeta_stats -- python3 tools/prof_zeta.py 24 9 > /dev/null 2>&1
  python3 tools/summarize_prof.py /tmp/zeta_stats 2>/dev/null | head -12 || true
  f=$(find /tmp/zeta_stats -name '*kernel_stats.csv' | head -1); [ -n "$f" ] && head -8 "$f"
  for ctr in FETCH_SIZE WRITE_SIZE; do
    echo "## --pmc $ctr (KiB per launch; FETCH_SIZE doubled on gfx950 for 16-B/lane streaming reads: MI355X_MICROARCH.md HBM)"
    rm -rf /tmp/zeta_pmc_$ctr && rocprofv3 --kernel-trace --pmc $ctr --output-format csv -d /tmp/zeta_pmc_$ctr -- python3 tools/prof_zeta.py 24 3 > /dev/null 2>&1
    f=$(find /tmp/zeta_pmc_$ctr -name '*counter_collection.csv' | head -1)
    python3 - "$f" $ctr <<'PY'
import csv, sys, collections
rows = collections.defaultdict(list)
for r in csv.DictReader(open(sys.argv[1])):
    if r.get("Counter_Name", sys.argv[2]) == sys.argv[2]:
        rows[r["Kernel_Name"].split("(")[0]].append(float(r["Counter_Value"]))
for k, v in rows.items():
    if "zeta" in k or "fill" in k.lower() or "scatter" in k:
        print(f"{k}: {len(v)} launches, avg {sum(v)/len(v):.0f} KiB")
PY
  done
} > "$LOG" 2>&1
tail -40 "$LOG"
