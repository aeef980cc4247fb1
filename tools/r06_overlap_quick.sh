#!/bin/bash
# NOTE: measures the three-stream overlapped schedule (ZK_SHARD_OVERLAP) as it existed at commit 9312777; it lost to the serial loop up to A = 40 us
# and was removed together with k_round_mid (profiles/r06_shard_overlap_ab.log, r06_shard_overlap_trace.log).  Check that commit out to re-run.
# quick look: n = 21, gather_below 13, A = 15 -- serial, overlapped, overlapped with more HW queues
for q in "" 8; do
  for mode in serial overlap; do
    env_extra="ZK_SHARD_FAKE_ALLREDUCE_US=15"
    [ "$mode" = overlap ] && env_extra="$env_extra ZK_SHARD_OVERLAP=1"
    [ -n "$q" ] && env_extra="$env_extra GPU_MAX_HW_QUEUES=$q"
    echo "== $mode queues=${q:-default}: $(env $env_extra python3 tools/prof_shard.py 13 9 21 2>/dev/null | tail -1)"
  done
done
