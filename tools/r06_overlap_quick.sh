#!/bin/bash
# quick look: n = 21, gather_below 13, A = 15 -- serial, overlapped, overlapped with more HW queues
for q in "" 8; do
  for mode in serial overlap; do
    env_extra="ZK_SHARD_FAKE_ALLREDUCE_US=15"
    [ "$mode" = overlap ] && env_extra="$env_extra ZK_SHARD_OVERLAP=1"
    [ -n "$q" ] && env_extra="$env_extra GPU_MAX_HW_QUEUES=$q"
    echo "== $mode queues=${q:-default}: $(env $env_extra python3 tools/prof_shard.py 13 9 21 2>/dev/null | tail -1)"
  done
done
