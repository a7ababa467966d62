#!/bin/bash
# developer sweep of the if-if kernel's knobs: tools/sweep_ifif.sh  (refill nodeMin triMin batch)
cd "${GRAFT_REPO_ROOT:-/root/repo}"
for cfg in "40 32 24 64" "40 40 24 64" "40 32 32 64" "48 32 24 64" "32 32 24 64" "40 32 24 128" "40 32 24 32" "40 24 16 64" "40 48 32 64"; do
  set -- $cfg
  echo "== refill $1 nodeMin $2 triMin $3 batch $4"
  OKRT_FAST_REFILL=$1 OKRT_FAST_NODE_MIN=$2 OKRT_FAST_TRI_MIN=$3 OKRT_FAST_BATCH=$4 bash tools/bench_all.sh sponza dragon80k dragon glass | awk '{printf "%s %s  ", $1, $2} END {print ""}' || exit 1
done
