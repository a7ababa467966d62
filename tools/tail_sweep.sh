#!/bin/bash
# Sweep of the tail-drain export criteria on single-frame and 8-frame launch sets (developer tool).
cd "${GRAFT_REPO_ROOT:-/root/repo}"
for wl in sponza dragon; do
for cfg in "0 0" "8 0" "16 0" "24 0" "32 0" "0 8" "0 16" "0 32" "16 16"; do
  set -- $cfg
  echo "== $wl lanes<=$1 iters>=$2"
  OKRT_TAIL_LANES=$1 OKRT_TAIL_ITERS=$2 timeout -k 10 120 python tools/frame_scaling.py $wl 1 8 2>&1 | grep "F="
done; done
