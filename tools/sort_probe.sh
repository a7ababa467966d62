#!/bin/bash
# what ray coherence is worth to traceKernel: the bounce-ray queue fully sorted (OKRT_FAST_SORT) against the append order.
# The diagnostic is no longer in the product: `git apply tools/experiments/r04_ray_sort_probe.diff` first (round-4 tree shape).
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out/sort
for mode in 0 1 2 3 4; do
  echo "== OKRT_FAST_SORT=$mode"
  OKRT_FAST_SORT=$mode python3 tools/ab_trace.py ${@:-sponza dragon glass} || exit 1
done 2>&1 | tee gpurun_out/sort/ab_trace.txt
for mode in 0 2 3; do
  echo "== stamps OKRT_FAST_SORT=$mode"
  OKRT_FAST_SORT=$mode python3 tools/stamps2.py sponza dragon || exit 1
done 2>&1 | tee gpurun_out/sort/stamps.txt
