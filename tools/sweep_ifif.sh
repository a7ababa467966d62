#!/bin/bash
# developer sweep of the if-if kernel's thresholds (compile-time since round 2: one A/B library per setting), run from the
# build container: tools/sweep_ifif.sh  -> gpu-raytracer_amd/libokrt_ifif_<refill>_<nodeMin>_<triMin>.so; then on the GPU box
#   OKRT_LIBRARY=gpu-raytracer_amd/libokrt_ifif_40_32_24.so OKRT_FAST_BATCH=64 bash tools/bench_all.sh sponza dragon80k dragon glass
for cfg in "40 32 24" "40 40 24" "40 32 32" "48 32 24" "32 32 24" "40 24 16" "40 48 32"; do
  set -- $cfg
  bash /root/repo/tools/build_variant_all.sh ifif_$1_$2_$3 -DOKRT_REFILL_BELOW=$1 -DOKRT_NODE_MIN=$2 -DOKRT_TRI_MIN=$3 || exit 1
done
