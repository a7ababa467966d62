#!/bin/bash
# A/B library with flags applied to EVERY device source: tools/build_variant_all.sh <tag> [-D...] -> gpu-raytracer_amd/libokrt_<tag>.so
# (use with OKRT_LIBRARY=...; tools/build_variant.sh rebuilds one file only)
set -e
cd /root/repo/gpu-raytracer_amd
tag=$1; shift
mkdir -p _build/all_$tag
pids=""
for f in csrc/*.hip csrc/*.cpp; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -Wno-unused-value "$@" -c $f -o _build/all_$tag/$(basename $f).o &
  pids="$pids $!"
done
for p in $pids; do wait $p; done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o libokrt_$tag.so _build/all_$tag/*.o -ldl
echo built libokrt_$tag.so
