#!/bin/bash
# A/B library with extra defines applied to EVERY device source: tools/build_variant_all.sh <tag> [-D...] -> gpu-raytracer_amd/libokrt_<tag>.so
set -e
cd /root/repo/gpu-raytracer_amd
tag=$1; shift
mkdir -p _build/var_$tag
objs=""
for f in csrc/*.hip csrc/*.cpp; do
  o=_build/var_$tag/$(basename $f).o
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -Wno-unused-value "$@" -c $f -o $o &
  objs="$objs $o"
done
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o libokrt_$tag.so $objs -ldl
echo built libokrt_$tag.so
