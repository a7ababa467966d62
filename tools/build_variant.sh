#!/bin/bash
# A/B library: tools/build_variant.sh <tag> <file.hip> [-D...]  -> gpu-raytracer_amd/libokrt_<tag>.so (use with OKRT_LIBRARY=...)
set -e
cd /root/repo/gpu-raytracer_amd
tag=$1; f=$2; shift 2
python3 build.py >/dev/null 2>&1 || true
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -Wno-unused-value "$@" -c csrc/$f -o _build/variant_$tag.o
objs=$(ls _build/*.o | grep -v variant_ | grep -v "/$f.o")
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o libokrt_$tag.so $objs _build/variant_$tag.o
echo built libokrt_$tag.so
