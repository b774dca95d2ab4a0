#!/bin/bash
# build_variant.sh <tag> [extra hipcc flags ...]: a developer build of the library beside the product one -> taichislam_amd/lib/libtaichislam_hip_<tag>.so
# (objects under lib/obj_<tag>/; use with TSL_LIB=... or the tools/gpu_*_ab.sh scripts).  Only the sources given in SRCS (default: all) are compiled
# with the extra flags' objects; everything is linked together.
set -e
tag=$1; shift
R=$(cd "$(dirname "$0")/.." && pwd); C=$R/taichislam_amd/csrc; L=$R/taichislam_amd/lib; O=$L/obj_$tag; mkdir -p $O
F="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fhip-fp32-correctly-rounded-divide-sqrt -fno-fast-math -Wall -Wno-unused-function $*"
pids=()
for s in $C/*.hip; do b=$(basename $s .hip); ( hipcc $F -c $s -o $O/$b.o ) & pids+=($!); done
for p in "${pids[@]}"; do wait $p; done
hipcc --offload-arch=gfx950 -shared -fPIC $O/*.o -o $L/libtaichislam_hip_$tag.so
echo built $L/libtaichislam_hip_$tag.so
