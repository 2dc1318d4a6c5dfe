#!/bin/bash
# Build tuning variants of libtsdf_hip.so into build/variants/ (they travel to the GPU box with the snapshot):
#   tools/build_variants.sh name1="-DFLAG=1 -DOTHER=2" name2="..."
# Select one at run time with TSDF_HIP_LIB=build/variants/libtsdf_hip_<name>.so (tracking_sdf_amd/__init__.py).
set -e
ROOT="$(cd "$(dirname "$0")/.." && pwd)"
mkdir -p "$ROOT/build/variants"
for spec in "$@"; do
    name="${spec%%=*}"; flags="${spec#*=}"
    out="$ROOT/build/variants/libtsdf_hip_${name}.so"
    echo "building $name: $flags"
    make -s -C "$ROOT" LIB="$out" LIBDIR="$ROOT/build/variants" HIPEXTRA="$flags" lib &
done
wait
ls -la "$ROOT/build/variants"
