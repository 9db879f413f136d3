#!/bin/bash
# Timing-only builds of the library with ablation / tuning macros (never shipped, never loaded by default):
#   tools/build_variant.sh <name> "<extra hipcc flags>"   ->  build/variants/<name>/libcu2rec_amd.so
# Load one with CU2REC_AMD_LIB=build/variants/<name>/libcu2rec_amd.so (cu2rec_amd/_lib.py).
set -euo pipefail
name=$1; flags=${2:-}
root=$(cd "$(dirname "$0")/.." && pwd)
out=$root/build/variants/$name
mkdir -p "$out/obj"
make -C "$root/cu2rec_amd/csrc" -j4 OBJDIR="$out/obj" LIB="$out/libcu2rec_amd.so" \
     HIPFLAGS="--offload-arch=gfx950 -ffp-contract=off $flags" "$out/libcu2rec_amd.so"
echo "$out/libcu2rec_amd.so"
