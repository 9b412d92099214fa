#!/bin/bash
# Build a variant of the library with extra compiler flags into lut_ldpc_amd/lib_variants/<name>/ (git-ignored, travels to
# the GPU box) for A/B runs:  tools/build_variant.sh nt_ld "-DLUTLDPC_LD_AUX=2"   then   LUTLDPC_LIB=.../<name>/liblut_ldpc_amd.so python bench.py
set -e
name=$1; flags=$2
root=$(cd "$(dirname "$0")/.." && pwd)
out=$root/lut_ldpc_amd/lib_variants/$name
mkdir -p "$out"
make -s -j8 -C "$root/lut_ldpc_amd/csrc" OUT="$out" HIPFLAGS="-O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wall -Wno-unused-function -ffp-contract=off $flags" "$out/liblut_ldpc_amd.so"
echo "built $out/liblut_ldpc_amd.so"
