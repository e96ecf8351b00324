#!/bin/bash
# build a variant of libinfgen_hip.so from a scratch copy of csrc/: tools/build_variant.sh <name> [-DFLAG ...]  ->  build_exp/libinfgen_hip_<name>.so
# (swapped in on the GPU box with EXP_LIB=build_exp/libinfgen_hip_<name>.so by tools/bench_fourier.py / bench_attn.py, same-box A/B)
set -e
name=$1; shift
R=/root/repo; T=/tmp/variant_$name
rm -rf $T; mkdir -p $T/infgen_amd $R/build_exp
cp -r $R/infgen_amd/csrc $T/infgen_amd/csrc; cp -r $R/include $T/include
cd $T/infgen_amd/csrc; rm -f *.o
make -j8 CXXFLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fno-slp-vectorize -Wall -Wno-unused-variable $*" TARGET=$R/build_exp/libinfgen_hip_$name.so 2>&1 | grep -E "error|warning: v|Error" || true
ls -la $R/build_exp/libinfgen_hip_$name.so
