#!/bin/bash
# usage (here, no GPU needed): tools/exp_locus.sh <tag> "<-D flags>"   -> tools/exp/libpoolgen_hip_<tag>.so
# Timing experiments on the count operators: pg_locus_ops.hip rebuilt with experiment macros, linked with the shipped objects.
# Run on the GPU box with POOLGEN_HIP_LIB=tools/exp/libpoolgen_hip_<tag>.so python tools/bench_ops.py 100 1000000
set -e
cd "$(dirname "$0")/../poolgen_amd/csrc"
mkdir -p ../../tools/exp
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -Wno-unused-function $2 -c pg_locus_ops.hip -o ../../tools/exp/pg_locus_ops_$1.o
objs=$(ls *.o | grep -v pg_locus_ops.o)
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -o ../../tools/exp/libpoolgen_hip_$1.so $objs ../../tools/exp/pg_locus_ops_$1.o -ldl
rm -f ../../tools/exp/pg_locus_ops_$1.o
echo built tools/exp/libpoolgen_hip_$1.so
