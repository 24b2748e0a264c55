#!/bin/bash
# usage (here): tools/exp_kin.sh <tag> "<-D flags>" -> tools/exp/libpoolgen_hip_<tag>.so with pg_kinship.hip rebuilt under the flags
set -e
cd "$(dirname "$0")/../poolgen_amd/csrc"
mkdir -p ../../tools/exp
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -Wno-unused-function $2 -c pg_kinship.hip -o ../../tools/exp/pg_kinship_$1.o
objs=$(ls *.o | grep -v pg_kinship.o)
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -o ../../tools/exp/libpoolgen_hip_$1.so $objs ../../tools/exp/pg_kinship_$1.o -ldl
rm -f ../../tools/exp/pg_kinship_$1.o
echo built tools/exp/libpoolgen_hip_$1.so
