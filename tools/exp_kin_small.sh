#!/bin/bash
# GPU box: kinship at small pool counts -- 16-wave workgroups (shipped) against 8-wave ones, one or two per CU
cd "$GRAFT_REPO_ROOT"
for cfg in "poolgen_amd/csrc/libpoolgen_hip.so 1" "poolgen_amd/csrc/libpoolgen_hip.so 2" "tools/exp/libpoolgen_hip_w8.so 1" "tools/exp/libpoolgen_hip_w8.so 2" "tools/exp/libpoolgen_hip_w8.so 3"; do
  set -- $cfg
  echo "== $1 workgroups per CU: $2"
  POOLGEN_HIP_LIB=$1 POOLGEN_KIN_SLAB_MULT=$2 timeout -k 10 200 python tools/bench_kinship_n.py 48 64 100 112 150 2>&1 | grep -v amdgpu.ids
done
