#!/bin/bash
# GPU box: kinship pass at several pool counts, shipped library against every tools/exp/libpoolgen_hip_*.so (bench_kinship_n.py)
cd "$GRAFT_REPO_ROOT"
for lib in poolgen_amd/csrc/libpoolgen_hip.so tools/exp/libpoolgen_hip_*.so; do
  echo "== $lib"; POOLGEN_HIP_LIB=$lib timeout -k 10 300 python tools/bench_kinship_n.py "$@" 2>&1 | grep -v amdgpu.ids
done
