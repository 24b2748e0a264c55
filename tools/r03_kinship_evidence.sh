#!/bin/bash
# GPU box: the kinship experiments of round 3 that were NOT adopted, with their numbers -> gpurun_out/r03_kinship_*.log
# (build first, here: tools/exp_kin.sh kc32 "-DKIN_KC_DEF=32"; tools/exp_kin.sh kc64 "-DKIN_KC_DEF=64"; tools/mb_corun as its header says)
set -e
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
{
  echo "# stage length (loci per LDS stage) at 100 / 112 / 200 pools: shipped 16 against 32 and 64 (tools/bench_kinship_n.py)"
  for lib in poolgen_amd/csrc/libpoolgen_hip.so tools/exp/libpoolgen_hip_kc32.so tools/exp/libpoolgen_hip_kc64.so; do
    echo "== $lib"; POOLGEN_HIP_LIB=$lib timeout -k 10 200 python tools/bench_kinship_n.py 100 112
  done
} > gpurun_out/r03_kinship_stage_length.log 2>&1
{
  echo "# does a light streaming kernel run beside the (non-fused) kinship pass?  tools/mb_corun.hip"
  timeout -k 10 200 tools/mb_corun
} > gpurun_out/r03_kinship_corun.log 2>&1
tail -5 gpurun_out/r03_kinship_stage_length.log; tail -8 gpurun_out/r03_kinship_corun.log
