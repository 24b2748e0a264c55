#!/bin/bash
# usage (GPU box): tools/exp_kin_run.sh -- the headline step with the shipped library and every tools/exp/libpoolgen_hip_*.so
cd "$GRAFT_REPO_ROOT"
for rep in 1 2; do
for lib in poolgen_amd/csrc/libpoolgen_hip.so tools/exp/libpoolgen_hip_*.so; do
  POOLGEN_HIP_LIB=$lib python bench.py --no-cpu-baseline --no-secondary --no-sweep-legs --steps 20 --warmup 3 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('%-44s step %.3f ms  kinship %.3f ms  frac %.3f' % ('$lib', d['ms_per_step'], d['roofline']['avg_ms'], d['roofline']['frac']))"
done; done
