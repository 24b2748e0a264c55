#!/bin/bash
# usage (GPU box): tools/exp_locus_rows.sh [pools] [loci] -- the shipped library and every tools/exp/libpoolgen_hip_*.so with the ols_iter / chisq_test kernel fixed
cd "$GRAFT_REPO_ROOT"
for lib in poolgen_amd/csrc/libpoolgen_hip.so tools/exp/libpoolgen_hip_*.so; do
  for k in rows stream; do
  for c in "0.0 0.001" "0.005 0.01"; do
    POOLGEN_OLS_ITER_KERNEL=$k POOLGEN_HIP_LIB=$lib python tools/bench_ops_realistic.py ${1:-100} ${2:-1000000} $c 2>/dev/null | python -c "
import json,sys
out=[]
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); out.append('%s %.3f' % (d['op'], d['kernel_ms']))
print('%-40s %-6s %-12s' % ('$lib', '$k', '$c'), ' | '.join(out))"
  done; done
done
