#!/bin/bash
# usage (GPU box): tools/exp_locus_direct.sh [pools] [loci] -- the order-free kernel with and without its LDS staging buffer (POOLGEN_ROWS_DIRECT), every tools/exp library
cd "$GRAFT_REPO_ROOT"
for lib in poolgen_amd/csrc/libpoolgen_hip.so tools/exp/libpoolgen_hip_d*.so; do
  for d in 0 1; do
  for c in "0.0 0.001" "0.005 0.01"; do
    if [ $d = 1 ]; then export POOLGEN_ROWS_DIRECT=1; else unset POOLGEN_ROWS_DIRECT; fi
    POOLGEN_OLS_ITER_KERNEL=rows POOLGEN_HIP_LIB=$lib python tools/bench_ops_realistic.py ${1:-100} ${2:-1000000} $c 2>/dev/null | python -c "
import json,sys
out=[]
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); out.append('%s %.3f' % (d['op'], d['kernel_ms']))
print('%-44s direct=%s %-12s' % ('$lib', '$d', '$c'), ' | '.join(out))"
  done; done
done
