#!/bin/bash
# usage (GPU box): tools/exp_locus_run.sh [pools] [loci] -- every tools/exp/libpoolgen_hip_*.so and the shipped library, one process each
cd "$GRAFT_REPO_ROOT"
for lib in poolgen_amd/csrc/libpoolgen_hip.so tools/exp/libpoolgen_hip_*.so; do
  POOLGEN_HIP_LIB=$lib python tools/bench_ops.py ${1:-100} ${2:-1000000} 2>/dev/null | python -c "
import json,sys
out=[]
for l in sys.stdin:
    d=json.loads(l)
    if 'gbs_algorithmic' in d: out.append('%s %.3f' % (d['op'], d['kernel_ms']))
print('%-48s' % '$lib', ' | '.join(out))"
done
