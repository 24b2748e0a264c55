#!/bin/bash
# usage: tools/exp_ops.sh "<extra CXXFLAGS>" [pools] [loci] -- rebuild with experiment macros and time the locus operators
cd "$GRAFT_REPO_ROOT/poolgen_amd/csrc" && rm -f pg_locus_ops.o && make CXXFLAGS="-O3 -std=c++17 -fPIC -ffp-contract=off -Wno-unused-function $1" >/dev/null 2>&1 || { echo build failed; exit 1; }
cd "$GRAFT_REPO_ROOT" && python tools/bench_ops.py ${2:-100} ${3:-1000000} 2>/dev/null | python -c "
import json,sys
for l in sys.stdin:
    d=json.loads(l)
    if 'gbs_algorithmic' in d: print('$1', d['op'], 'kernel_ms=%.3f'%d['kernel_ms'], 'GB/s=%.0f'%d['gbs_algorithmic'])"
