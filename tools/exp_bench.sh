#!/bin/bash
# usage: tools/exp_bench.sh "<extra CXXFLAGS>"  -- rebuild the library on the GPU box with experiment macros and time the kernels
cd "$GRAFT_REPO_ROOT/poolgen_amd/csrc" && rm -f pg_kinship.o pg_sweep.o && make CXXFLAGS="-O3 -std=c++17 -fPIC -ffp-contract=off -Wno-unused-function $1" >/dev/null 2>&1 || { echo build failed; exit 1; }
cd "$GRAFT_REPO_ROOT" && timeout -k 10 300 python bench.py --steps 10 --no-cpu-baseline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$1', 'step_ms=%.3f'%d['ms_per_step'], {k:(round(v['avg_ms'],3) if isinstance(v,dict) else round(v,3)) for k,v in d['kernels'].items()})"
