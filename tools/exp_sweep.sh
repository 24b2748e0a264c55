#!/bin/bash
# usage: tools/exp_sweep.sh "<-D flags>" ... : rebuild pg_sweep.o with each flag set and print the two-pass bench's sweep time
cd "$GRAFT_REPO_ROOT/poolgen_amd/csrc"
for flags in "$@"; do
  rm -f pg_sweep.o && make CXXFLAGS="-O3 -std=c++17 -fPIC -ffp-contract=off -Wall -Wno-unused-function $flags" libpoolgen_hip.so > /dev/null 2>&1 || { echo "build failed: $flags"; continue; }
  cd ../.. && echo "== $flags" && POOLGEN_TWO_PASS=1 python bench.py --steps 10 --warmup 2 --no-cpu-baseline 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('step', round(d['ms_per_step'],3), 'sweep', d['kernels']['k_ols_sweep'])"; cd poolgen_amd/csrc
done
rm -f pg_sweep.o && make libpoolgen_hip.so > /dev/null 2>&1
