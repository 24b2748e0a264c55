#!/bin/bash
# usage: tools/exp_gp.sh "<-D flags>" ... : rebuild pg_gp.o with each flag set, profile bench_ridge, print predict_folds time
cd "$GRAFT_REPO_ROOT/poolgen_amd/csrc"
for flags in "$@"; do
  rm -f pg_gp.o && make CXXFLAGS="-O3 -std=c++17 -fPIC -ffp-contract=off -Wall -Wno-unused-function $flags" libpoolgen_hip.so > /dev/null 2>&1 || { echo "build failed: $flags"; exit 1; }
  cd ../.. && echo "== $flags" && bash tools/prof_any.sh exp tools/bench_ridge.py 500 5000000 2 10 | grep -E "k_gp_predict_folds|k_gp_beta<12>" ; cd poolgen_amd/csrc
done
rm -f pg_gp.o && make libpoolgen_hip.so > /dev/null 2>&1
