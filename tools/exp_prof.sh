#!/bin/bash
# usage: tools/exp_prof.sh "<extra CXXFLAGS>" [pools] [loci] -- rebuild with experiment macros, rocprofv3 per-kernel times of the locus operators
cd "$GRAFT_REPO_ROOT/poolgen_amd/csrc" && rm -f pg_locus_ops.o && make CXXFLAGS="-O3 -std=c++17 -fPIC -ffp-contract=off -Wno-unused-function $1" >/dev/null 2>&1 || { echo build failed; exit 1; }
cd "$GRAFT_REPO_ROOT" && echo "== $1" && tools/prof_ops.sh ${2:-100} ${3:-1000000} | grep -E "k_locus_(first|close)<0"
