#!/bin/bash
# usage (GPU box): tools/exp_parse.sh [copies=10] -- what the sync parser does on a cold multi-GB text: plain or pinned output, with and without pre-faulting
cd "$GRAFT_REPO_ROOT"; d=/tmp/pgparse; rm -rf $d; mkdir -p $d
hipcc -O2 -std=c++17 -Ipoolgen_amd/csrc/host tools/parse_bench.cpp poolgen_amd/csrc/host/host_util.cpp -o $d/parse_bench -lpthread || exit 1
python3 tools/gen_sync.py $d/base.sync $d/phen.csv 200 200000
: > $d/big.sync
for i in $(seq 1 ${1:-10}); do cat $d/base.sync >> $d/big.sync; done
ls -la $d/big.sync
for mode in "malloc" "pinned" "pinned pread" "malloc pread" "pinned" "pinned pread"; do
  for t in 16; do PGH_TIMING= $d/parse_bench $d/big.sync $t 256 $mode 2>/dev/null; done
done
$d/parse_bench $d/big.sync 8 256 pinned 2>/dev/null
$d/parse_bench $d/big.sync 16 64 pinned 2>/dev/null
$d/parse_bench $d/big.sync 16 1024 pinned 2>/dev/null
rm -rf $d
