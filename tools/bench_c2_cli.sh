#!/bin/bash
# usage: tools/bench_c2_cli.sh [threads=16] -- BASELINE config 2 from text: 100 pools x 1 M loci of sync text through
# `poolgen ols_iter` (and chisq_test), wall-clocked
thr=${1:-16}
d=${TMPDIR:-/tmp}/pg_c2; rm -rf $d; mkdir -p $d
python3 tools/gen_sync.py $d/base.sync $d/phen.csv 100 200000
: > $d/big.sync
for i in 1 2 3 4 5; do pre=$(printf "\\$(printf '%03o' $((96 + i)))"); sed "s/^chr/${pre}chr/" $d/base.sync >> $d/big.sync; done
python3 -c "import os,sys; print('file bytes', os.path.getsize(sys.argv[1]))" $d/big.sync
for an in ols_iter chisq_test; do
  for rep in 1 2; do
    rm -f $d/out.csv
    s=$(date +%s.%N)
    PGH_TIMING=1 poolgen_amd/csrc/poolgen $an -f $d/big.sync -p $d/phen.csv --phen-value-col 2 --n-threads $thr -o $d/out.csv 2> $d/err.txt
    e=$(date +%s.%N)
    python3 -c "import sys; print(sys.argv[1], 'run', sys.argv[2], 'wall %.3f s' % (float(sys.argv[4]) - float(sys.argv[3])))" $an $rep $s $e
    grep -E "^poolgen:" $d/err.txt
  done
  wc -l $d/out.csv
done
rm -rf $d
