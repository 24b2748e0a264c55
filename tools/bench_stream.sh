#!/bin/bash
# usage: tools/bench_stream.sh [copies=10] [threads=16] -- config-5 rehearsal: a multi-GB sorted sync text through the
# piece-wise ols_iter_with_kinship path of the CLI (parse | H2D + loader + partial kinship, then sweep), wall-clocked.
set -e
copies=${1:-10}; thr=${2:-16}
d=${TMPDIR:-/tmp}/pg_stream; rm -rf $d; mkdir -p $d
t0=$(date +%s.%N)
python3 tools/gen_sync.py $d/base.sync $d/phen.csv 200 200000
t1=$(date +%s.%N)
: > $d/big.sync
for i in $(seq 1 $copies); do
  pre=$(printf "\\$(printf '%03o' $((96 + i)))")      # a, b, c, ...: chromosome names stay sorted
  sed "s/^chr/${pre}chr/" $d/base.sync >> $d/big.sync
done
t2=$(date +%s.%N)
python3 -c "import os,sys; print('file bytes', os.path.getsize(sys.argv[1])); print('generate base %.1f s, replicate %.1f s' % (float(sys.argv[3]) - float(sys.argv[2]), float(sys.argv[4]) - float(sys.argv[3])))" $d/big.sync $t0 $t1 $t2
for rep in 1 2; do
  rm -f $d/out.csv
  s=$(date +%s.%N)
  PGH_TIMING=1 poolgen_amd/csrc/poolgen ols_iter_with_kinship -f $d/big.sync -p $d/phen.csv --phen-value-col 2 --n-threads $thr -o $d/out.csv 2> $d/timing.$rep.txt
  e=$(date +%s.%N)
  python3 -c "import sys; print('run', sys.argv[1], 'wall %.3f s' % (float(sys.argv[3]) - float(sys.argv[2])))" $rep $s $e
  cat $d/timing.$rep.txt | tail -12
done
wc -l $d/out.csv
rm -rf $d
