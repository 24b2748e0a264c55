#!/bin/bash
# usage: tools/bench_stream.sh [copies=10] [threads=16] [sync|pileup] -- config-5 rehearsal: a multi-GB sorted sync (or
# mpileup) text through the piece-wise ols_iter_with_kinship path of the CLI (parse | H2D + loader + partial kinship, then
# sweep), wall-clocked.
set -e
copies=${1:-10}; thr=${2:-16}; kind=${3:-sync}
d=${TMPDIR:-/tmp}/pg_stream; rm -rf $d; mkdir -p $d
t0=$(date +%s.%N)
if [ "$kind" = pileup ]; then python3 tools/gen_pileup.py $d/base.sync $d/phen.csv 200 30000; else python3 tools/gen_sync.py $d/base.sync $d/phen.csv 200 200000; fi
t1=$(date +%s.%N)
big=$d/big.$kind
: > $big
for i in $(seq 1 $copies); do
  pre=$(printf '%02d' $i)                              # 01, 02, ...: chromosome names stay sorted (up to 99 copies)
  sed "s/^chr/${pre}chr/" $d/base.sync >> $big
done
t2=$(date +%s.%N)
python3 -c "import os,sys; print('file bytes', os.path.getsize(sys.argv[1])); print('generate base %.1f s, replicate %.1f s' % (float(sys.argv[3]) - float(sys.argv[2]), float(sys.argv[4]) - float(sys.argv[3])))" $big $t0 $t1 $t2
for rep in 1 2; do
  rm -f $d/out.csv
  s=$(date +%s.%N)
  PGH_TIMING=1 poolgen_amd/csrc/poolgen ols_iter_with_kinship -f $big -p $d/phen.csv --phen-value-col 2 --n-threads $thr -o $d/out.csv 2> $d/timing.$rep.txt
  e=$(date +%s.%N)
  python3 -c "import sys; print('run', sys.argv[1], 'wall %.3f s' % (float(sys.argv[3]) - float(sys.argv[2])))" $rep $s $e
  grep -E '^poolgen:' $d/timing.$rep.txt
done
wc -l $d/out.csv
rm -rf $d
