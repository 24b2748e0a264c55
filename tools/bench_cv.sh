#!/bin/bash
# usage: tools/bench_cv.sh [pools=200] [loci=100000] -- genomic_prediction_cross_validation end to end (default 10 folds x 3
# replicates x 6 models) on a synthetic sync file, wall-clocked
n=${1:-200}; L=${2:-100000}
d=${TMPDIR:-/tmp}/pg_cv; rm -rf $d; mkdir -p $d
python3 tools/gen_sync.py $d/x.sync $d/phen.csv $n $L
s=$(date +%s.%N)
PGH_TIMING=1 poolgen_amd/csrc/poolgen genomic_prediction_cross_validation -f $d/x.sync -p $d/phen.csv --phen-value-col 2 --n-threads 16 -o $d/cv.csv 2> $d/err.txt
e=$(date +%s.%N)
python3 -c "import sys; print('wall %.2f s' % (float(sys.argv[2]) - float(sys.argv[1])))" $s $e
grep -E "^poolgen:" $d/err.txt
head -4 $d/cv.csv; wc -l $d/cv.csv; ls $d | grep genomic_predictors | head -8
rm -rf $d
