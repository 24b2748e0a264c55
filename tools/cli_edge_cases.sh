#!/bin/bash
# usage: tools/cli_edge_cases.sh -- tiny and odd inputs through every subcommand: each must end with a result or a
# one-line error (exit 1), never a crash / signal / GPU fault
B=poolgen_amd/csrc/poolgen
d=${TMPDIR:-/tmp}/pg_edge; rm -rf $d; mkdir -p $d
printf 'chr1\t10\tN\t5:5:0:0:0:0\t3:7:0:0:0:0\n' > $d/one.sync
printf '#p,s,y\nA,10,1.0\nB,10,2.0\n' > $d/two.csv
printf 'chr1\t10\tN\t5:5:0:0:0:0\t3:7:0:0:0:0\t6:4:0:0:0:0\nchr1\t20\tN\t1:9:0:0:0:0\t2:8:0:0:0:0\t9:1:0:0:0:0\nchr1\t20\tN\t1:9:0:0:0:0\t2:8:0:0:0:0\t9:1:0:0:0:0\nchr2\t5\tN\t0:0:0:0:0:0\t2:8:0:0:0:0\t9:1:0:0:0:0\n' > $d/dup.sync
printf '#p,s,y\nA,10,1.0\nB,10,NA\nC,10,0.5\n' > $d/three.csv
printf '#p,s,y\nA,10,1.0\nB,10,2.5\nC,10,0.5\n' > $d/three_ok.csv
: > $d/empty.sync
run() { local name=$1; shift; "$@" > $d/out.txt 2> $d/err.txt; rc=$?; printf '%-44s rc=%-3s %s\n' "$name" $rc "$(head -c 150 $d/err.txt | tr '\n' ' ')"; if [ $rc -gt 1 ]; then echo "  ^^^ CRASH"; fi; }
for a in chisq_test pearson_corr ols_iter ols_iter_with_kinship fst heterozygosity genomic_prediction_cross_validation; do
  run "$a one locus two pools" $B $a -f $d/one.sync -p $d/two.csv --phen-value-col 2 -o $d/o1_$a.csv --min-loci-per-window 1
  run "$a dup/uncovered, NA phenotype" $B $a -f $d/dup.sync -p $d/three.csv --phen-value-col 2 -o $d/o2_$a.csv --min-loci-per-window 1
  run "$a dup/uncovered" $B $a -f $d/dup.sync -p $d/three_ok.csv --phen-value-col 2 -o $d/o3_$a.csv --min-loci-per-window 1 --min-coverage-depth 0 --max-missingness-rate 0.5
  run "$a empty file" $B $a -f $d/empty.sync -p $d/two.csv --phen-value-col 2 -o $d/o4_$a.csv
  run "$a pools mismatch" $B $a -f $d/dup.sync -p $d/two.csv --phen-value-col 2 -o $d/o5_$a.csv
done
run "streamed tiny pieces" env PGH_STREAM_CHUNK_BYTES=64 $B ols_iter_with_kinship -f $d/dup.sync -p $d/three_ok.csv --phen-value-col 2 -o $d/o6.csv
run "missing file" $B ols_iter -f $d/nope.sync -p $d/two.csv --phen-value-col 2
run "bad flag" $B ols_iter -f $d/one.sync -p $d/two.csv --what
ls $d/*.csv | wc -l
rm -rf $d
