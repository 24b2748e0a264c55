#!/bin/bash
# usage: tools/parse_exp.sh -- parser and CLI phase times on a 1.09 GB sync text (400 k loci x 200 pools), several thread counts
d=/tmp/pg_px; rm -rf $d; mkdir -p $d
python3 tools/gen_sync.py $d/base.sync $d/phen.csv 200 100000
for i in 1 2 3 4; do cat $d/base.sync >> $d/big.sync; done
ls -l $d/big.sync | cut -d' ' -f5
for t in 8 16 32 64; do echo "threads $t"; poolgen_amd/csrc/hostcheck parsetime $d/big.sync $t | tail -1; done
for t in 16 32; do PGH_TIMING=1 poolgen_amd/csrc/poolgen ols_iter -f $d/big.sync -p $d/phen.csv --phen-value-col 2 --n-threads $t -o $d/o$t.csv 2>&1 | grep -E "parse sync|parse  |allocate|GPU operator|format"; done
rm -rf $d
