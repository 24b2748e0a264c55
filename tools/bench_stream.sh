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
  grep -E '^poolgen:' $d/timing.$rep.txt | grep -v 'clock at'
  # the parser's own laps, summed over the pieces; then what lies outside the program's laps: exec + dynamic loading before
  # main(), the teardown of the process after its last line
  python3 - $s $e $d/timing.$rep.txt <<'PY'
import collections, re, sys
s, e = float(sys.argv[1]), float(sys.argv[2])
txt = open(sys.argv[3]).read()
laps = collections.defaultdict(list)
for what, sec in re.findall(r'^parse_sync_file: (.*?)\s+([0-9.]+) s$', txt, re.M):
    laps[what].append(float(sec))
for what, v in laps.items():
    print('  parser, %d pieces: %-12s sum %.3f s, median %.4f, max %.4f' % (len(v), what, sum(v), sorted(v)[len(v) // 2], max(v)))
st = dict(re.findall(r'clock at (\w+)\s+([0-9.]+)', txt))
if 'main' in st and 'exit' in st:
    print('  before main() %.3f s, main() to exit %.3f s, after exit (process teardown) %.3f s' % (float(st['main']) - s, float(st['exit']) - float(st['main']), e - float(st['exit'])))
PY
done
wc -l $d/out.csv
rm -rf $d
