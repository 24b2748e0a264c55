#!/bin/bash
# GPU box: the count operators at 100 and 200 pools x 1 M loci (BASELINE configs[1] and its 200-pool sibling):
#   gpurun_out/r03_secondary_ops.jsonl        one JSON line per operator and pool count (HIP-event kernel times, tools/bench_ops.py)
#   gpurun_out/r03_secondary_ops_n<N>_kernel_stats.csv   rocprofv3 --kernel-trace --stats of the same command
set -e
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
: > gpurun_out/r03_secondary_ops.jsonl
for n in 100 200; do
  timeout -k 10 300 python3 tools/bench_ops.py $n 1000000 >> gpurun_out/r03_secondary_ops.jsonl
  rm -rf gpurun_out/prof_ops_$n
  timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_ops_$n -- python3 tools/bench_ops.py $n 1000000 > gpurun_out/prof_ops_$n.log 2>&1
  f=$(find gpurun_out/prof_ops_$n -name "*kernel_stats.csv" | head -1)
  cp "$f" gpurun_out/r03_secondary_ops_n${n}_kernel_stats.csv
  rm -rf gpurun_out/prof_ops_$n
done
cat gpurun_out/r03_secondary_ops.jsonl | python3 -c "
import json,sys
for l in sys.stdin:
    d=json.loads(l)
    if 'frac_of_hbm_peak' in d: print(d['op'], d['pools'], '%.4f ms' % d['kernel_ms'], '%.3f' % d['frac_of_hbm_peak'])"
for n in 100 200; do grep -E "k_locus" gpurun_out/r03_secondary_ops_n${n}_kernel_stats.csv | cut -c1-160; done
