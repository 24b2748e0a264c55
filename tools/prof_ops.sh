#!/bin/bash
# usage: tools/prof_ops.sh [pools] [loci] -- rocprofv3 kernel-trace of the locus operators, per-kernel average durations
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
rm -rf gpurun_out/prof_ops && mkdir -p gpurun_out/prof_ops
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_ops -- python3 tools/bench_ops.py ${1:-100} ${2:-1000000} > gpurun_out/prof_ops.log 2>&1
echo "exit $?"
f=$(find gpurun_out/prof_ops -name "*kernel_stats.csv" | head -1)
python3 - "$f" <<'PY'
import csv, sys
for r in csv.DictReader(open(sys.argv[1])):
    print("%-70s calls %5s avg_us %10.1f" % (r["Name"][:70], r["Calls"], float(r["AverageNs"]) / 1e3))
PY
