#!/bin/bash
# usage: tools/prof_any.sh <tag> <python script> [args...] -- rocprofv3 kernel-trace summary (our kernels only)
tag=$1; shift
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
rm -rf gpurun_out/prof_$tag && mkdir -p gpurun_out/prof_$tag
timeout -k 10 600 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_$tag -- python3 "$@" > gpurun_out/prof_$tag.log 2>&1
echo "exit $?"; tail -2 gpurun_out/prof_$tag.log | cut -c1-400
f=$(find gpurun_out/prof_$tag -name "*kernel_stats.csv" | head -1)
python3 - "$f" <<'PY'
import csv, sys
tot = 0.0
for r in csv.DictReader(open(sys.argv[1])):
    if "k_" in r["Name"] and "at::" not in r["Name"]:
        print("%-60s calls %5s total_ms %9.2f avg_us %9.1f" % (r["Name"].replace("(anonymous namespace)::", "")[:60], r["Calls"], float(r["TotalDurationNs"]) / 1e6, float(r["AverageNs"]) / 1e3))
        tot += float(r["TotalDurationNs"]) / 1e6
print("our kernels total ms:", round(tot, 2))
PY
