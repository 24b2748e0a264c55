#!/bin/bash
# usage: tools/pmc_any.sh <tag> "<counters>" <python script> [args...] -- one rocprofv3 --pmc pass; prints per-kernel means of the counters
tag=$1; ctrs=$2; shift 2
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
rm -rf gpurun_out/pmc_$tag && mkdir -p gpurun_out/pmc_$tag
timeout -k 10 300 rocprofv3 --pmc $ctrs --kernel-trace --output-format csv -d gpurun_out/pmc_$tag -- python3 "$@" > gpurun_out/pmc_$tag.log 2>&1
echo "pmc $tag exit $?"
f=$(find gpurun_out/pmc_$tag -name "*counter_collection.csv" | head -1)
python3 - "$f" <<'PY'
import csv, sys, collections
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open(sys.argv[1])):
    name = r["Kernel_Name"].replace("(anonymous namespace)::", "")[:40]
    acc[name][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, d in acc.items():
    if "k_" not in k or "at::" in k: continue
    print(k, {c: "%.3g" % (sum(v) / len(v)) for c, v in d.items()}, "launches", len(next(iter(d.values()))))
PY
