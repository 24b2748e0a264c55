#!/bin/bash
# Round-4 evidence for the count operators on error-bearing counts, one MI355X box:
#   gpurun_out/r04_ops_realistic.jsonl           tools/bench_ops_realistic.py at 100 and 200 pools x 1 M loci (kernel ms = HIP events around
#                                                the first pass AND the second pass; listed fraction; five error / filter cases)
#   gpurun_out/r04_ops_realistic_kernel_stats.txt  rocprofv3 --kernel-trace --stats of the same script, one case per run
# usage: tools/r04_ops_realistic.sh   (copy both files into profiles/ afterwards)
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
: > gpurun_out/r04_ops_realistic.jsonl
for n in 100 200; do python3 tools/bench_ops_realistic.py $n 1000000 2>/dev/null | grep '^{' >> gpurun_out/r04_ops_realistic.jsonl; done
out=gpurun_out/r04_ops_realistic_kernel_stats.txt
echo "# rocprofv3 --kernel-trace --stats -- python3 tools/bench_ops_realistic.py 100 1000000 <error_rate> <maf>   (12 calls per operator: 2 warm-up + 10; our kernels only)" > $out
for c in "0.0 0.001" "0.005 0.01" "0.005 0.001"; do
  rm -rf gpurun_out/prof_real && mkdir -p gpurun_out/prof_real
  timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_real -- python3 tools/bench_ops_realistic.py 100 1000000 $c > gpurun_out/prof_real.log 2>&1
  echo "## error_rate, min_allele_frequency = $c (rc $?)" >> $out
  f=$(find gpurun_out/prof_real -name "*kernel_stats.csv" | head -1)
  python3 - "$f" >> $out <<'PY'
import csv, sys
for r in csv.DictReader(open(sys.argv[1])):
    if "k_" in r["Name"] and "at::" not in r["Name"]:
        print("%-64s calls %5s total_ms %9.2f avg_us %9.1f" % (r["Name"].replace("(anonymous namespace)::", "").replace("void ", "")[:64], r["Calls"], float(r["TotalDurationNs"]) / 1e6, float(r["AverageNs"]) / 1e3))
PY
  rm -rf gpurun_out/prof_real
done
cat $out | cut -c1-130
python3 - <<'PY'
import json
for l in open("gpurun_out/r04_ops_realistic.jsonl"):
    d = json.loads(l)
    print("%3d pools err %.3f maf %.3f %-13s %.3f ms  frac %.3f  listed %.4f" % (d["pools"], d["error_rate"], d["min_allele_frequency"], d["op"], d["kernel_ms"], d["frac_of_8TBs"], d["listed_fraction"]))
PY
