#!/bin/bash
# usage: tools/pmc_kin.sh <tag> <pools> [pools ..] -- HBM traffic of the plain kinship pass per pool count (FETCH_SIZE / WRITE_SIZE passes
# of rocprofv3 over tools/bench_kinship_n.py), with and without the XCD placement of the block pairs (POOLGEN_KIN_NO_XCD=1)
tag=$1; shift
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
: > gpurun_out/pmc_kin_${tag}.txt
for mode in xcd noxcd; do
  if [ $mode = noxcd ]; then export POOLGEN_KIN_NO_XCD=1; else unset POOLGEN_KIN_NO_XCD; fi
  python3 tools/bench_kinship_n.py "$@" 2>/dev/null | sed "s/^/[$mode] /" >> gpurun_out/pmc_kin_${tag}.txt
  for ctr in FETCH_SIZE WRITE_SIZE; do
    rm -rf gpurun_out/pmc_kin_tmp && mkdir -p gpurun_out/pmc_kin_tmp
    timeout -k 10 300 rocprofv3 --pmc $ctr --kernel-trace --output-format csv -d gpurun_out/pmc_kin_tmp -- python3 tools/bench_kinship_n.py "$@" > gpurun_out/pmc_kin_tmp.log 2>&1
    f=$(find gpurun_out/pmc_kin_tmp -name "*counter_collection.csv" | head -1)
    python3 - "$f" "$mode" >> gpurun_out/pmc_kin_${tag}.txt <<'PY'
import csv, sys
rows = [r for r in csv.DictReader(open(sys.argv[1])) if "k_kinship_syrk" in r["Kernel_Name"]]
rows.sort(key=lambda r: int(r["Dispatch_Id"]))
for i in range(0, len(rows), 3):   # tools/bench_kinship_n.py launches the pass three times per pool count
    v = [float(r["Counter_Value"]) for r in rows[i:i + 3]]
    print(f"[{sys.argv[2]}] pool count #{i // 3}: grid {rows[i]['Grid_Size']:>8s} {rows[i]['Counter_Name']:12s} mean of {len(v)} launches = {sum(v) / len(v):.6g} KiB")
PY
    rm -rf gpurun_out/pmc_kin_tmp
  done
done
cat gpurun_out/pmc_kin_${tag}.txt
