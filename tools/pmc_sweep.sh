#!/bin/bash
# usage: tools/pmc_sweep.sh <tag> <variants> -- rocprofv3 --pmc passes over tools/bench_sweep.py (2 reps), per-kernel means
tag=$1; variants=$2
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
export SWEEP_VARIANTS=$variants
i=0
for ctrs in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY" \
            "SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_SMEM SQ_INSTS_SALU SQ_WAIT_INST_LDS SQ_INSTS_VMEM_RD GRBM_GUI_ACTIVE" \
            "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum"; do
  i=$((i+1))
  rm -rf gpurun_out/pmc_${tag}_$i && mkdir -p gpurun_out/pmc_${tag}_$i
  timeout -k 10 300 rocprofv3 --pmc $ctrs --kernel-trace --output-format csv -d gpurun_out/pmc_${tag}_$i -- python3 tools/bench_sweep.py 200 10000000 2 > gpurun_out/pmc_${tag}_$i.log 2>&1
  echo "pass $i exit $?"
  f=$(find gpurun_out/pmc_${tag}_$i -name "*counter_collection.csv" | head -1)
  python3 - "$f" <<'PY' >> gpurun_out/pmc_${tag}_summary.txt
import csv, sys, collections
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open(sys.argv[1])):
    name = r["Kernel_Name"].replace("(anonymous namespace)::", "").split("(")[0][:40]
    if "k_ols_sweep" not in name: continue
    acc[name][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, d in sorted(acc.items()):
    for c, v in d.items():
        print(f"{k:36s} {c:28s} n={len(v):3d} mean={sum(v)/len(v):.6g}")
PY
  rm -rf gpurun_out/pmc_${tag}_$i
done
cat gpurun_out/pmc_${tag}_summary.txt
