#!/bin/bash
# usage: tools/pmc_ops.sh <tag> <pools> <loci> <counter> [<counter> ...]  (one rocprofv3 --pmc pass over tools/bench_ops.py)
tag=$1; pools=$2; loci=$3; shift 3
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
rm -rf gpurun_out/pmc_$tag && mkdir -p gpurun_out/pmc_$tag
timeout -k 10 120 rocprofv3 --pmc "$@" --kernel-trace --output-format csv -d gpurun_out/pmc_$tag -- python3 tools/bench_ops.py $pools $loci > gpurun_out/pmc_$tag.log 2>&1
echo "pmc $tag exit $?"
python3 tools/pmc_summarize.py gpurun_out/pmc_$tag
