#!/bin/bash
# usage: tools/pmc_pass.sh <tag> <counter> [<counter> ...]   (one rocprofv3 --pmc pass over a short bench run)
# The program after `--` is python itself (no env/bash hop), as the pool requires.
tag=$1; shift
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/pmc_$tag
timeout -k 10 300 rocprofv3 --pmc "$@" --kernel-trace --output-format csv -d gpurun_out/pmc_$tag -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline > gpurun_out/pmc_$tag.log 2>&1
echo "pmc $tag exit $?"
python3 tools/pmc_summarize.py gpurun_out/pmc_$tag
