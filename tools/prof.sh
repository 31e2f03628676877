#!/bin/bash
# usage (on the GPU box): tools/prof.sh <tag> <bench.py args...>  -> gpurun_out/<tag>/…_kernel_stats.csv
tag=$1; shift
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/$tag -- python3 $GRAFT_REPO_ROOT/bench.py --no_cpu_baseline --no_fp32_path "$@" > $GRAFT_REPO_ROOT/gpurun_out/$tag.log 2>&1
