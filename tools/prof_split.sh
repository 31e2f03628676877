#!/bin/bash
# On the GPU box: kernel-trace statistics of the COGMEN step in the split compute modes (gpurun_out/prof_<mode>/).
set -o pipefail
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
for m in "$@"; do
  rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_$m -- python3 $R/bench.py --dtype $m --steps 200 --warmup 10 --no_cpu_baseline --no_fp32_path --kernel_reps 20 > $R/gpurun_out/prof_$m.log 2>&1 || exit 1
  f=$(ls -t $R/gpurun_out/prof_$m/*/*kernel_stats.csv | head -1)
  echo "== $m"; head -12 $f | cut -c1-150
done
