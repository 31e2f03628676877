#!/bin/bash
# On the GPU box: kernel-trace statistics of the COGMEN step at B = 512 (the throughput point), gpurun_out/prof_b512_<mode>/
set -o pipefail
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
for m in "$@"; do
  rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_b512_$m -- python3 $R/bench.py --dtype $m --batch 512 --steps 30 --warmup 5 --no_cpu_baseline --no_fp32_path --kernel_reps 10 > $R/gpurun_out/prof_b512_$m.log 2>&1 || exit 1
  f=$(ls -t $R/gpurun_out/prof_b512_$m/*/*kernel_stats.csv | head -1)
  echo "== $m"; tail -1 $R/gpurun_out/prof_b512_$m.log | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('ms_per_step', d['ms_per_step'], 'value', d['value'])"
  python3 - "$f" <<'PY'
import csv,sys
for r in list(csv.DictReader(open(sys.argv[1])))[:9]:
    print("%-70s calls %4s avg %8.2f us" % (r["Name"][28:98], r["Calls"], float(r["AverageNs"])/1e3))
PY
done
