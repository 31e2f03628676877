#!/bin/bash
# On the GPU box: kernel-trace statistics of one module's bench step: tools/prof_module.sh <module> [bench args]; gpurun_out/prof_<module>/
set -o pipefail
R=$GRAFT_REPO_ROOT
m=$1; shift
cd /tmp && export TMPDIR=/tmp
rm -rf $R/gpurun_out/prof_$m
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_$m -- python3 $R/bench.py --module $m --steps 50 --warmup 5 --no_cpu_baseline --no_fp32_path --kernel_reps 5 "$@" > $R/gpurun_out/prof_$m.log 2>&1 || exit 1
f=$(ls -t $R/gpurun_out/prof_$m/*/*kernel_stats.csv | head -1)
tail -1 $R/gpurun_out/prof_$m.log | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('ms_per_step', d['ms_per_step'], 'value', d['value'])"
python3 - "$f" <<'PY'
import csv,sys
for r in list(csv.DictReader(open(sys.argv[1])))[:26]:
    print("%-64s calls %4s avg %8.2f us" % (r["Name"].replace("(anonymous namespace)::","")[:64], r["Calls"], float(r["AverageNs"])/1e3))
PY
