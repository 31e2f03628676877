#!/bin/bash
# On the GPU box: kernel traces of every module + the two PMC passes of the COGMEN bench (separate runs, as the
# MI355X guide prescribes), then default bench lines.  Outputs under gpurun_out/.
set -o pipefail
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_r3 -- python3 $R/bench.py --steps 100 --warmup 10 --no_cpu_baseline --no_fp32_path > $R/gpurun_out/prof_r3.log 2>&1 || exit 1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $R/gpurun_out/pmc_fetch -- python3 $R/bench.py --steps 20 --warmup 3 --no_cpu_baseline --no_fp32_path --no_graph > $R/gpurun_out/pmc_fetch.log 2>&1 || exit 1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $R/gpurun_out/pmc_write -- python3 $R/bench.py --steps 20 --warmup 3 --no_cpu_baseline --no_fp32_path --no_graph > $R/gpurun_out/pmc_write.log 2>&1 || exit 1
for m in dagerc dgcn mmgcn; do
  st=100; [ $m = dagerc ] && st=10
  rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_$m -- python3 $R/bench.py --module $m --steps $st --warmup 3 --no_cpu_baseline --no_fp32_path > $R/gpurun_out/prof_$m.log 2>&1 || exit 1
done
cd $R
python bench.py > gpurun_out/bench_cogmen.log 2>&1 || exit 1
python bench.py --dtype f32 --no_cpu_baseline --no_fp32_path > gpurun_out/bench_cogmen_f32.log 2>&1 || exit 1
python bench.py --batch 512 --no_cpu_baseline --no_fp32_path --steps 50 --warmup 5 > gpurun_out/bench_cogmen_b512.log 2>&1 || exit 1
for m in dgcn mmgcn; do python bench.py --module $m > gpurun_out/bench_$m.log 2>&1 || exit 1; done
python bench.py --module dagerc --steps 20 --warmup 3 > gpurun_out/bench_dagerc.log 2>&1 || exit 1
python bench.py --rehearse_dp --no_cpu_baseline --no_fp32_path > gpurun_out/bench_dp.log 2>&1 || exit 1
echo collected
