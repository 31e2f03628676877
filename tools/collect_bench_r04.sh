#!/bin/bash
# On the GPU box (round 4): the bench lines quoted in BASELINE.md / DESIGN.md -> gpurun_out/r4/bench_*.log
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r4
mkdir -p $O
cd $R
python bench.py > $O/bench_cogmen.log 2>$O/bench_cogmen.err || exit 1
echo "cogmen done"
python bench.py --batch 512 --no_cpu_baseline --no_fp32_path --steps 50 --warmup 5 > $O/bench_cogmen_b512.log 2>/dev/null || exit 1
python bench.py --dtype f32x32 --batch 512 --no_cpu_baseline --no_fp32_path --steps 50 --warmup 5 > $O/bench_cogmen_b512_f32x32.log 2>/dev/null || exit 1
python bench.py --rehearse_dp --no_cpu_baseline --no_fp32_path > $O/bench_dp.log 2>/dev/null || exit 1
python bench.py --dtype f32x32 --rehearse_dp --no_cpu_baseline --no_fp32_path > $O/bench_dp_f32x32.log 2>/dev/null || exit 1
echo "cogmen variants done"
for m in dgcn mmgcn; do python bench.py --module $m > $O/bench_$m.log 2>/dev/null || exit 1; done
python bench.py --module dagerc --steps 20 --warmup 3 > $O/bench_dagerc.log 2>/dev/null || exit 1
echo collected
