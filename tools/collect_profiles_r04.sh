#!/bin/bash
# On the GPU box (round 4): kernel traces and the PMC passes the roofline blocks of bench.py cite, for every module and for the
# COGMEN parity path.  Separate rocprofv3 runs per counter set (MI355X_MICROARCH.md: --pmc never together with the trace
# domains).  Everything lands under gpurun_out/r4/<tag>/{trace,fetch,write,mfma}; tools/refresh_profiles_r04.sh (here, after gpurun
# merged gpurun_out/) turns them into the tracked summaries profiles/r04_*.
set -o pipefail
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r4
cd /tmp && export TMPDIR=/tmp
run() {   # tag, bench arguments
  local tag=$1; shift
  mkdir -p $O/$tag
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/$tag/trace -- python3 $R/bench.py "$@" --no_cpu_baseline --no_fp32_path > $O/$tag/trace.log 2>&1 || return 1
  echo "$tag trace done"
}
pmc() {   # tag, bench arguments (eager: one dispatch record per launch)
  local tag=$1; shift
  rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/$tag/fetch -- python3 $R/bench.py "$@" --no_cpu_baseline --no_fp32_path --no_graph > $O/$tag/fetch.log 2>&1 || return 1
  rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/$tag/write -- python3 $R/bench.py "$@" --no_cpu_baseline --no_fp32_path --no_graph > $O/$tag/write.log 2>&1 || return 1
  rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES --output-format csv -d $O/$tag/mfma -- python3 $R/bench.py "$@" --no_cpu_baseline --no_fp32_path --no_graph > $O/$tag/mfma.log 2>&1 || return 1
  echo "$tag pmc done"
}
for dt in bf16 f32x32 f32x2 f32x3 f32; do
  run cogmen_b32_$dt --dtype $dt --steps 100 --warmup 10 --kernel_reps 20 || exit 1
done
pmc cogmen_b32_bf16 --dtype bf16 --steps 20 --warmup 3 --kernel_reps 10 || exit 1
pmc cogmen_b32_f32x32 --dtype f32x32 --steps 20 --warmup 3 --kernel_reps 10 || exit 1
run cogmen_b512_bf16 --batch 512 --steps 30 --warmup 5 --kernel_reps 10 || exit 1
# what the B = 512 projection waits for (SQ / TCP / TCC counters, two passes)
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INSTS_VMEM_RD --output-format csv -d $O/cogmen_b512_bf16/sq -- python3 $R/bench.py --batch 512 --steps 6 --warmup 2 --no_cpu_baseline --no_fp32_path --no_graph --kernel_reps 5 > $O/cogmen_b512_bf16/sq.log 2>&1 || echo "sq failed"
rocprofv3 --pmc TCP_TCC_READ_REQ_sum TCP_TCC_READ_REQ_LATENCY_sum TCC_HIT_sum TCC_MISS_sum TCP_PENDING_STALL_CYCLES_sum TA_BUSY_sum --output-format csv -d $O/cogmen_b512_bf16/tc -- python3 $R/bench.py --batch 512 --steps 6 --warmup 2 --no_cpu_baseline --no_fp32_path --no_graph --kernel_reps 5 > $O/cogmen_b512_bf16/tc.log 2>&1 || echo "tc failed"
echo "b512 counters done"
for m in dgcn mmgcn dagerc; do
  st=100; [ $m = dagerc ] && st=10; [ $m = mmgcn ] && st=20
  run $m --module $m --steps $st --warmup 3 --kernel_reps 10 || exit 1
  ps=10; [ $m = dagerc ] && ps=4
  pmc $m --module $m --steps $ps --warmup 2 --kernel_reps 5 || exit 1
done
echo collected
