#!/bin/bash
# On the GPU box: LDS bank-conflict share per kernel for every module's train step (own --pmc pass).
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
for m in cogmen dgcn mmgcn dagerc; do
  st=20; [ $m = dagerc ] && st=3
  rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_BUSY_CU_CYCLES --output-format csv -d $R/gpurun_out/pmc_lds_$m -- python3 $R/bench.py --module $m --steps $st --warmup 2 --no_cpu_baseline --no_fp32_path --no_graph > $R/gpurun_out/pmc_lds_$m.log 2>&1 || echo "$m failed"
done
echo done
