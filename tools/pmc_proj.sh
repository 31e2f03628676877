#!/bin/bash
# On the GPU box: SQ / TCP / TCC counters of the COGMEN step at B = 512 (two --pmc passes), to see what the input projection
# waits for.  Summarise here with: python tools/pmc_proj_summary.py
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INSTS_VMEM_RD --output-format csv -d $R/gpurun_out/pmc_proj_sq -- python3 $R/bench.py --batch 512 --steps 6 --warmup 2 --no_cpu_baseline --no_fp32_path --no_graph > $R/gpurun_out/pmc_proj_sq.log 2>&1 || echo sq failed
rocprofv3 --pmc TCP_TCC_READ_REQ_sum TCP_TCC_READ_REQ_LATENCY_sum TCC_HIT_sum TCC_MISS_sum TCP_PENDING_STALL_CYCLES_sum TA_BUSY_sum --output-format csv -d $R/gpurun_out/pmc_proj_tc -- python3 $R/bench.py --batch 512 --steps 6 --warmup 2 --no_cpu_baseline --no_fp32_path --no_graph > $R/gpurun_out/pmc_proj_tc.log 2>&1 || echo tc failed
echo done
