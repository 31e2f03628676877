#!/bin/bash
# On the GPU box: MFMA busy share per kernel of the COGMEN step (own --pmc pass).
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F32 --output-format csv -d $R/gpurun_out/pmc_mfma -- python3 $R/bench.py --steps 20 --warmup 3 --no_cpu_baseline --no_fp32_path --no_graph > $R/gpurun_out/pmc_mfma.log 2>&1 || echo failed
echo done
