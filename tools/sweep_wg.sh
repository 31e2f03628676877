#!/bin/bash
# On the GPU box: sweep the weight-gradient launch's k-steps per work item (ERC_WG_STEPS) at B = 512 and B = 32 (bf16 mode)
cd $GRAFT_REPO_ROOT
for s in 128 256 512 1024 2048; do
  echo "B=512 ERC_WG_STEPS=$s $(ERC_WG_STEPS=$s python bench.py --batch 512 --steps 30 --warmup 5 --no_cpu_baseline --no_fp32_path --kernel_reps 20 2>/dev/null | python -c 'import sys,json; d=json.loads([l for l in sys.stdin if l.startswith("{")][-1]); print("ms/step %.4f  dominant %s %.1f us" % (d["ms_per_step"], d["roofline"]["kernel"][:24], d["roofline"]["avg_us"]))')"
done
for s in 32 48 64 96 128; do
  echo "B=32 ERC_WG_STEPS=$s $(ERC_WG_STEPS=$s python bench.py --steps 200 --warmup 20 --no_cpu_baseline --no_fp32_path --kernel_reps 50 2>/dev/null | python -c 'import sys,json; d=json.loads([l for l in sys.stdin if l.startswith("{")][-1]); print("ms/step %.4f  dominant %s %.1f us" % (d["ms_per_step"], d["roofline"]["kernel"][:24], d["roofline"]["avg_us"]))')"
done
