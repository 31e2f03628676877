for st in 48 64 96 128; do
  ERC_WG_STEPS=$st timeout -k 10 120 python bench.py --no_cpu_baseline --steps 300 --warmup 30 2>/dev/null | grep -o '"ms_per_step": [0-9.]*' | sed "s/^/steps=$st /"
done
