#!/bin/bash
# A/B of the second stream for the weight gradients that do not wait for the BiLSTM backward (DialogueGCN, MMGCN)
set -o pipefail
O=$GRAFT_REPO_ROOT/gpurun_out/side
mkdir -p $O
for m in "$@"; do
  for s in 0 1; do
    ERC_SIDE_STREAM=$s python3 bench.py --module $m --steps 200 --warmup 10 --no_cpu_baseline --no_fp32_path --kernel_reps 5 > $O/${m}_$s.log 2>$O/${m}_$s.err || exit 1
    python3 - <<PY
import json
for l in open("$O/${m}_$s.log"):
    if l.startswith("{"):
        d=json.loads(l); print("$m side=$s", d["ms_per_step"], d["value"])
PY
  done
done
