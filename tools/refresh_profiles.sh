#!/bin/bash
# Here (after tools/collect_profiles.sh + tools/pmc_mfma.sh ran on the GPU box and gpurun merged gpurun_out/): rebuild the
# tracked summaries under profiles/ from the newest run of every directory.
set -e
cd "$(dirname "$0")/.."
R=${ROUND:-r03}
T=${R}_cogmen_b32_bf16
python tools/pmc_summary.py --trace gpurun_out/prof_r3 --fetch gpurun_out/pmc_fetch --write gpurun_out/pmc_write --tag $T --kernel wgrad_bf16 > /dev/null
cp profiles/${T}_pmc.json profiles/${T}_wgrad_bf16_pmc.json
rm profiles/${T}_pmc.json
for k in cogmen_project_graph:project_graph cogmen_fwd_tile:cogmen_fwd_tile cogmen_bwd_tile:cogmen_bwd_tile head_fused:head_fused; do
  kn=${k%%:*}; tg=${k##*:}
  python tools/pmc_summary.py --fetch gpurun_out/pmc_fetch --write gpurun_out/pmc_write --trace gpurun_out/prof_r3 --tag ${T}_$tg --kernel $kn --out /tmp/pmcs > /dev/null
  cp /tmp/pmcs/${T}_${tg}_pmc.json profiles/
done
for m in dagerc dgcn mmgcn; do
  python tools/pmc_summary.py --trace gpurun_out/prof_$m --tag ${R}_$m --kernel zzz --out /tmp/pmcs > /dev/null
  cp /tmp/pmcs/${R}_${m}_kernel_stats.csv profiles/
done
python - <<'PY'
import csv, glob, os, json
f = max(glob.glob("gpurun_out/pmc_mfma/**/*counter_collection.csv", recursive=True), key=os.path.getmtime)
acc = {}
for r in csv.DictReader(open(f)):
    acc.setdefault(r["Kernel_Name"], {}).setdefault(r["Counter_Name"], []).append(float(r["Counter_Value"]))
out = {}
for k, d in acc.items():
    if "SQ_VALU_MFMA_BUSY_CYCLES" in d and "SQ_BUSY_CU_CYCLES" in d:
        busy = sum(d["SQ_VALU_MFMA_BUSY_CYCLES"]) / len(d["SQ_VALU_MFMA_BUSY_CYCLES"])
        cu = sum(d["SQ_BUSY_CU_CYCLES"]) / len(d["SQ_BUSY_CU_CYCLES"])
        if cu > 0 and busy > 0:
            out[k.replace("(anonymous namespace)::", "").split("(")[0]] = round(busy / (4 * cu), 4)
p = "profiles/%s_cogmen_b32_bf16_mfma_pmc.json" % os.environ.get("ROUND", "r03")
key = "mfma_busy = SQ_VALU_MFMA_BUSY_CYCLES / (4 SIMDs x SQ_BUSY_CU_CYCLES), own --pmc pass (tools/pmc_mfma.sh)"
old = json.load(open(p)) if os.path.exists(p) else {}
key = next((k for k in old if k.startswith("mfma_busy")), key)
old[key] = dict(sorted(out.items(), key=lambda kv: -kv[1]))
json.dump(old, open(p, "w"), indent=1)
PY
python tools/top_kernels.py gpurun_out/prof_r3 8
