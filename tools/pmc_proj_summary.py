"""Per-launch averages of the counters collected by tools/pmc_proj.sh for the kernels of the COGMEN step."""
import csv, glob, os, sys
for d in ("gpurun_out/pmc_proj_sq", "gpurun_out/pmc_proj_tc"):
    fs = glob.glob(d + "/**/*counter_collection.csv", recursive=True)
    if not fs:
        print("no data in", d)
        continue
    f = max(fs, key=os.path.getmtime)
    acc = {}
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].replace("(anonymous namespace)::", "").split("(")[0][:34]
        acc.setdefault(k, {}).setdefault(r["Counter_Name"], []).append(float(r["Counter_Value"]))
    want = sys.argv[1:] or ["gemm_bf16a_persist", "wgrad_table", "cogmen_fwd_tile", "cogmen_bwd_tile", "head_fused"]
    for k, dct in acc.items():
        if any(w in k for w in want):
            print(k, {c: round(sum(v) / len(v)) for c, v in sorted(dct.items())}, "launches", len(next(iter(dct.values()))))
