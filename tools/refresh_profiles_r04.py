#!/usr/bin/env python3
"""Here, after tools/collect_profiles_r04.sh ran on the GPU box and gpurun merged gpurun_out/r4: write the tracked summaries
profiles/r04_<tag>_kernel_stats.csv (rocprofv3 --kernel-trace --stats) and profiles/r04_<tag>_<kernel>_pmc.json (HBM bytes per
launch from the separate FETCH_SIZE / WRITE_SIZE passes with the gfx950 correction of MI355X_MICROARCH.md -- read bytes = 2 x
FETCH_SIZE KiB -- and the matrix-core busy share SQ_VALU_MFMA_BUSY_CYCLES / (4 SIMDs x SQ_BUSY_CU_CYCLES) from its own pass)."""
import csv
import glob
import json
import os
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(REPO, "gpurun_out", "r4")
OUT = os.path.join(REPO, "profiles")
ROUND = "r04"
# tag -> kernels whose counters are summarised (substring of the kernel name -> file tag)
KERNELS = {
    "cogmen_b32_bf16": {"cogmen_project_graph_kernel": "project_graph", "cogmen_fwd_tile_kernel": "cogmen_fwd_tile", "head_fused_kernel": "head_fused",
                        "cogmen_bwd_tile_kernel": "cogmen_bwd_tile", "wgrad_bf16_kernel": "wgrad_bf16"},
    "cogmen_b32_f32x32": {"cogmen_project_graph_kernel": "project_graph", "cogmen_fwd_tile_kernel": "cogmen_fwd_tile", "head_fused_kernel": "head_fused",
                          "cogmen_bwd_tile_kernel": "cogmen_bwd_tile", "wgrad_bf16_kernel": "wgrad_bf16"},
    "dagerc": {"dag_rec_fwd_kernel": "dag_rec_fwd", "dag_rec_bwd_kernel": "dag_rec_bwd", "wgrad_table": "wgrad_table"},
    "mmgcn": {"gcnii_chain_kernel<false": "gcnii_chain_fwd", "gcnii_chain_kernel<true": "gcnii_chain_bwd", "gemm_x3_kernel": "gemm_x3",
              "lstm_fwd_kernel": "lstm_fwd", "lstm_bwd_kernel": "lstm_bwd", "wgrad_table": "wgrad_table"},
    "dgcn": {"lstm_fwd_kernel": "lstm_fwd", "lstm_bwd_kernel": "lstm_bwd", "brgcn_fwd_tile_kernel": "brgcn_fwd_tile",
             "brgcn_bwd_source_tile_kernel": "brgcn_bwd_source_tile", "brgcn_bwd_target_tile_kernel": "brgcn_bwd_target_tile", "wgrad_table": "wgrad_table", "dgcn_tail_kernel": "dgcn_tail"},
}


def newest(pattern):
    hits = glob.glob(pattern, recursive=True)
    return max(hits, key=os.path.getmtime) if hits else None


def counters(path):
    acc = {}
    if path is None:
        return acc
    for r in csv.DictReader(open(path)):
        acc.setdefault(r["Kernel_Name"], {}).setdefault(r["Counter_Name"], []).append(float(r["Counter_Value"]))
    return acc


def avg(acc, sub, name):
    vals = [v for k, d in acc.items() if sub in k for v in d.get(name, [])]
    return (sum(vals) / len(vals), len(vals)) if vals else (None, 0)


def main():
    os.makedirs(OUT, exist_ok=True)
    for tag_dir in sorted(glob.glob(os.path.join(SRC, "*"))):
        tag = os.path.basename(tag_dir)
        stats = newest(os.path.join(tag_dir, "trace", "**", "*kernel_stats.csv"))
        rows = []
        if stats:
            rows = list(csv.DictReader(open(stats)))
            keep = ["Name", "Calls", "TotalDurationNs", "AverageNs", "Percentage", "MinNs", "MaxNs", "StdDev"]
            with open(os.path.join(OUT, "%s_%s_kernel_stats.csv" % (ROUND, tag)), "w", newline="") as fh:
                w = csv.writer(fh)
                w.writerow(keep)
                for r in rows:
                    w.writerow([r[k] for k in keep])
        fetch = counters(newest(os.path.join(tag_dir, "fetch", "**", "*counter_collection.csv")))
        write = counters(newest(os.path.join(tag_dir, "write", "**", "*counter_collection.csv")))
        mfma = counters(newest(os.path.join(tag_dir, "mfma", "**", "*counter_collection.csv")))
        for sub, ktag in KERNELS.get(tag, {}).items():
            # kernel_substring: what bench.py matches against the label of the step's dominant kernel
            out = {"kernel_substring": {"gcnii_chain_fwd": "gcnii_chain_fwd_kernel", "gcnii_chain_bwd": "gcnii_chain_bwd_kernel"}.get(ktag, sub.split("<")[0]),
                   "kernel": sub}
            for r in rows:
                if sub in r["Name"]:
                    out["avg_duration_us"], out["calls"] = float(r["AverageNs"]) / 1e3, int(r["Calls"])
                    break
            f, nf = avg(fetch, sub, "FETCH_SIZE")
            w_, nw = avg(write, sub, "WRITE_SIZE")
            if f is not None and w_ is not None:
                out.update({"FETCH_SIZE_KiB_per_launch": f, "fetch_launches": nf, "WRITE_SIZE_KiB_per_launch": w_, "write_launches": nw,
                            "hbm_bytes_per_launch": (2.0 * f + w_) * 1024.0,
                            "correction": "read bytes = 2 x FETCH_SIZE KiB (gfx950 half-count of 128-B requests), writes exact"})
            b, nb = avg(mfma, sub, "SQ_VALU_MFMA_BUSY_CYCLES")
            c, _ = avg(mfma, sub, "SQ_BUSY_CU_CYCLES")
            if b is not None and c:
                out["mfma_busy"] = b / (4.0 * c)
                out["mfma_busy_note"] = "SQ_VALU_MFMA_BUSY_CYCLES / (4 SIMDs x SQ_BUSY_CU_CYCLES), own --pmc pass, %d launches" % nb
            if len(out) > 2:
                json.dump(out, open(os.path.join(OUT, "%s_%s_%s_pmc.json" % (ROUND, tag, ktag)), "w"), indent=1)
        # B = 512: what the projection waits for
        sq = counters(newest(os.path.join(tag_dir, "sq", "**", "*counter_collection.csv")))
        tc = counters(newest(os.path.join(tag_dir, "tc", "**", "*counter_collection.csv")))
        if sq or tc:
            out = {}
            for name, acc in (("sq", sq), ("tc", tc)):
                for k, d in acc.items():
                    kk = k.replace("(anonymous namespace)::", "").split("(")[0][:48]
                    if any(s in kk for s in ("gemm_bf16a_persist", "cogmen_fwd_tile", "cogmen_bwd_tile", "head_", "wgrad_bf16")):
                        out.setdefault(kk, {}).update({c: round(sum(v) / len(v)) for c, v in sorted(d.items())})
            json.dump({"per_launch_counter_averages": out,
                       "note": "B = 512 (N = 33 k nodes), eager launches; separate --pmc passes for the SQ and the TCP / TCC counters"},
                      open(os.path.join(OUT, "%s_%s_sq_tcp_pmc.json" % (ROUND, tag)), "w"), indent=1)
        print(tag, "kernel stats" if stats else "-", "pmc" if (fetch or write) else "-")


if __name__ == "__main__":
    main()
