#!/usr/bin/env python3
"""Where a work item of the bf16 weight-gradient launch (csrc/wgrad_bf16.hip) spends its time: replays the bench
workload's launch with phase stamps on for one item of every record's first tile.

    python tools/wgrad_stamps.py [--batch 32]
"""
import argparse
import os
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
import torch  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=32)
    ap.add_argument("--compute", default="bf16", help="bf16 | f32x2 | f32x3")
    a = ap.parse_args()
    from bench import synthetic_batch
    from erc_amd import capi
    import track_mm.cogmen as plugin
    params = plugin.ParamsType().from_args(["--dataset=iemocap-cogmen-sbert-6", "--modality=atv", "--compute=" + a.compute])
    tr = plugin.COGMENTrainer(params, "cuda:0")
    batch = tr.prepare_batch(synthetic_batch(params, a.batch, 110, seed=1))
    for _ in range(3):
        tr.train_step(batch)
    capi.start_recording()
    tr.train_step(batch)
    rec = capi.stop_recording()
    torch.cuda.synchronize()
    call = [e for e in rec if e[0] in ("erc_wgrad_bf16", "erc_wgrad_bf16_wide", "erc_wgrad_bf16_adam", "erc_wgrad_split", "erc_wgrad_split_adam")][0]
    print(call[0])
    n_items = call[1][4 if "split" in call[0] else 3]
    labels = ["first loads issued (A, gather stage, B)", "K loop", "LDS reduce + slab stores issued", "slab drained", "arrival ticket"]
    for item in (0, 1, 2, 3, n_items // 2):
        st = torch.zeros(16, dtype=torch.int64, device="cuda:0")
        capi.wgrad_bf16_set_stamps(st, item)
        acc = torch.zeros(16, dtype=torch.float64)
        reps = 20
        for _ in range(reps):
            capi.replay(call)
            torch.cuda.synchronize()
            s = st.cpu().double()
            acc += s - s[0]
        capi.wgrad_bf16_set_stamps(None)
        acc /= reps
        print("== work item %d of %d (10 ns ticks)" % (item, n_items))
        prev = 0.0
        for k, lab in enumerate(labels):
            v = float(acc[k + 1])
            print("   %-42s %6.2f us   (at %6.2f)" % (lab, (v - prev) * 0.01, v * 0.01))
            prev = v
        if call[0].endswith("adam"):
            print("   fused optimizer: all splits arrived %6.2f, slabs summed %6.2f, quad 0 updated %6.2f, + shadows %6.2f, quad 1 %6.2f, bias strip %6.2f" % tuple(
                float(acc[k]) * 0.01 for k in (8, 9, 13, 11, 12, 10)))
        elif item < call[1][1] * 0 + 64:
            print("   last arriver of the tile: starts at %6.2f, slabs summed at %6.2f, stored at %6.2f" % (
                float(acc[8]) * 0.01, float(acc[9]) * 0.01, float(acc[10]) * 0.01))


if __name__ == "__main__":
    main()
