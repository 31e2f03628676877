#!/usr/bin/env python3
"""Where the two fused COGMEN graph kernels (csrc/cogmen_fused.hip) spend their time: replays the bench workload's
launches with phase stamps switched on and prints the middle workgroup's per-phase time.

    python tools/cogmen_stamps.py [--batch 32]
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
    a = ap.parse_args()
    from bench import synthetic_batch
    from erc_amd import capi
    import track_mm.cogmen as plugin
    params = plugin.ParamsType().from_args(["--dataset=iemocap-cogmen-sbert-6", "--modality=atv", "--compute=bf16"])
    tr = plugin.COGMENTrainer(params, "cuda:0")
    batch = tr.prepare_batch(synthetic_batch(params, a.batch, 110, seed=1))
    for _ in range(3):
        tr.train_step(batch)
    capi.start_recording()
    tr.train_step(batch)
    rec = capi.stop_recording()
    torch.cuda.synchronize()
    names = {"erc_cogmen_fwd_tile": ["H0 rows -> LDS", "relation means", "H1 product", "QKVS product", "QKVS store + attention",
                                     "BatchNorm partials", "(arrival ..) last arriver done"],
             "erc_head_fused_bn": ["BatchNorm sums + W0 -> LDS + A fragments", "product 1 (Z), partial logits", "cross entropy", "dZ -> LDS",
                                   "product 2 (dH3), dY, column partials", "partial record + drain", "(arrival ..) last arriver done"],
             "erc_cogmen_bwd_tile": ["slices + tiles -> LDS", "band dA + d(score)", "dq / dk / dv band products", "dH1 product", "dP", "dH0 product"]}
    for entry, labels in names.items():
        call = [e for e in rec if e[0] == entry][0]
        st = torch.zeros(16, dtype=torch.int64, device="cuda:0")
        (capi.head_set_stamps if "head" in entry else capi.cogmen_set_stamps)(st)
        acc = torch.zeros(16, dtype=torch.float64)
        reps = 20
        ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        ev0.record()
        for _ in range(reps):
            capi.replay(call)
            torch.cuda.synchronize()
            s = st.cpu().double()
            acc += s - s[0]
        ev1.record()
        torch.cuda.synchronize()
        (capi.head_set_stamps if "head" in entry else capi.cogmen_set_stamps)(None)
        acc /= reps
        print("== %s (middle workgroup, 10 ns ticks)" % entry)
        prev = 0.0
        for k, lab in enumerate(labels):
            v = float(acc[k + 1])
            print("   %-34s %6.2f us   (at %6.2f)" % (lab, (v - prev) * 0.01, v * 0.01))
            prev = v


if __name__ == "__main__":
    main()
