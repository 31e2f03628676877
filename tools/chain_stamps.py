#!/usr/bin/env python3
"""Where a layer of MMGCN's persistent GCNII chain (csrc/gcnii_chain.hip) spends its time: replays the bench workload's
forward / backward chain launch with phase stamps switched on and prints workgroup 0's average per-phase time.

    python tools/chain_stamps.py [--batch 16]
"""
import argparse
import os
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
import torch  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=16)
    a = ap.parse_args()
    from bench import synthetic_batch
    from erc_amd import capi
    import track_mm.mmgcn as plugin
    params = plugin.ParamsType().from_args(["--dataset=iemocap-cogmen-sbert-6", "--modality=atv"])
    tr = plugin.MMGCNTrainer(params, "cuda:0")
    batch = tr.prepare_batch(synthetic_batch(params, a.batch, 110, seed=1))
    for _ in range(3):
        tr.train_step(batch)
    capi.start_recording()
    tr.train_step(batch)
    rec = capi.stop_recording()
    torch.cuda.synchronize()
    print("config (flag pitch, grid cap, dialogues per launch):", tr.model._last_ws["chain_cfg"])
    names = {"erc_gcnii_chain_fwd": {1: "row-local product h V", 2: "z -> exchange, LDS, save", 3: "drain", 4: "flag + wait",
                                     5: "gather + cross", 6: "block product A z (+ V_l+1 requests)", 7: "epilogue"},
             "erc_gcnii_chain_bwd": {2: "dg = dh . mask -> exchange", 3: "drain", 4: "flag + wait", 5: "gather + cross",
                                     6: "block product A dg (+ V_l requests)", 7: "dz -> LDS", 8: "dz saved, row-local dz V^T"}}
    for entry in names:
        call = [e for e in rec if e[0] == entry][0]
        st = torch.zeros(64, 16, dtype=torch.int64, device="cuda:0")
        capi.gcnii_chain_set_stamps(st)
        ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        ev0.record()
        capi.replay(call)
        ev1.record()
        torch.cuda.synchronize()
        capi.gcnii_chain_set_stamps(None)
        s = st.cpu().double()
        layers = list(range(8, 56))
        ms = ev0.elapsed_time(ev1)
        ticks_per_layer = float((s[55, 0] - s[8, 0]) / 47)
        us_per_tick = ms * 1e3 / 64 / ticks_per_layer           # calibrated on the launch itself
        print("== %s: %.3f ms, %.2f us / layer (workgroup 0: dialogue 0, modality 0, part 0)" % (entry, ms, ms * 1e3 / 64))
        prev = 0
        sub = {9: "   gather requests issued", 10: "   gathered rows in LDS"}
        for k in (9, 10):
            print("   %-36s +%5.2f us after the wait" % (sub[k], float((s[layers, k] - s[layers, 4]).mean()) * us_per_tick))
        for k in sorted(names[entry]):
            d = float((s[layers, k] - s[layers, prev]).mean()) * us_per_tick
            print("   %-40s %6.2f us" % (names[entry][k], d))
            prev = k
        nxt = float((s[[l + 1 for l in layers], 0] - s[layers, prev]).mean()) * us_per_tick
        print("   %-40s %6.2f us" % ("to the next layer's start", nxt))


if __name__ == "__main__":
    main()
