#!/usr/bin/env python3
"""Where workgroup 0 of erc_dgcn_tail (csrc/dgcn_tail.hip) spends its time: the bench workload's launch replayed with phase stamps."""
import importlib
import os
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
import torch  # noqa: E402


def main():
    import bench
    from erc_amd import capi
    plugin = importlib.import_module("track_mm.dgcn")
    params = plugin.ParamsType().from_args(["--dataset=meld-mmgcn-7", "--modality=atv", "--compute=bf16", "--loss_weights=False"])
    params.train.batch_size = 32
    tr = plugin.DGCNTrainer(params, torch.device("cuda:0"))
    b = tr.prepare_batch(bench.synthetic_batch(params, 32, 33, seed=1))
    for _ in range(3):
        tr.train_step(b)
    capi.start_recording()
    tr.train_step(b)
    rec = capi.stop_recording()
    torch.cuda.synchronize()
    call = [e for e in rec if e[0] == "erc_dgcn_tail"][0]
    st = torch.zeros(16, dtype=torch.int64, device="cuda:0")
    capi.dgcn_tail_set_stamps(st)
    acc = torch.zeros(16, dtype=torch.float64)
    reps = 20
    for _ in range(reps):
        capi.replay(call)
        torch.cuda.synchronize()
        s = st.cpu().double()
        acc += s - s[0]
    capi.dgcn_tail_set_stamps(None)
    acc /= reps
    labels = {1: "CSR bounds in LDS", 2: "window Hc rows, features, edge list in LDS", 4: "AGG sums", 5: "GraphConv product",
              6: "lin1", 7: "lin2 partials", 8: "cross entropy", 9: "dZc", 10: "dXc", 15: "dAGG / dHc"}
    prev = 0.0
    for k in sorted(labels):
        v = float(acc[k])
        print("   %-46s %6.2f us   (at %6.2f)" % (labels[k], (v - prev) * 0.01, v * 0.01))
        prev = v
    t0 = torch.cuda.Event(enable_timing=True)
    t1 = torch.cuda.Event(enable_timing=True)
    t0.record()
    for _ in range(50):
        capi.replay(call)
    t1.record()
    torch.cuda.synchronize()
    print("launch, back to back: %.2f us" % (t0.elapsed_time(t1) * 1e3 / 50))


if __name__ == "__main__":
    main()
