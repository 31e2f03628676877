#!/usr/bin/env python3
"""Phase stamps of head_rows_kernel (csrc/head.hip: the head for more than 8 192 rows) at B = 512: wavefront 0 of the middle
workgroup.    python tools/head_rows_stamps.py"""
import os
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
import torch  # noqa: E402


def main():
    from bench import synthetic_batch
    from erc_amd import capi
    import track_mm.cogmen as plugin
    params = plugin.ParamsType().from_args(["--dataset=iemocap-cogmen-sbert-6", "--modality=atv", "--compute=bf16"])
    params.train.batch_size = 512
    tr = plugin.COGMENTrainer(params, "cuda:0")
    batch = tr.prepare_batch(synthetic_batch(params, 512, 110, seed=1))
    for _ in range(2):
        tr.train_step(batch)
    capi.start_recording()
    tr.train_step(batch)
    rec = capi.stop_recording()
    torch.cuda.synchronize()
    call = [e for e in rec if e[0] == "erc_head_fused"][0]
    st = torch.zeros(16, dtype=torch.int64, device="cuda:0")
    capi.head_set_stamps(st)
    acc = torch.zeros(16, dtype=torch.float64)
    reps = 10
    for _ in range(reps):
        capi.replay(call)
        torch.cuda.synchronize()
        s = st.cpu().double()
        acc += s - s[0]
    capi.head_set_stamps(None)
    acc /= reps
    labels = ["W0 + constants -> LDS (issue)", "P1: rows, BatchNorm, H3 out", "P2: product 1, Z out", "P3: logits, cross entropy", "P4: dZ -> LDS tile",
              "P5 / P6: product 2, dY, column partials", "barrier (slowest wavefront of the workgroup)", "record + drain"]
    prev = 0.0
    for k, lab in enumerate(labels):
        v = float(acc[k + 1])
        print("   %-48s %7.2f us   (at %7.2f)" % (lab, (v - prev) * 0.01, v * 0.01))
        prev = v


if __name__ == "__main__":
    main()
