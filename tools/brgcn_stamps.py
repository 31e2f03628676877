"""Phase times of erc_brgcn_fwd_tile inside a DialogueGCN step: python tools/brgcn_stamps.py"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402


def main():
    from bench import synthetic_batch
    from erc_amd import capi
    import track_mm.dgcn as plugin
    params = plugin.ParamsType().from_args(["--dataset=meld-mmgcn-7", "--modality=atv", "--reimplement", "--loss_weights=False"])
    tr = plugin.DGCNTrainer(params, "cuda:0")
    batch = tr.prepare_batch(synthetic_batch(params, 32, 33, seed=1))
    for _ in range(3):
        tr.train_step(batch)
    st = torch.zeros(16, dtype=torch.int64, device="cuda:0")
    capi.brgcn_set_stamps(st)
    tr.train_step(batch)
    torch.cuda.synchronize()
    capi.brgcn_set_stamps(None)
    t = st.cpu().tolist()
    names = ["aggregate (2 nodes per wavefront)", "barrier", "matrix product (K split over 8 wavefronts)", "barrier",
             "partials -> slab"]
    for k, n in enumerate(names):
        print("  %-48s %6.2f us" % (n, (t[k + 1] - t[k]) / 100.0))
    print("  total %.2f us" % ((t[5] - t[0]) / 100.0))
    print("edge-side backward tile (thread 0 = wavefront 0):")
    for k, n in enumerate(["dZ blocks (matrix cores)", "barrier", "dots, first node", "dots, second node"]):
        print("  %-48s %6.2f us" % (n, (t[8 + k + 1] - t[8 + k]) / 100.0))


main()
