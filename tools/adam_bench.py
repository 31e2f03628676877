"""Time the optimizer launch of the COGMEN bf16 step alone (200 launches captured in one HIP graph):
with the shadow table (ERC_ADAM_SEGS=0/1 selects the work decomposition) and without any shadow."""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from erc_amd import capi
from erc_amd.cogmen import COGMENTrainer
from erc_amd.params import ERCParams


def timed(fn, reps=200):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(reps):
            fn()
    g.replay()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); g.replay(); e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / reps


p = ERCParams().from_args(["--dataset=iemocap-cogmen-sbert-6", "--compute=bf16"])
tr = COGMENTrainer(p, "cuda:0")
tr.model.flat.grad.normal_()
print("with shadow table (ERC_ADAM_SEGS=%s): %.2f us" % (os.environ.get("ERC_ADAM_SEGS", "1"), timed(tr.optim.step)))
tab = tr.optim.shadow_table
tr.optim.shadow_table = None
print("no shadows: %.2f us" % timed(tr.optim.step))
