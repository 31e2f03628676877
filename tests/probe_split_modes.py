"""diagnostic: which launch of a split-mode COGMEN step carries the deviation from the fp32 oracle (run on the GPU box)"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from tests.util_cases import cogmen_case, run_cogmen_parity
import erc_amd.cogmen as cg
from erc_amd import engine

case = cogmen_case(B=32, min_len=20, max_len=110, dims=dict(a=100, t=768, v=512), seed=1)


def show(tag, res):
    top = sorted(res["grad_errs"].items(), key=lambda kv: -kv[1])[:4]
    print("%-34s logits %.2e  grad %.2e  %s" % (tag, res["logit_err"], res["grad_err"], ", ".join("%s %.1e" % kv for kv in top)), flush=True)


show("f32 (unfused exact)", run_cogmen_parity(case, compute="f32"))
show("f32x2", run_cogmen_parity(case, compute="f32x2"))
show("f32x3", run_cogmen_parity(case, compute="f32x3"))
# projection exact, weight gradients two terms
orig = cg.COGMENModule.__init__
def init_nopg(self, *a, **k):
    orig(self, *a, **k)
    self.fuse_project_graph = False
cg.COGMENModule.__init__ = init_nopg
show("f32x2, fp32 projection", run_cogmen_parity(case, compute="f32x2"))
cg.COGMENModule.__init__ = orig
# projection two terms, weight gradients three terms
d16 = engine.GemmPlanner.defer16
def d16_3(self, *a, **k):
    self.split_terms = 3
    return d16(self, *a, **k)
engine.GemmPlanner.defer16 = d16_3
show("f32x2, weight gradients x3", run_cogmen_parity(case, compute="f32x2"))
engine.GemmPlanner.defer16 = d16
