"""diagnostic: every intermediate of a split-mode COGMEN step with the fused tile kernels against the same step on the unfused
exact-fp32 graph kernels (run on the GPU box)"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from tests.util_cases import cogmen_case, to_device
from erc_amd.cogmen import COGMENModule

compute = sys.argv[1] if len(sys.argv) > 1 else "f32x2"
case = cogmen_case(B=4, min_len=3, max_len=14, dims=dict(a=12, t=20, v=16), seed=3)
if len(sys.argv) > 2:
    case = cogmen_case(B=8, min_len=20, max_len=60, dims=dict(a=100, t=768, v=512), seed=6)
torch.manual_seed(0)
D, C = case["D"], case["n_classes"]
mods = []
for fused in (False, True):
    torch.manual_seed(0)
    m = COGMENModule(D, 100, 17, 2, C, compute=compute)
    with torch.no_grad():
        m.gcn.bn.weight.uniform_(0.5, 1.5), m.gcn.bn.bias.uniform_(-0.3, 0.3), m.gcn.conv1.bias.uniform_(-0.1, 0.1)
    m.finalize("cuda:0")
    m.use_fused_graph = fused
    m.train()
    m.drop_p = 0.0
    mods.append(m)
batch = to_device(case["batch"], "cuda:0")
out = []
for m in mods:
    m.loss_and_grads(batch)
    torch.cuda.synchronize()
    out.append(m._last_ws)
a, b = out
N = int(batch["label"].shape[0])
print("N", N, "fused flags", a.get("fused"), b.get("fused"))
for k in ("H0", "M", "inv_cnt", "H1", "QKVS", "alpha", "H2", "H3", "Z", "logits", "dlogits", "dZ", "dH3", "dQKVS", "dH1", "dH0"):
    x, y = a[k].double(), b[k].double()
    if k == "alpha":
        n_e = int(a["g"]["counts"][1])
        x, y = x[:n_e], y[:n_e]
    err = (x - y).abs()
    print("%-8s max|ref| %.3e  max err %.3e  rel %.2e   worst idx %s" % (k, float(x.abs().max()), float(err.max()), float(err.max() / (x.abs().max() + 1e-30)),
                                                                  tuple(int(v) for v in torch.nonzero(err == err.max())[0]) if bool((err == err.max()).any()) else 'nan'))
for name in mods[0].flat.params:
    x, y = mods[0].flat.g(name).double(), mods[1].flat.g(name).double()
    print("grad %-30s max|ref| %.3e err %.3e" % (name, float(x.abs().max()), float((x - y).abs().max())))
