import sys; sys.path.insert(0, "/root/repo")
import torch
from tests.util_cases import cogmen_case, to_device
from erc_amd.cogmen import COGMENModule
case = cogmen_case(B=4, min_len=3, max_len=14, dims=dict(a=12, t=20, v=16), seed=3)
m = COGMENModule(case["D"], 100, 17, case["n_speakers"], case["n_classes"]).finalize("cuda:0")
m.train(); m.drop_p = 0.0
b = to_device(case["batch"], "cuda:0")
m.loss_and_grads(b)
ws = next(iter(m._ws.values()))
fp = m.flat
W0 = fp.w("cls.0.weight").view(100, 100)
dZ, H2, saved = ws["dZ"], ws["H2"], ws["bn_saved"]
ga, be = fp.w("gcn.bn.weight"), fp.w("gcn.bn.bias")
xh = (H2 - saved[:100]) * saved[100:]
zz = xh * ga + be
dY = (dZ @ W0) * torch.where(zz > 0, 1.0, 0.01)
got = ws["dH3"]
err = (got - dY).abs()
print("N", dZ.shape[0], "max err", float(err.max()), "ref max", float(dY.abs().max()))
bad = (err > 1e-6).nonzero()
print("bad count", bad.shape[0], "rows", sorted(set(bad[:, 0].tolist()))[:40])
print("cols", sorted(set(bad[:, 1].tolist()))[:120])
ratio = (got / dY)
print("ratios row0:", ratio[0, :8].tolist())
print("ratios row1:", ratio[1, :8].tolist())
raw = dZ @ W0
print("got/raw row0:", (got / raw)[0, :8].tolist())
print("lrelu' row0:", torch.where(zz > 0, 1.0, 0.01)[0, :8].tolist())
# does got row R match the reference of another row?
for R in (0, 1, 2, 4):
    d = ((dY - got[R:R+1]).abs().max(dim=1).values)
    print("got row", R, "closest ref row", int(d.argmin()), float(d.min()))
    d2 = ((raw - got[R:R+1]).abs().max(dim=1).values)
    print("   vs raw closest", int(d2.argmin()), float(d2.min()))
mask = torch.where(zz > 0, 1.0, 0.01)
got_raw = (got / mask).double()
dZp = torch.linalg.solve(W0.double().t(), got_raw.t()).t()   # dZ' W0 = got_raw
diff = (dZp - dZ.double()).abs()
print("tile reconstruction: max diff", float(diff.max()), "dZ max", float(dZ.abs().max()))
badm = diff > 1e-3 * float(dZ.abs().max())
for R in range(0, 8):
    cols = badm[R].nonzero().flatten().tolist()
    print("row", R, "bad cols", cols[:40], "n", len(cols))
R = 0
cols = badm[R].nonzero().flatten().tolist()[:6]
for c in cols:
    # which true dZ entry equals the reconstructed value?
    val = dZp[R, c]
    d = (dZ.double() - val).abs()
    idx = d.argmin()
    print("  row0 col", c, "holds", float(val), "true", float(dZ[R, c]), "matches dZ[", int(idx // 100), int(idx % 100), "] err", float(d.min()))
