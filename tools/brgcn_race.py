"""debug: the fused RGCN tile kernels inside a guarded arena, every check on the host (no device reductions)"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from erc_amd import capi
from erc_amd.cogmen import build_graph_tensors
DEV = "cuda:0"
torch.manual_seed(0)
B, T, S, NB, Fd, O = int(os.environ.get("NB_DLG", "6")), 33, 9, 30, 200, 100
R = 2 * S * S
lengths = torch.randint(1, T + 1, (B,))
spk = torch.randint(0, S, (B, T))
g, ei, et = build_graph_tensors(lengths.to(DEV), spk.to(DEV), 10, 10, S)
N, E = g["counts"].cpu().tolist()
XW = Fd + O
Sg = capi.brgcn_fwd_tile_slabs()
GUARD = 1 << 20
sizes = dict(xw=N * XW, norm=E, att=R * NB, basis=NB * Fd * O, root=Fd * O, Z=N * NB * Fd, sl=Sg * N * O, gout=N * O,
             dsl=Sg * N * Fd, TT=E * NB, dn=Sg * E, datt=R * NB)
off, views = GUARD, {}
for k, n in sizes.items():
    off = (off + 63) // 64 * 64
    views[k] = (off, n)
    off += n + GUARD
total = off
SENT = 12345.0
host = np.full(total, SENT, dtype=np.float32)
rng = np.random.default_rng(0)
def put(k, a):
    o, n = views[k]
    host[o:o + n] = a.reshape(-1)
put("xw", rng.standard_normal(N * XW).astype(np.float32)); put("norm", rng.random(E).astype(np.float32))
put("att", (rng.standard_normal(R * NB) * 0.3).astype(np.float32)); put("basis", (rng.standard_normal(NB * Fd * O) * 0.1).astype(np.float32))
put("root", (rng.standard_normal(Fd * O) * 0.1).astype(np.float32)); put("gout", (rng.standard_normal(N * O) * 0.1).astype(np.float32))
ref = None
for it in range(int(os.environ.get("ITERS", "6"))):
    arena = torch.from_numpy(host.copy()).to(DEV)
    def v(k, *shape):
        o, n = views[k]
        return arena[o:o + n].view(*shape)
    if it % 2:
        capi.poison_lds()
    capi.brgcn_fwd_tile(v("xw", N, XW), XW, Fd, O, N, g, v("norm", E), v("att", R, NB), NB, v("basis", NB, Fd, O), v("root", Fd, O),
                        v("Z", N, NB * Fd), v("sl", Sg, N, O))
    capi.brgcn_bwd_source_tile(v("gout", N, O), O, Fd, O, N, g, v("norm", E), v("att", R, NB), NB, v("basis", NB, Fd, O),
                               v("root", Fd, O), v("dsl", Sg, N, Fd))
    capi.brgcn_bwd_edges_tile(v("xw", N, XW), XW, Fd, O, N, R, g, v("norm", E), v("att", R, NB), NB, v("basis", NB, Fd, O),
                              v("gout", N, O), O, v("TT", E, NB), v("dn", Sg, E), E, v("datt", R, NB))
    torch.cuda.synchronize()
    out = arena.cpu().numpy()
    mask = np.ones(total, dtype=bool)
    for k, (o, n) in views.items():
        mask[o:o + n] = False
    hit = np.nonzero((out != SENT) & mask)[0]
    msg = []
    if hit.size:
        msg.append("GUARD hits %d first %s" % (hit.size, hit[:6].tolist()))
    for k in ("xw", "norm", "att", "basis", "root", "gout"):
        o, n = views[k]
        if not np.array_equal(out[o:o + n], host[o:o + n]):
            msg.append("input %s changed" % k)
    res = {k: out[views[k][0]:views[k][0] + views[k][1]].copy() for k in ("Z", "sl", "dsl", "TT", "dn", "datt")}
    for k, a in res.items():
        if (a == SENT).any():
            msg.append("%s has %d unwritten" % (k, int((a == SENT).sum())))
    if ref is None:
        ref = res
    else:
        for k in res:
            d = int((res[k] != ref[k]).sum())
            if d:
                msg.append("%s differs from run 0 in %d" % (k, d))
    print("iter", it, "; ".join(msg) if msg else "clean")
print("N", N, "E", E, "layout", views)
