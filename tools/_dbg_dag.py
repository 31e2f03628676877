import os, sys, torch
sys.path.insert(0, "/root/repo")
import numpy as np
from tests.util_cases import fill_params, to_device
from erc_amd.dagerc import DAGERCModule
fx = np.load("/root/repo/tests/golden/dagerc_small.npz", allow_pickle=False)
batch = {k[3:]: torch.from_numpy(fx[k]) for k in fx.files if k.startswith("in_")}
D, C = int(fx["dims"].sum()), int(fx["n_classes"])
model = DAGERCModule(emb_dim=D, dropout=0.0, n_classes=C, gnn_layers=4)
fill_params(model, int(fx["param_seed"]))
model.finalize("cuda:0")
model.train()
model.loss_and_grads(to_device(batch, "cuda:0"))
ws = model._last_ws
out = {"dHall": ws["dHall"].cpu(), "grad": model.flat.grad.cpu()}
for k in ("DGI", "DGH", "dM", "dks", "A"):
    for l in range(4):
        out["%s%d" % (k, l)] = ws[k][l].cpu()
print("cfg", ws["cfg"], "err", int(model.rec_state[0]))
torch.save(out, sys.argv[1])
