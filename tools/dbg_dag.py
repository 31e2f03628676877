import sys, os; sys.path.insert(0, "/root/repo")
import torch, importlib
from bench import synthetic_batch
import erc_amd.dagerc as D
plugin = importlib.import_module("track_mm.dagerc")
params = plugin.ParamsType().from_args(["--dataset=iemocap-cogmen-6", "--modality=atv", "--reimplement"])
tr = plugin.DAGERCTrainer(params, torch.device("cuda:0"))
b = tr.prepare_batch(synthetic_batch(params, 16, 110, seed=3))
tr.model.train()
# enlarge cl_state for the timestamps
tr.model.loss_and_grads(b)
ws = next(iter(tr.model._ws.values()))
pass
for _ in range(3):
    tr.model.loss_and_grads(b)
torch.cuda.synchronize()
t = ws["cl_state"][1:1 + 7].cpu().tolist()
names = ["start", "1 loads+matvec_t Wr", "1 exchange+sum", "2 pointwise", "3 matvec_t 1800", "3 exchange+sum", "4 attention bwd + updates"]
for i in range(1, 7):
    d = (t[i] - t[i - 1]) & 0xffffffff
    print("%-18s +%6d ticks = %.2f us" % (names[i], d, d / 2270.0))
print("total", ((t[6] - t[0]) & 0xffffffff) / 2270.0, "us")
