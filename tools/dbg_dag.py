import sys, os; sys.path.insert(0, "/root/repo")
import torch, importlib, time
from bench import synthetic_batch
plugin = importlib.import_module("track_mm.dagerc")
outs = {}
for P in (1, 8):
    os.environ["ERC_DAG_CLUSTER"] = str(P)
    params = plugin.ParamsType().from_args(["--dataset=iemocap-cogmen-6", "--modality=atv", "--reimplement"])
    torch.manual_seed(0)
    tr = plugin.DAGERCTrainer(params, torch.device("cuda:0"))
    tr.model.train(); tr.model.drop_p = 0.0 if hasattr(tr.model, "drop_p") else None
    b = tr.prepare_batch(synthetic_batch(params, 5, 23, seed=3))
    stats = tr.model.loss_and_grads(b)
    torch.cuda.synchronize()
    ws = next(iter(tr.model._ws.values()))
    print("P", P, "cluster", ws["cluster"], "err flag", int(ws["cl_state"][0]), "loss", float(stats[0]))
    outs[P] = tr.model.flat.grad.clone()
d = (outs[1] - outs[8]).abs()
print("max |dgrad|", float(d.max()), "ref max", float(outs[1].abs().max()), "nan", bool(torch.isnan(outs[8]).any()))
