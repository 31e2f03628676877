import sys
sys.path.insert(0,'/root/repo')
import torch, importlib
from erc_amd import capi
import bench
plugin = importlib.import_module("track_mm.dgcn")
params = plugin.ParamsType().from_args(["--dataset=meld-mmgcn-7", "--modality=atv", "--compute=bf16", "--loss_weights=False"])
params.train.batch_size = 32
tr = plugin.DGCNTrainer(params, torch.device("cuda:0"))
hb = bench.synthetic_batch(params, 32, 33, seed=1)
b = tr.prepare_batch(hb)
for _ in range(2): tr.train_step(b)
capi.start_recording()
tr.train_step(b)
rec = capi.stop_recording()
for i,(n,a) in enumerate(rec): print(i, n)
