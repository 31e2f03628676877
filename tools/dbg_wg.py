import sys; sys.path.insert(0, "/root/repo")
import torch, importlib
from bench import synthetic_batch
plugin = importlib.import_module("track_mm.cogmen")
params = plugin.ParamsType().from_args(["--dataset=iemocap-cogmen-sbert-6", "--modality=atv", "--compute=bf16"])
params.train.batch_size = 32
tr = plugin.COGMENTrainer(params, torch.device("cuda:0"))
b = tr.prepare_batch(synthetic_batch(params, 32, 110, seed=1))
tr.train_step(b)
ws = next(iter(tr.model._ws.values()))
ws["wgrad_counters"] = torch.zeros(8192, dtype=torch.int32, device="cuda:0")
for _ in range(5):
    tr.train_step(b)
torch.cuda.synchronize()
t = ws["wgrad_counters"][2048:2048 + 5].cpu().tolist()
names = ["start", "idx staged", "K loop done", "tile reduced + slab stored", "arrival known"]
for i in range(1, 5):
    print("%-28s +%.2f us" % (names[i], ((t[i] - t[i - 1]) & 0xffffffff) / 2270.0))
print("items", ws["wgrad_items"])
