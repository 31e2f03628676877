import sys; sys.path.insert(0, "/root/repo")
import torch
from tests.util_cases import cogmen_case, run_cogmen_parity
import tests.util_cases as U
orig = U.rel_err
res = run_cogmen_parity(cogmen_case(B=32, min_len=20, max_len=110, dims=dict(a=100, t=768, v=512), seed=16), compute="bf16")
print("logit", res["logit_err"], "loss", res["loss_err"], "norm", res["grad_norm_err"])
for k, v in sorted(res["grad_errs"].items(), key=lambda kv: -kv[1]): print("%-32s %.4f" % (k, v))
