"""Time the encoder's row kernels in isolation (HIP-graph replay of 20 launches)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from erc_amd import capi
dev = "cuda:0"
M, D = 3520, 1380
def timeit(f, name, nbytes):
    for _ in range(3): f()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(20): f()
    g.replay(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); g.replay(); e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1e3 / 20
    print("%-40s %.1f us  %.0f GB/s" % (name, us, nbytes / us * 1e-3), flush=True)
a, b = torch.randn(M, D, device=dev), torch.randn(M, D, device=dev)
gam, bet = torch.rand(D, device=dev) + 0.5, torch.randn(D, device=dev)
yf, yh = torch.zeros(M, D, device=dev), torch.zeros(M, D, dtype=torch.bfloat16, device=dev)
ss, st = torch.zeros(M, D, device=dev), torch.zeros(2 * M, device=dev)
rng = torch.tensor([3, 11], dtype=torch.int64, device=dev)
for p in (0.0, 0.5):
    timeit(lambda: capi.enc_add_layernorm_train(a, b, D, M, gam, bet, 1e-5, p, rng if p else None, 1, yf, yh, ss, st), "ln fwd p=%.1f" % p, M * D * 18)
nb = capi.enc_layernorm_bwd_blocks(M)
ds, db = torch.zeros(M, D, device=dev), torch.zeros(M, D, dtype=torch.bfloat16, device=dev)
part = torch.zeros(nb, 2 * D, device=dev)
inv = torch.arange(M, dtype=torch.int32, device=dev)
for p in (0.0, 0.5):
    timeit(lambda: capi.enc_layernorm_bwd(a, None, b, ss, st, gam, D, M, p, rng if p else None, 1, ds, db, part), "ln bwd dy_b p=%.1f" % p, M * D * 18)
    timeit(lambda: capi.enc_layernorm_bwd(a, inv, None, ss, st, gam, D, M, p, rng if p else None, 1, ds, db, part), "ln bwd map p=%.1f" % p, M * D * 14)
B, T, heads = 32, 110, 6
qkv = (torch.randn(B * T, 3 * D, device=dev) * 0.5).to(torch.bfloat16)
dout = torch.randn(B * T, D, device=dev).to(torch.bfloat16)
out = torch.zeros(B * T, D, dtype=torch.bfloat16, device=dev)
dqkv = torch.zeros(B * T, 3 * D, dtype=torch.bfloat16, device=dev)
lens = torch.randint(20, 111, (B,), dtype=torch.int64, device=dev)
fl = 4.0 * B * T * T * D
for p in (0.0, 0.5):
    timeit(lambda: capi.enc_attention_train(qkv, B, T, D, heads, lens, p, rng if p else None, 2, out), "attention fwd p=%.1f" % p, 0)
    timeit(lambda: capi.enc_attention_bwd(qkv, dout, B, T, D, heads, lens, p, rng if p else None, 2, dqkv), "attention bwd p=%.1f" % p, 0)
