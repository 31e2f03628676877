#!/usr/bin/env python3
"""erc_gemm_x3 against erc_gemm_f32 on the two big products of MMGCN's GCNII chain (R3 = 3180 rows, 200 x 12800)."""
import os, sys
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
import torch
from erc_amd import capi


def t_us(fn, reps=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / reps


dev = "cuda:0"
R3, FD, L = 3180, 200, 12800
H0 = torch.randn(R3, FD, device=dev)
U = torch.randn(FD, L, device=dev) * 0.1          # [k][n] as gcnii_chain_prep writes it
UT = U.t().contiguous()                           # [n][k]
Call = torch.zeros(R3, L, device=dev)
Call3 = torch.zeros(R3, L, device=dev)
print("Call = H0 U      f32: %.1f us" % t_us(lambda: capi.gemm_f32(H0, FD, 0, None, U, L, 1, None, Call, L, R3, L, FD)))
print("Call = H0 UT^T    x3: %.1f us" % t_us(lambda: capi.gemm_x3(H0, FD, UT, FD, Call3, L, R3, L, FD)))
print("   max |diff| %.3e of max %.3e" % (float((Call - Call3).abs().max()), float(Call.abs().max())))
DG = torch.randn(R3, L, device=dev)
d1 = torch.zeros(R3, FD, device=dev)
from erc_amd.engine import GemmPlanner, linear_fwd
pl = GemmPlanner(dev, 64 << 20)
print("dH0 = DG U^T     f32: %.1f us" % t_us(lambda: (pl.reset(), linear_fwd(pl, DG, L, None, U, None, d1, FD, R3, FD, L))))
for S in (5, 10, 16, 20):
    slabs = torch.zeros(S * R3 * FD, device=dev)
    d3 = torch.zeros(R3, FD, device=dev)

    def run():
        capi.gemm_x3(DG, L, U, L, slabs, FD, R3, FD, L, split_k=S, c_slab=R3 * FD)
        if S > 1:
            capi.slab_reduce(slabs, S, R3 * FD, None, FD, 0, d3, R3 * FD)
    print("dH0 x3 split %2d: %.1f us" % (S, t_us(run)), "  max |diff| %.3e of %.3e" % (float(((slabs[:R3 * FD].view(R3, FD) if S == 1 else d3) - d1).abs().max()), float(d1.abs().max())))
