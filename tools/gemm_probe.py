"""Time erc_enc_gemm_bf16 on the encoder's shapes (HIP-graph replay of 20 launches): us and TFLOP/s per shape."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from erc_amd import capi
shapes = [(3520, 1380, 1380), (3520, 4140, 1380), (3520, 2048, 1380), (3520, 1380, 2048), (1380, 2048, 3520), (4140, 1380, 3520)]
if len(sys.argv) > 1:
    shapes = [tuple(int(v) for v in a.split("x")) for a in sys.argv[1:]]
dev = "cuda:0"
for M, N, K in shapes:
    a = torch.randn(M, K, device=dev).to(torch.bfloat16)
    w = torch.randn(N, K, device=dev).to(torch.bfloat16)
    c = torch.zeros(M, N, device=dev)
    f = lambda: capi.enc_gemm_bf16(a, K, w, K, None, c, None, N, M, N, K)
    for _ in range(3):
        f()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(20):
            f()
    g.replay(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); g.replay(); e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1e3 / 20
    print("M=%d N=%d K=%d  %.1f us  %.0f TFLOP/s" % (M, N, K, us, 2.0 * M * N * K / us * 1e-6), flush=True)
    # the vendor library on the same shape (torch.matmul -> hipBLASLt), for reference only
    import torch.nn.functional as F
    c2 = torch.empty(M, N, device=dev, dtype=torch.bfloat16)
    f2 = lambda: torch.matmul(a, w.t(), out=c2)
    for _ in range(3):
        f2()
    torch.cuda.synchronize()
    g2 = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g2):
        for _ in range(20):
            f2()
    g2.replay(); torch.cuda.synchronize()
    e0.record(); g2.replay(); e1.record(); torch.cuda.synchronize()
    us2 = e0.elapsed_time(e1) * 1e3 / 20
    print("      hipBLASLt via torch.matmul: %.1f us  %.0f TFLOP/s" % (us2, 2.0 * M * N * K / us2 * 1e-6), flush=True)
