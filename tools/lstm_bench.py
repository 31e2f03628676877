"""Time the BiLSTM recurrence kernels alone (csrc/lstm.hip): python tools/lstm_bench.py [B T packed]"""
import sys
import torch
sys.path.insert(0, ".")
from erc_amd import capi

H = 100


def run(B, T, packed, reps=50):
    dev = "cuda:0"
    torch.manual_seed(0)
    rows = B * T
    lens = torch.randint(max(1, T // 2), T + 1, (B,), device=dev, dtype=torch.int64) if packed else None
    GX = torch.randn(rows, 8 * H, device=dev) * 0.5
    Whh = torch.randn(2, 4 * H, H, device=dev) * 0.1
    bhh = torch.randn(2, 4 * H, device=dev) * 0.1
    z = lambda *s: torch.zeros(*s, device=dev)
    Hout, gates, Cst, Hprev, dGX = z(rows, 2 * H), z(rows, 8 * H), z(rows, 2 * H), z(rows, 2 * H), z(rows, 8 * H)
    dH = torch.randn(rows, 2 * H, device=dev) * 0.1
    sb, st = (T, 1) if packed else (1, B)
    fwd = lambda: capi.lstm_scan_fwd(GX, 8 * H, Whh, bhh, lens, None, sb, st, B, T, Hout, 2 * H, None, 0, 0.0, None, 0,
                                     gates, Cst, Hprev)
    bwd = lambda: capi.lstm_scan_bwd(Whh, lens, None, sb, st, B, T, gates, Cst, dH, 2 * H, 0.0, None, 0, dGX)
    out = {}
    for name, fn in (("fwd", fwd), ("bwd", bwd)):
        for _ in range(5):
            fn()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            fn()
        e1.record()
        torch.cuda.synchronize()
        out[name] = e0.elapsed_time(e1) / reps * 1e3
    stamps = torch.zeros(8, dtype=torch.int64, device=dev)
    capi.lstm_set_stamps(stamps)
    fwd()
    torch.cuda.synchronize()
    capi.lstm_set_stamps(None)
    t = stamps.cpu().tolist()
    print("  fwd step stamps (cycles): products %d | reduce + activation %d | cell %d | barrier %d | step %d" % (
        t[1] - t[0], t[2] - t[1], t[3] - t[2], t[4] - t[3], t[4] - t[0]))
    print("  scan loop of workgroup (0,0): %d steps, %d cycles in %.2f us -> %.2f GHz, %.0f cycles/step" % (
        t[7], t[5], t[6] / 100.0, t[5] / (t[6] * 10.0), t[5] / max(t[7], 1)))
    print("B=%d T=%d %s: fwd %.1f us (%.2f us/step)  bwd %.1f us (%.2f us/step)" % (
        B, T, "packed" if packed else "unpacked", out["fwd"], out["fwd"] / T, out["bwd"], out["bwd"] / T))


if __name__ == "__main__":
    if len(sys.argv) > 1:
        run(int(sys.argv[1]), int(sys.argv[2]), bool(int(sys.argv[3])))
    else:
        run(32, 33, True)
        run(32, 110, False)
