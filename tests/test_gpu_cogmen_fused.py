"""Stage-by-stage check of the fused COGMEN graph kernels (csrc/cogmen_fused.hip, bf16 compute mode) against the
unfused fp32 kernels of the parity path (graph_ops.hip, head.hip -- themselves checked against the oracle) plus
float64 torch products on the same bf16-rounded operands.  Pins every intermediate the two launches write."""
import math

import pytest
import torch

from tests.util_cases import cogmen_case, to_device

pytestmark = pytest.mark.gpu
F, R = 100, 8


def rb(t):
    return t.to(torch.bfloat16).to(torch.float32)


def bf16_ulp(ref):
    """spacing of bf16 numbers around |ref| (8 significant bits)"""
    return torch.clamp(ref.abs(), min=1e-30).log2().floor().exp2() * 2.0 ** -7


@pytest.mark.parametrize("case", [
    dict(B=9, min_len=1, max_len=30, seed=5),
    dict(B=32, min_len=20, max_len=110, seed=16),     # the benched shape's node count (~2000)
    dict(B=3, min_len=1, max_len=2, seed=7),
], ids=["ragged", "config2", "tiny"])
@pytest.mark.parametrize("n_spk", [2, 3], ids=["two-speakers", "edge-gather"])
def test_fused_kernels_stagewise(case, n_spk):
    from erc_amd import capi
    from erc_amd.cogmen import COGMENModule, WP, WF
    dev = "cuda:0"
    c = cogmen_case(dims=dict(a=12, t=20, v=16), **case)
    torch.manual_seed(case["seed"])
    m = COGMENModule(c["D"], 100, 17, 2, 6, compute="bf16").finalize(dev)
    with torch.no_grad():
        m.gcn.conv1.bias.uniform_(-0.1, 0.1)
        m.gcn.bn.weight.uniform_(0.5, 1.5)
    m.refresh_shadows()
    b = to_device(c["batch"], dev)
    B, T = b["input_tensor"].shape[:2]
    N = int(b["label"].shape[0])
    ws = m._workspace(B, T, N, dev)
    g, fp = ws["g"], m.flat
    spk = b["speaker_tensor"]
    capi.window_graph_build(b["text_length"], spk, spk.stride(0), spk.stride(1), B, T, WP, WF, 2, N, ws["E"], g)
    E = int(g["in_ptr"][N])
    f32 = lambda *s: torch.zeros(*s, dtype=torch.float32, device=dev)
    H0 = torch.randn(N, F, device=dev)
    scale = 1.0 / math.sqrt(F)
    bn = m.gcn.bn
    rm0, rv0 = bn.running_mean.clone(), bn.running_var.clone()
    capi.poison_lds()
    capi.cogmen_fwd_tile(H0, F, N, WP, WF, g, m._sh["catT"], fp.w("gcn.conv1.bias"), m._sh["q"],
                         fp.w("gcn.conv2.lin_query.bias"), scale, ws["Mb"], ws["Mb"].shape[1], ws["inv_cnt"], ws["H1b"], ws["H1b"].shape[1], ws["QKVS"],
                         ws["H2"], F, ws["alpha"], bn_fused=True, running_mean=bn.running_mean, running_var=bn.running_var,
                         momentum=bn.momentum, eps=bn.eps, saved=ws["bn_saved"], bn_ws=ws["bn_tile_ws"], n_speakers=n_spk)
    # ---- relation means: the fp32 kernel's result rounded to bf16.  The fused kernel divides by rcp + one Newton step and,
    #      with two speakers, takes window sums as differences of per-speaker prefix sums (n_spk = 3 only selects the
    #      edge-by-edge gather: the graph below is built with two speakers either way): fp32 last-bit differences, which can
    #      only show where the mean sits on a bf16 rounding boundary
    M_ref, inv_ref = f32(N, 9 * F), f32(N, R)
    capi.rgcn_mean_fwd(H0, F, F, R, N, g, M_ref, 9 * F, inv_ref)
    Mb = ws["Mb"][:, :900].float()
    assert bool(((Mb - M_ref).abs() <= bf16_ulp(M_ref) * 0.51 + 1e-6).all())
    assert float((Mb != rb(M_ref)).float().mean()) < 2e-3
    assert torch.equal(ws["inv_cnt"], inv_ref)
    # ---- H1 = rb(M) rb(Wcat) + b: within one bf16 step of the float64 product
    Wcat = torch.cat([fp.w("gcn.conv1.weight").reshape(R * F, F), fp.w("gcn.conv1.root")], 0)
    H1_ref = (Mb.double() @ rb(Wcat).double() + fp.w("gcn.conv1.bias").double()).float()
    H1b = ws["H1b"][:, :F].float()
    assert bool(((H1b - H1_ref).abs() <= bf16_ulp(H1_ref) + 1e-6).all())
    assert float((H1b != rb(H1_ref)).float().mean()) < 0.01        # only boundary cases may round the other way
    # ---- QKVS from the stored H1 operand
    Wq = torch.cat([fp.w("gcn.conv2.lin_%s.weight" % n) for n in ("query", "key", "value", "skip")], 0)
    bq = torch.cat([fp.w("gcn.conv2.lin_%s.bias" % n) for n in ("query", "key", "value", "skip")], 0)
    Q_ref = (H1b.double() @ rb(Wq).double().t() + bq.double()).float()
    assert float((ws["QKVS"] - Q_ref).abs().max()) < 2e-5 * max(1.0, float(Q_ref.abs().max()))
    # ---- attention on the kernel's own QKVS == the unfused attention kernel
    H2_ref, al_ref = f32(N, F), f32(ws["E"])
    capi.tconv_attn_fwd(ws["QKVS"], 4 * F, F, N, scale, g, H2_ref, F, al_ref)
    assert float((ws["H2"] - H2_ref).abs().max()) < 1e-5
    assert float((ws["alpha"][:E] - al_ref[:E]).abs().max()) < 1e-6
    # ---- BatchNorm statistics
    mean, var = ws["H2"].double().mean(0), ws["H2"].double().var(0, unbiased=False)
    assert float((ws["bn_saved"][:F].double() - mean).abs().max()) < 1e-6
    assert float((ws["bn_saved"][F:].double() - 1.0 / torch.sqrt(var + bn.eps)).abs().max()) < 1e-4
    if N > 1:
        assert float((bn.running_mean.double() - (0.9 * rm0.double() + 0.1 * mean)).abs().max()) < 1e-6
        assert float((bn.running_var.double() - (0.9 * rv0.double() + 0.1 * var * N / (N - 1))).abs().max()) < 1e-5

    # ================================================================ backward
    dY = torch.randn(N, F, device=dev) * 0.1
    bn_bwd = torch.randn(2 * F, device=dev) * 0.01
    gamma = fp.w("gcn.bn.weight")
    capi.poison_lds()
    capi.cogmen_bwd_tile(dY, ws["H2"], F, N, WP, WF, gamma, ws["bn_saved"], bn_bwd, ws["QKVS"], ws["alpha"], g, ws["inv_cnt"],
                         m._sh["qT"], m._sh["wb"], scale, ws["dQKVS"], ws["dH1"], ws["dH0"], F, n_speakers=n_spk)
    dH2_ref, dQ_ref, dsc = f32(N, F), f32(N, 4 * F), f32(ws["E"])
    capi.tconv_attn_bwd(ws["QKVS"], 4 * F, F, N, scale, g, ws["alpha"], dY, F, dQ_ref, dsc,
                        bn=(ws["H2"], F, gamma, ws["bn_saved"], bn_bwd, dH2_ref))
    s = max(1.0, float(dQ_ref.abs().max()))
    assert float((ws["dQKVS"] - dQ_ref).abs().max()) < 1e-5 * s, float((ws["dQKVS"] - dQ_ref).abs().max())
    dH1_ref = (rb(ws["dQKVS"]).double() @ rb(Wq).double()).float()
    assert float((ws["dH1"] - dH1_ref).abs().max()) < 2e-5 * max(1.0, float(dH1_ref.abs().max()))
    # dH0 = sum_r rb(dP_r) rb(W_r)^T, dP_r through the unfused relation-mean backward one block at a time
    dH0_ref = torch.zeros(N, F, dtype=torch.float64, device=dev)
    for r in range(R + 1):
        dM = f32(N, 9 * F)
        dM[:, r * F:(r + 1) * F] = ws["dH1"]
        dP = f32(N, F)
        capi.rgcn_mean_bwd(dM, 9 * F, F, R, N, g, ws["inv_cnt"], dP, F)
        dH0_ref += rb(dP).double() @ rb(Wcat[r * F:(r + 1) * F]).double().t()
    # (an fp32 difference in a dP entry that sits on a bf16 rounding boundary moves that operand by one bf16 step)
    d = (ws["dH0"].double() - dH0_ref).abs()
    sc = max(1e-3, float(dH0_ref.abs().max()))
    assert float(d.max()) < 5e-3 * sc and float(d.mean()) < 1e-4 * sc, (float(d.max()), float(d.mean()), sc)


@pytest.mark.parametrize("case", [
    dict(B=9, min_len=1, max_len=30, seed=5, dims=dict(a=12, t=20, v=16)),
    dict(B=32, min_len=20, max_len=110, seed=16, dims=dict(a=100, t=768, v=512)),   # the benched shape: D = 1380
    dict(B=3, min_len=1, max_len=2, seed=7, dims=dict(a=12, t=20, v=16)),
    dict(B=70, min_len=1, max_len=9, seed=8, dims=dict(a=100, t=100, v=512)),       # more dialogues than a wavefront; D = 712
], ids=["ragged", "config2", "tiny", "many-dialogues"])
@pytest.mark.parametrize("S,wp,wf", [(2, 5, 5), (9, 10, 10), (3, 2, 7), (2, -1, 3)], ids=["cogmen", "meld", "asymmetric", "unbounded-past"])
def test_project_graph_equals_graph_build_plus_projection(case, S, wp, wf):
    """csrc/cogmen_project.hip against the two launches it replaces: every array of erc_window_graph_build bit-equal
    (node_off, node_row, node_spk, both CSRs, out_eid, counts -- edge indices and relation ids are integer work), H0
    bit-equal to erc_gemm_bf16a_stream through the node -> row gather (same fragments, same summation order)."""
    from erc_amd import capi
    dev = "cuda:0"
    c = cogmen_case(n_speakers=S, **case)
    b = to_device(c["batch"], dev)
    x = b["input_tensor"].to(torch.bfloat16)
    B, T, D = x.shape
    N = int(b["label"].shape[0])
    spk, lens = b["speaker_tensor"], b["text_length"]
    w = (wp if wp >= 0 else T) + (wf if wf >= 0 else T) + 1
    E = max(1, N * min(w, T))
    mk = lambda: dict({k: torch.full((n,), -7, dtype=torch.int32, device=dev) for k, n in
                       dict(node_off=B + 1, node_row=N, node_spk=N, in_ptr=N + 1, in_src=E, in_typ=E, out_ptr=N + 1, out_dst=E,
                            out_typ=E, out_eid=E, counts=2).items()})
    g_ref, g_new = mk(), mk()
    capi.window_graph_build(lens, spk, spk.stride(0), spk.stride(1), B, T, wp, wf, S, N, E, g_ref)
    torch.manual_seed(3)
    W = (torch.randn(F, D, device=dev) / math.sqrt(D)).to(torch.bfloat16)
    bias = torch.randn(F, device=dev)
    H0_ref = torch.full((N, F), float("nan"), device=dev)
    H0_new = torch.full((N, F), float("nan"), device=dev)
    capi.gemm_bf16a_stream(x, D, g_ref["node_row"], W, D, H0_ref, F, N, F, D, bias=bias)
    assert capi.cogmen_project_graph_ok(D, F, B, D, D)
    capi.poison_lds()
    capi.cogmen_project_graph(x, D, W, D, bias, H0_new, F, F, D, lens, spk, B, T, wp, wf, S, N, E, g_new)
    torch.cuda.synchronize()
    n_e = int(g_ref["counts"][1])
    assert int(g_new["counts"][0]) == N and int(g_new["counts"][1]) == n_e
    for k in g_ref:
        n_live = n_e if k in ("in_src", "in_typ", "out_dst", "out_typ", "out_eid") else g_ref[k].numel()
        assert torch.equal(g_new[k][:n_live].cpu(), g_ref[k][:n_live].cpu()), k
        assert bool((g_new[k][n_live:] == -7).all()), k          # nothing written past the live entries
    if D >= 1024:   # the reference launch took the same persistent kernel: same fragments, same order -> same bits
        assert torch.equal(H0_new.cpu(), H0_ref.cpu())
    else:           # small shapes: the reference launch is the streaming kernel (other K split): fp32 summation order differs
        assert float((H0_new - H0_ref).abs().max()) < 2e-5
    assert not bool(torch.isnan(H0_new).any())
