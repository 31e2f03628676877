"""GPU parity tests of the individual libercgraft operators against the CPU oracle
(oracle/*.py, plain torch fp32/fp64 on the host).  Every call goes through the C-ABI."""
import math

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

DEV = "cuda:0"


@pytest.fixture(scope="module")
def capi():
    from erc_amd import capi as c
    c.lib()
    return c


def _close(got, want, atol, rtol=1e-5):
    got, want = got.detach().cpu().double(), want.detach().cpu().double()
    err = (got - want).abs()
    tol = atol + rtol * want.abs()
    assert bool((err <= tol).all()), "max err %.3e (tol %.1e) at %s" % (
        float(err.max()), atol, np.unravel_index(int(err.argmax()), err.shape))


# ------------------------------------------------------------------ K1 graph
@pytest.mark.parametrize("name", ["graph_cogmen_s2_w5", "graph_dgcn_s9_w10", "graph_asym_s3_w2_4"])
def test_window_graph_golden(capi, golden, name):
    """bit-exact edges / relation ids vs the reference's batch_graphify (golden fixture)."""
    from erc_amd.cogmen import build_graph_tensors
    g = golden(name)
    lengths = torch.from_numpy(g["lengths"]).to(DEV)
    spk = torch.from_numpy(g["speakers"]).to(DEV)
    wp, wf, S = int(g["wp"]), int(g["wf"]), int(g["n_speakers"])
    gr, ei, et = build_graph_tensors(lengths, spk, wp, wf, S)
    N, E = gr["counts"].cpu().tolist()
    assert N == int(g["lengths"].sum()) and E == g["edge_index"].shape[1]
    np.testing.assert_array_equal(ei[:, :E].cpu().numpy(), g["edge_index"])
    np.testing.assert_array_equal(et[:E].cpu().numpy(), g["edge_type"])
    _check_csr(gr, g["edge_index"], g["edge_type"], N, E, g["lengths"], spk.shape[1])


def _check_csr(gr, ei, et, N, E, lengths, T):
    in_ptr = gr["in_ptr"][:N + 1].cpu().numpy()
    np.testing.assert_array_equal(gr["in_src"][:E].cpu().numpy(), ei[0])
    np.testing.assert_array_equal(gr["in_typ"][:E].cpu().numpy(), et)
    deg = np.bincount(ei[1], minlength=N)
    np.testing.assert_array_equal(np.diff(in_ptr), deg)
    # out-CSR: same edge set grouped by source, out_eid points back into the in-CSR
    out_ptr = gr["out_ptr"][:N + 1].cpu().numpy()
    out_dst = gr["out_dst"][:E].cpu().numpy()
    out_typ = gr["out_typ"][:E].cpu().numpy()
    out_eid = gr["out_eid"][:E].cpu().numpy()
    np.testing.assert_array_equal(np.diff(out_ptr), np.bincount(ei[0], minlength=N))
    src_of = np.repeat(np.arange(N), np.diff(out_ptr))
    np.testing.assert_array_equal(ei[0][out_eid], src_of)
    np.testing.assert_array_equal(ei[1][out_eid], out_dst)
    np.testing.assert_array_equal(et[out_eid], out_typ)
    assert len(np.unique(out_eid)) == E
    # node tables
    off = np.concatenate([[0], np.cumsum(lengths)])
    np.testing.assert_array_equal(gr["node_off"].cpu().numpy(), off)
    rows = np.concatenate([b * T + np.arange(L) for b, L in enumerate(lengths)]) if N else np.zeros(0)
    np.testing.assert_array_equal(gr["node_row"][:N].cpu().numpy(), rows)


@pytest.mark.parametrize("B,T,S,wp,wf", [(32, 110, 2, 5, 5), (64, 33, 9, 10, 10), (7, 40, 3, -1, 2), (5, 12, 2, 3, -1),
                                         (512, 110, 2, 5, 5)])
def test_window_graph_full_size(capi, B, T, S, wp, wf):
    """BASELINE-size batches: bit-exact vs the oracle's closed form (itself pinned to the reference)."""
    from erc_amd.cogmen import build_graph_tensors
    from oracle.graph import window_graph_closed_form
    rng = np.random.RandomState(B + T)
    lengths = rng.randint(1, T + 1, size=B)
    lengths[0] = T
    spk = rng.randint(0, S, size=(B, T))
    gr, ei, et = build_graph_tensors(torch.from_numpy(lengths).to(DEV), torch.from_numpy(spk).to(DEV), wp, wf, S)
    N, E = gr["counts"].cpu().tolist()
    ei_c, et_c = window_graph_closed_form(lengths, spk, wp, wf, S)
    assert N == lengths.sum() and E == ei_c.shape[1]
    np.testing.assert_array_equal(ei[:, :E].cpu().numpy(), ei_c)
    np.testing.assert_array_equal(et[:E].cpu().numpy(), et_c)
    _check_csr(gr, ei_c, et_c, N, E, lengths, T)


def test_window_graph_time_major_speakers(capi):
    """[T,B] speaker layout (MMGCN batch_first=False) through the stride arguments."""
    from erc_amd.cogmen import build_graph_tensors
    from oracle.graph import window_graph_closed_form
    rng = np.random.RandomState(0)
    lengths = np.array([4, 9, 1, 7])
    spk = rng.randint(0, 2, size=(4, 9))
    spk_tb = torch.from_numpy(spk).to(DEV).t().contiguous().t()  # [B,T] view over [T,B] storage
    assert spk_tb.stride() == (1, 4)
    gr, ei, et = build_graph_tensors(torch.from_numpy(lengths).to(DEV), spk_tb, 5, 5, 2)
    N, E = gr["counts"].cpu().tolist()
    ei_c, et_c = window_graph_closed_form(lengths, spk, 5, 5, 2)
    np.testing.assert_array_equal(ei[:, :E].cpu().numpy(), ei_c)
    np.testing.assert_array_equal(et[:E].cpu().numpy(), et_c)


# ------------------------------------------------------------------- K2 GEMM
def _gemm_ref(A, B):
    return (A.double() @ B.double()).float()


@pytest.mark.parametrize("M,N,K", [(70, 100, 1380), (2080, 100, 100), (33, 6, 100), (257, 400, 100), (1, 1, 1),
                                   (64, 32, 32), (130, 900, 100), (2080, 100, 900)])
@pytest.mark.parametrize("split", [1, 3])
def test_gemm_nt_nn(capi, M, N, K, split):
    g = torch.Generator().manual_seed(M * 7 + N)
    A = torch.randn(M, K, generator=g)
    W = torch.randn(N, K, generator=g)  # nn.Linear layout
    bias = torch.randn(N, generator=g)
    want = _gemm_ref(A, W.t())
    Ad, Wd = A.to(DEV), W.to(DEV)
    nchunk = -(-K // 32)
    S = min(split, nchunk)
    # NT
    slabs = torch.zeros(S, M, N, device=DEV)
    capi.gemm_f32(Ad, K, 0, None, Wd, K, 0, None, slabs, N, M, N, K, split_k=S, c_slab=M * N)
    _close(slabs.sum(0), want, 2e-4 * math.sqrt(K / 100))
    # NN with bias + relu epilogue (S == 1)
    Wk = W.t().contiguous().to(DEV)  # [K,N]
    out = torch.zeros(M, N, device=DEV)
    capi.gemm_f32(Ad, K, 0, None, Wk, N, 1, None, out, N, M, N, K, bias=bias.to(DEV), act=1)
    _close(out, torch.relu(want + bias), 2e-4 * math.sqrt(K / 100))
    # accumulate
    capi.gemm_f32(Ad, K, 0, None, Wk, N, 1, None, out, N, M, N, K, accumulate=1)
    _close(out, torch.relu(want + bias) + want, 4e-4 * math.sqrt(K / 100))


@pytest.mark.parametrize("rows,n_out,n_in", [(2080, 100, 1380), (61, 6, 100), (300, 400, 100), (5, 3, 7), (2080, 100, 100)])
def test_gemm_wgrad_tn(capi, rows, n_out, n_in):
    """dW = dY^T X (+ bias grad through the ones column), split-K slabs reduced by erc_slab_reduce."""
    g = torch.Generator().manual_seed(rows + n_out)
    dY = torch.randn(rows, n_out, generator=g)
    X = torch.randn(rows, n_in, generator=g)
    S = min(5, -(-rows // 32))
    slabs = torch.zeros(S, n_out, n_in, device=DEV)
    bslabs = torch.zeros(S, n_out, device=DEV)
    capi.gemm_f32(dY.to(DEV), n_out, 1, None, X.to(DEV), n_in, 1, None, slabs, n_in, n_out, n_in, rows,
                  split_k=S, c_slab=n_out * n_in, ones_col=1, bias_out=bslabs, bias_slab=n_out)
    out = torch.zeros(n_out, n_in, device=DEV)
    capi.slab_reduce(slabs, S, n_out * n_in, None, 0, 0, out, n_out * n_in)
    tol = 3e-4 * math.sqrt(rows / 100)
    _close(out, _gemm_ref(dY.t(), X), tol)
    _close(bslabs.sum(0), dY.double().sum(0).float(), tol)
    # [in,out]-stored weight (PyG): dW = X^T dY, bias grad through the ones ROW
    slabs2 = torch.zeros(S, n_in, n_out, device=DEV)
    bslabs2 = torch.zeros(S, n_out, device=DEV)
    capi.gemm_f32(X.to(DEV), n_in, 1, None, dY.to(DEV), n_out, 1, None, slabs2, n_out, n_in, n_out, rows,
                  split_k=S, c_slab=n_out * n_in, ones_col=2, bias_out=bslabs2, bias_slab=n_out)
    _close(slabs2.sum(0), _gemm_ref(X.t(), dY), tol)
    _close(bslabs2.sum(0), dY.double().sum(0).float(), tol)


def test_gemm_wide_tiles(capi):
    """problems big enough to select the 64x128 workgroup tile (NF = 8), all three operand layouts."""
    g = torch.Generator().manual_seed(12)
    M, N, K = 2100, 6000, 100
    A, W = torch.randn(M, K, generator=g), torch.randn(N, K, generator=g)
    out = torch.zeros(M, N, device=DEV)
    capi.gemm_f32(A.to(DEV), K, 0, None, W.to(DEV), K, 0, None, out, N, M, N, K)            # NT
    _close(out, _gemm_ref(A, W.t()), 3e-4)
    Wk = W.t().contiguous()
    capi.gemm_f32(A.to(DEV), K, 0, None, Wk.to(DEV), N, 1, None, out, N, M, N, K)           # NN
    _close(out, _gemm_ref(A, W.t()), 3e-4)
    At = torch.randn(300, 4200, generator=g)                                               # TN: K-major A [K=300, M=4200]
    Bk = torch.randn(300, 900, generator=g)
    o2 = torch.zeros(4200, 900, device=DEV)
    capi.gemm_f32(At.to(DEV), 4200, 1, None, Bk.to(DEV), 900, 1, None, o2, 900, 4200, 900, 300)
    _close(o2, _gemm_ref(At.t(), Bk), 5e-4)


def test_gemm_gather_and_unaligned(capi):
    """fused gather of valid rows of the padded block; leading dimension not a multiple of 4 (MELD D=1242)."""
    g = torch.Generator().manual_seed(9)
    for D in (1242, 1380, 37):
        Xpad = torch.randn(5 * 11, D, generator=g)
        rows = torch.tensor([0, 1, 2, 11, 12, 22, 23, 24, 25, 33, 44, 45, 54], dtype=torch.int32)
        W = torch.randn(100, D, generator=g)
        want = _gemm_ref(Xpad[rows.long()], W.t())
        out = torch.zeros(len(rows), 100, device=DEV)
        capi.gemm_f32(Xpad.to(DEV), D, 0, rows.to(DEV), W.to(DEV), D, 0, None, out, 100, len(rows), 100, D)
        _close(out, want, 2e-4 * math.sqrt(D / 100))
        dY = torch.randn(len(rows), 100, generator=g)
        dW = torch.zeros(100, D, device=DEV)
        capi.gemm_f32(dY.to(DEV), 100, 1, None, Xpad.to(DEV), D, 1, rows.to(DEV), dW, D, 100, D, len(rows))
        _close(dW, _gemm_ref(dY.t(), Xpad[rows.long()]), 2e-4)


def test_gemm_relu_dropout_epilogues(capi):
    g = torch.Generator().manual_seed(4)
    M, N, K = 300, 100, 100
    A, W = torch.randn(M, K, generator=g), torch.randn(N, K, generator=g)
    rng = torch.tensor([7, 123], dtype=torch.int64, device=DEV)
    Z = torch.zeros(M, N, device=DEV)
    capi.gemm_f32(A.to(DEV), K, 0, None, W.to(DEV), K, 0, None, Z, N, M, N, K, act=3, act_scale=2.0, drop_p=0.5,
                  rng_state=rng)
    Z2 = torch.zeros(M, N, device=DEV)
    capi.gemm_f32(A.to(DEV), K, 0, None, W.to(DEV), K, 0, None, Z2, N, M, N, K, act=3, act_scale=2.0, drop_p=0.5,
                  rng_state=rng)
    assert torch.equal(Z, Z2)  # same (offset, seed) -> same mask
    relu = torch.relu(_gemm_ref(A, W.t()))
    kept = Z.cpu() != 0
    _close(Z.cpu()[kept], 2.0 * relu[kept], 5e-4)
    frac = float(kept.float().sum() / (relu > 0).float().sum())
    assert 0.45 < frac < 0.55, frac
    rng2 = torch.tensor([8, 123], dtype=torch.int64, device=DEV)
    capi.gemm_f32(A.to(DEV), K, 0, None, W.to(DEV), K, 0, None, Z2, N, M, N, K, act=3, act_scale=2.0, drop_p=0.5,
                  rng_state=rng2)
    assert not torch.equal(Z, Z2)  # next step -> fresh mask
    # act 2: backward of relu+dropout
    dY = torch.randn(M, 6, generator=g)
    W3 = torch.randn(6, N, generator=g)
    dZ = torch.zeros(M, N, device=DEV)
    capi.gemm_f32(dY.to(DEV), 6, 0, None, W3.to(DEV), N, 1, None, dZ, N, M, N, 6, act=2, aux=Z, ldaux=N, act_scale=2.0)
    want = torch.where(Z.cpu() > 0, 2.0 * _gemm_ref(dY, W3), torch.zeros(()))
    _close(dZ, want, 1e-4)


@pytest.mark.parametrize("rows,D", [(2080, 1380), (77, 1242), (130, 712), (9, 40)])
def test_gemm_bf16_feature_operand(capi, rows, D):
    """bf16 feature block: forward projection and weight gradient vs fp32 math on the SAME bf16-rounded operands."""
    g = torch.Generator().manual_seed(rows)
    T = 7
    Xpad = torch.randn(rows + 20, D, generator=g).to(torch.bfloat16)
    idx = torch.sort(torch.randperm(rows + 20, generator=g)[:rows]).values.to(torch.int32)
    W = torch.randn(100, D, generator=g)
    Wb = W.to(torch.bfloat16).float()
    Xg = Xpad.float()[idx.long()]
    S = min(3, -(-D // 64))
    slabs = torch.zeros(S, rows, 100, device=DEV)
    capi.gemm_bf16x(Xpad.to(DEV), D, 0, idx.to(DEV), W.to(DEV), D, 0, None, 1, slabs, 100, rows, 100, D, split_k=S,
                    c_slab=rows * 100)
    _close(slabs.sum(0), _gemm_ref(Xg, Wb.t()), 3e-4 * math.sqrt(D / 100))
    dY = torch.randn(rows, 100, generator=g)
    dYb = dY.to(torch.bfloat16).float()
    S2 = min(4, -(-rows // 64))
    wsl = torch.zeros(S2, 100, D, device=DEV)
    bsl = torch.zeros(S2, 100, device=DEV)
    capi.gemm_bf16x(dY.to(DEV), 100, 1, None, Xpad.to(DEV), D, 1, idx.to(DEV), 0, wsl, D, 100, D, rows, split_k=S2,
                    c_slab=100 * D, ones_col=1, bias_out=bsl, bias_slab=100)
    _close(wsl.sum(0), _gemm_ref(dYb.t(), Xg), 3e-4 * math.sqrt(rows / 100))
    _close(bsl.sum(0), dYb.double().sum(0).float(), 3e-4 * math.sqrt(rows / 100))


# ------------------------------------------------------------- K3 / K4 graph ops
def _graph_case(seed, B=6, T=17, S=2, wp=5, wf=5):
    from erc_amd.cogmen import build_graph_tensors
    rng = np.random.RandomState(seed)
    lengths = rng.randint(1, T + 1, size=B)
    lengths[0] = T
    spk = rng.randint(0, S, size=(B, T))
    gr, ei, et = build_graph_tensors(torch.from_numpy(lengths).to(DEV), torch.from_numpy(spk).to(DEV), wp, wf, S)
    N, E = gr["counts"].cpu().tolist()
    return gr, ei[:, :E].cpu(), et[:E].cpu(), N, E


@pytest.mark.parametrize("S", [2, 3])
def test_rgcn_mean_fwd_bwd(capi, S):
    """relation-mean aggregation + GEMM == RGCNConv(mean) restatement; S=3 makes relation ids >= 8 that must be ignored."""
    from oracle.pyg import RGCNConvMean
    F, R = 100, 8
    gr, ei, et, N, E = _graph_case(11 + S, S=S)
    torch.manual_seed(0)
    conv = RGCNConvMean(F, F, R)
    with torch.no_grad():
        conv.bias.uniform_(-0.2, 0.2)
    x = torch.randn(N, F, requires_grad=True)
    want = conv(x, ei, et)
    gout = torch.randn(N, F)
    want.backward(gout)
    xd = x.detach().to(DEV)
    M = torch.zeros(N, (R + 1) * F, device=DEV)
    inv = torch.zeros(N, R, device=DEV)
    capi.rgcn_mean_fwd(xd, F, F, R, N, gr, M, (R + 1) * F, inv)
    Wcat = torch.cat([conv.weight.detach().reshape(R * F, F), conv.root.detach()], 0).to(DEV)
    out = torch.zeros(N, F, device=DEV)
    capi.gemm_f32(M, (R + 1) * F, 0, None, Wcat, F, 1, None, out, F, N, F, (R + 1) * F, bias=conv.bias.detach().to(DEV))
    _close(out, want, 1e-4)
    # backward: dM = gout Wcat^T ; dx = gather over out-edges
    dM = torch.zeros(N, (R + 1) * F, device=DEV)
    capi.gemm_f32(gout.to(DEV), F, 0, None, Wcat, F, 0, None, dM, (R + 1) * F, N, (R + 1) * F, F)
    dx = torch.zeros(N, F, device=DEV)
    capi.rgcn_mean_bwd(dM, (R + 1) * F, F, R, N, gr, inv, dx, F)
    _close(dx, x.grad, 1e-4)
    dW = torch.zeros((R + 1) * F, F, device=DEV)
    db = torch.zeros(F, device=DEV)
    capi.gemm_f32(M, (R + 1) * F, 1, None, gout.to(DEV), F, 1, None, dW, F, (R + 1) * F, F, N, ones_col=2, bias_out=db)
    _close(dW[:R * F].reshape(R, F, F), conv.weight.grad, 1e-4)
    _close(dW[R * F:], conv.root.grad, 1e-4)
    _close(db, conv.bias.grad, 1e-4)


def test_tconv_attention_fwd_bwd(capi):
    from oracle.pyg import TransformerConv1
    F = 100
    gr, ei, et, N, E = _graph_case(5)
    torch.manual_seed(1)
    conv = TransformerConv1(F, F)
    x = torch.randn(N, F, requires_grad=True)
    want = conv(x, ei)
    gout = torch.randn(N, F)
    want.backward(gout)
    Wq = torch.cat([conv.lin_query.weight, conv.lin_key.weight, conv.lin_value.weight, conv.lin_skip.weight], 0).detach()
    bq = torch.cat([conv.lin_query.bias, conv.lin_key.bias, conv.lin_value.bias, conv.lin_skip.bias], 0).detach()
    xd = x.detach().to(DEV)
    qkvs = torch.zeros(N, 4 * F, device=DEV)
    capi.gemm_f32(xd, F, 0, None, Wq.to(DEV), F, 0, None, qkvs, 4 * F, N, 4 * F, F, bias=bq.to(DEV))
    out = torch.zeros(N, F, device=DEV)
    alpha = torch.zeros(E, device=DEV)
    capi.tconv_attn_fwd(qkvs, 4 * F, F, N, 0.1, gr, out, F, alpha)
    _close(out, want, 1e-4)
    # alpha sums to one per target
    sums = torch.zeros(N).index_add_(0, ei[1], alpha.cpu())
    _close(sums, torch.ones(N), 1e-5)
    dqkvs = torch.zeros(N, 4 * F, device=DEV)
    dscore = torch.zeros(E, device=DEV)
    capi.tconv_attn_bwd(qkvs, 4 * F, F, N, 0.1, gr, alpha, gout.to(DEV), F, dqkvs, dscore)
    dx = torch.zeros(N, F, device=DEV)
    capi.gemm_f32(dqkvs, 4 * F, 0, None, Wq.to(DEV), F, 1, None, dx, F, N, F, 4 * F)
    _close(dx, x.grad, 2e-4)
    dW = torch.zeros(4 * F, F, device=DEV)
    db = torch.zeros(4 * F, device=DEV)
    capi.gemm_f32(dqkvs, 4 * F, 1, None, xd, F, 1, None, dW, F, 4 * F, F, N, ones_col=1, bias_out=db)
    want_dW = torch.cat([conv.lin_query.weight.grad, conv.lin_key.weight.grad, conv.lin_value.weight.grad,
                         conv.lin_skip.weight.grad], 0)
    want_db = torch.cat([conv.lin_query.bias.grad, conv.lin_key.bias.grad, conv.lin_value.bias.grad,
                         conv.lin_skip.bias.grad], 0)
    _close(dW, want_dW, 2e-4)
    _close(db, want_db, 2e-4)


# ------------------------------------------------------------ K5 / S3 / S4
@pytest.mark.parametrize("N", [1, 2, 37, 2080])
def test_batchnorm_lrelu(capi, N):
    F = 100
    torch.manual_seed(N)
    bn = torch.nn.BatchNorm1d(F)
    with torch.no_grad():
        bn.weight.uniform_(0.5, 1.5), bn.bias.uniform_(-0.5, 0.5)
        bn.running_mean.uniform_(-1, 1), bn.running_var.uniform_(0.5, 2)
    act = torch.nn.LeakyReLU()
    x = (torch.randn(N, F) * 2 + 0.7).requires_grad_()
    rm, rv = bn.running_mean.clone().to(DEV), bn.running_var.clone().to(DEV)
    ws = torch.zeros(capi.bn_ws_floats(F), device=DEV)
    saved = torch.zeros(2 * F, device=DEV)
    y = torch.zeros(N, F, device=DEV)
    gam, bet = bn.weight.detach().to(DEV), bn.bias.detach().to(DEV)
    # eval
    bn.eval()
    capi.bn_lrelu_fwd(x.detach().to(DEV), F, N, F, gam, bet, rm, rv, 0.1, 1e-5, 0.01, False, saved, y, F, ws)
    _close(y, act(bn(x)), 1e-5)
    if N == 1:
        return  # torch refuses batch statistics over a single row
    bn.train()
    want = act(bn(x))
    gout = torch.randn(N, F)
    want.backward(gout)
    capi.bn_lrelu_fwd(x.detach().to(DEV), F, N, F, gam, bet, rm, rv, 0.1, 1e-5, 0.01, True, saved, y, F, ws)
    _close(y, want, 2e-5)
    _close(rm, bn.running_mean, 1e-5)
    _close(rv, bn.running_var, 1e-5)
    dx = torch.zeros(N, F, device=DEV)
    dg, db = torch.zeros(F, device=DEV), torch.zeros(F, device=DEV)
    capi.bn_lrelu_bwd(x.detach().to(DEV), F, N, F, gam, bet, saved, 0.01, gout.to(DEV), F, dx, F, dg, db, ws)
    _close(dx, x.grad, 5e-5, 1e-4)
    _close(dg, bn.weight.grad, 1e-4, 1e-4)
    _close(db, bn.bias.grad, 1e-4, 1e-4)


@pytest.mark.parametrize("weighted,mapped", [(False, False), (True, False), (False, True)])
def test_cross_entropy(capi, weighted, mapped):
    C, n = 6, 777
    torch.manual_seed(2)
    rows = n + 100 if mapped else n
    logits = (torch.randn(rows, C) * 3).requires_grad_()
    ys = torch.randint(0, C, (n,))
    w = torch.tensor([1 / 0.086747, 1 / 0.144406, 1 / 0.227883, 1 / 0.160585, 1 / 0.127711, 1 / 0.252668]) if weighted else None
    row_map = torch.sort(torch.randperm(rows)[:n]).values.to(torch.int32) if mapped else None
    sel = logits[row_map.long()] if mapped else logits
    loss = torch.nn.functional.cross_entropy(sel, ys, weight=w)
    loss.backward()
    dl = torch.zeros(rows, C, device=DEV)
    stats = torch.zeros(256, device=DEV)
    capi.cross_entropy(logits.detach().to(DEV), C, C, n, row_map.to(DEV) if mapped else None, ys.to(DEV),
                       w.to(DEV) if weighted else None, 1.0, dl, C, stats)
    s = stats.cpu()
    assert abs(float(s[0]) - float(loss)) < 2e-6 * max(1.0, float(loss))
    assert int(s[1]) == int((sel.argmax(-1) == ys).sum())
    _close(dl, logits.grad, 1e-7, 1e-4)


@pytest.mark.parametrize("decoupled,wd,clip", [(False, 1e-8, 0.0), (False, 3e-5, 0.0), (True, 1e-2, 5.0), (True, 1e-2, 0.05)])
def test_fused_adam_matches_torch(capi, decoupled, wd, clip):
    n = 10_007
    torch.manual_seed(3)
    p0 = torch.randn(n)
    ref = torch.nn.Parameter(p0.clone())
    opt = (torch.optim.AdamW if decoupled else torch.optim.Adam)([ref], lr=1e-3, weight_decay=wd)
    p = p0.clone().to(DEV)
    m, v = torch.zeros(n, device=DEV), torch.zeros(n, device=DEV)
    state = torch.zeros(4 + 512, dtype=torch.int64, device=DEV)      # ERC_ADAM_STATE_WORDS
    state[2] = 1
    gnorm, ws = torch.zeros(1, device=DEV), torch.zeros(1024, device=DEV)
    for it in range(4):
        g = torch.randn(n) * (0.01 if it % 2 else 1.0)
        ref.grad = g.clone()
        if clip > 0:
            torch.nn.utils.clip_grad_norm_([ref], clip)
        opt.step()
        gd = g.to(DEV)
        if clip > 0:
            capi.grad_norm(gd, n, 1.0, gnorm, ws)
            assert abs(float(gnorm) - float(g.norm())) < 1e-4 * float(g.norm())
        capi.adam_step(p, gd, m, v, n, 1e-3, 0.9, 0.999, 1e-8, wd, decoupled, 1.0, clip, gnorm if clip > 0 else None, state)
        _close(p, ref.detach(), 2e-6, 1e-5)
    st = state.cpu().tolist()
    n_wg = (n // 4 + 255) // 256
    # state[4 ..): the workgroups' private copies of the step count, ALL kept current (a launch with another grid owns others)
    assert st[:4] == [4, 4, 1, 0] and st[4:] == [4] * 512 and n_wg < 512


@pytest.mark.parametrize("C,weighted,p", [(6, True, 0.5), (7, False, 0.0), (4, True, 0.0)])
def test_head_ce_matches_torch(capi, C, weighted, p):
    """erc_head_ce: last Linear + F.cross_entropy + their backward through the ReLU / inverted-dropout mask in one launch
    (dgcn_models.py:163-170) against torch autograd."""
    torch.manual_seed(C)
    N, Fh = 531, 100
    pre = torch.randn(N, Fh)
    keep = (torch.rand(N, Fh) >= p).float() / (1.0 - p) if p > 0 else torch.ones(N, Fh)
    W, b = torch.randn(C, Fh) * 0.2, torch.randn(C) * 0.1
    y = torch.randint(0, C, (N,))
    w = torch.rand(C) + 0.5 if weighted else None
    pre_t = pre.clone().requires_grad_(True)
    Z = torch.relu(pre_t) * keep
    logits = Z @ W.t() + b
    logits.retain_grad()
    loss = torch.nn.functional.cross_entropy(logits, y, weight=w)
    loss.backward()
    Zd = Z.detach().to(DEV)
    out_l, out_dl, out_dz = torch.zeros(N, C, device=DEV), torch.zeros(N, C, device=DEV), torch.zeros(N, Fh, device=DEV)
    stats = torch.zeros(capi.head_ce_stats_floats(N), device=DEV)
    for _ in range(2):      # twice: the arrival counter in stats must be back at zero
        capi.head_ce(Zd, Fh, Fh, C, N, W.to(DEV), b.to(DEV), y.to(DEV), w.to(DEV) if weighted else None, 1.0 / (1.0 - p),
                     out_l, C, out_dl, C, out_dz, Fh, stats)
    _close(out_l, logits.detach(), 2e-5, 1e-5)
    _close(out_dl, logits.grad, 1e-6, 1e-4)
    _close(out_dz, pre_t.grad, 2e-6, 1e-4)       # = dZ through relu and the dropout mask
    st = stats.cpu()
    assert abs(float(st[0]) - float(loss.detach())) < 2e-5 and int(st[1]) == int((logits.argmax(-1) == y).sum())


def test_wgrad_table_runs_of_equal_records_and_more_than_32_runs(capi):
    """erc_wgrad_table finds a work item's record from by-value RUNS of equally sized records (MMGCN: the 128 weight gradients of
    the GCNII chain in one launch): 36 records of one shape followed by 40 records of 40 different sizes = 41 runs, two launches."""
    from erc_amd.engine import GemmPlanner
    g = torch.Generator().manual_seed(78)
    pl = GemmPlanner(DEV, 16)
    shapes = [(300, 40, 24)] * 36 + [(64 + 7 * i, 8 + 3 * i, 70 + i) for i in range(40)]
    refs = []
    for K, M, N in shapes:
        A = torch.randn(K, M, generator=g).to(DEV)
        Bd = torch.randn(K, N, generator=g).to(DEV)
        Cm = torch.full((M, N), float("nan"), device=DEV)
        pl.defer(A, M, Bd, N, Cm, N, M, N, K, 0, None)
        refs.append((Cm, _gemm_ref(A.cpu().t(), Bd.cpu()), K))
    cache = {}
    pl.flush_wgrads(cache)
    torch.cuda.synchronize()
    assert int(cache["wgrad_counters"].abs().sum()) == 0
    for Cm, ref, K in refs:
        _close(Cm, ref, 3e-4 * math.sqrt(max(K, 100) / 100))


def test_wgrad_table(capi):
    """erc_wgrad_table: several dW = A^T B[gather] products in one launch (fp32 / bf16 B, bias strips, vector and
    scalar access paths, split and unsplit K), launched twice to check that the arrival counters are left zero."""
    from erc_amd.engine import GemmPlanner
    g = torch.Generator().manual_seed(77)
    cases = [  # K, M, N, ones, gather, bf16
        (1982, 100, 1380, 1, True, True),
        (1982, 900, 100, 2, False, False),
        (1982, 400, 100, 1, False, False),
        (1982, 6, 100, 1, False, False),
        (333, 100, 1380, 1, True, False),
        (61, 7, 13, 2, False, False),
        (9000, 68, 72, 1, True, True),
        (3, 100, 100, 1, False, False),
    ]
    pl = GemmPlanner(DEV, 16)
    refs = []
    for K, M, N, ones, gather, bf16 in cases:
        A = torch.randn(K, M, generator=g).to(DEV)
        rows_b = K + 50 if gather else K
        Bm = torch.randn(rows_b, N, generator=g)
        if bf16:
            Bm = Bm.bfloat16()
        gi = torch.randperm(rows_b, generator=g)[:K].int().to(DEV) if gather else None
        Bd = Bm.to(DEV)
        Cm = torch.full((M, N), float("nan"), device=DEV)
        bo = torch.full((M if ones == 1 else N,), float("nan"), device=DEV)
        pl.defer(A, M, Bd, N, Cm, N, M, N, K, ones, bo, gather=gi)
        Bg = Bm.float()[gi.cpu().long()] if gather else Bm.float()
        refs.append((Cm, bo, _gemm_ref(A.cpu().t(), Bg), (A.cpu() if ones == 1 else Bg).double().sum(0).float(), K))
    cache = {}
    for rep in range(2):
        if rep:      # a flush forgets its records (several flushes per step: engine.GemmPlanner.flush_wgrads): hand them back
            pl.deferred, pl.flushed = pl.flushed, []
        pl.flush_wgrads(cache)
        torch.cuda.synchronize()
        assert not pl.deferred and len(pl.flushed) == len(cases)
        assert int(cache["wgrad_counters"].abs().sum()) == 0
        for Cm, bo, ref, bref, K in refs:
            tol = 3e-4 * math.sqrt(max(K, 100) / 100)
            _close(Cm, ref, tol)
            _close(bo, bref, tol)
            Cm.fill_(float("nan")), bo.fill_(float("nan"))


@pytest.mark.parametrize("M,K,N,w_bf16", [(1982, 1380, 100, True), (500, 712, 100, False), (700, 1242, 37, True),
                                          (9000, 1380, 100, True), (8200, 712, 100, True), (8300, 96, 24, True)])
def test_gemm_bf16a_projection(capi, M, K, N, w_bf16):
    """Input projection on a bf16 feature block: C = relu(X[gather] W^T + b), fp32 accumulate.  Small row counts use the
    streaming kernel, >= 8192 rows (bf16 W) the persistent kernel with the weights resident in registers."""
    g = torch.Generator().manual_seed(M + K)
    X = torch.randn(M + 40, K, generator=g).bfloat16()
    Wf = (torch.randn(N, K, generator=g) / math.sqrt(K))
    W = Wf.bfloat16() if w_bf16 else Wf
    b = torch.randn(N, generator=g)
    idx = torch.randperm(M + 40, generator=g)[:M].int()
    out = torch.full((M, N), float("nan"), device=DEV)
    capi.gemm_bf16a_stream(X.to(DEV), K, idx.to(DEV), W.to(DEV), K, out, N, M, N, K, bias=b.to(DEV), act=1)
    ref = torch.relu(X[idx.long()].double() @ W.bfloat16().double().t() + b.double()).float()
    _close(out, ref, 2e-3)


@pytest.mark.parametrize("K", [1, 7, 130, 1982, 9001])
def test_wgrad_bf16_batched_launch(K):
    """csrc/wgrad_bf16.hip (every weight gradient of the COGMEN bf16 step as one launch, both operands bf16 in memory)
    against float64 products of the same bf16 values: the five record shapes of the step -- gathered wide B, transposed
    stores, N not a multiple of 4, padded pitches whose pad columns hold (finite) garbage -- and bit-reproducibility."""
    import torch
    from erc_amd.engine import GemmPlanner
    dev = "cuda:0"
    g = torch.Generator(device="cpu").manual_seed(K)
    rnd = lambda *s: torch.randn(*s, generator=g)
    R = K + 5                                                     # rows of the gathered operand's source block
    recs = []   # (A [K, lda] bf16, M, B, N, gather, ct, bias_a?, bias_b?)
    def pad(t, ld, fill):
        out = torch.full((t.shape[0], ld), fill, dtype=torch.float32)
        out[:, :t.shape[1]] = t
        return out.to(torch.bfloat16).to(dev)
    gather = torch.randint(0, R, (K,), generator=g).to(torch.int32)
    shapes = [(100, 104, 1380, 1380, True, False, True, False),   # dW1 = dH0^T X[gather]
              (100, 104, 900, 904, False, True, True, False),     # d[W_r; root]^T
              (100, 104, 400, 400, False, True, False, True),     # d[q;k;v;s]
              (100, 104, 100, 104, False, False, True, False),    # cls.0
              (100, 104, 6, 8, False, True, False, True),         # cls.3 (N = 6: no vector stores along n)
              (36, 40, 52, 52, False, False, True, True)]         # small odd sizes
    pl = GemmPlanner(dev, 64)
    cache, want = {}, []
    for M, lda, N, ldb, gath, ct, ba, bb in shapes:
        A = pad(rnd(K, M), lda, 3.0)                              # pad columns: finite garbage, must not leak into the result
        Bsrc = pad(rnd(R if gath else K, N), ldb, -2.0)
        C = torch.full((N, M) if ct else (M, N), float("nan"), device=dev)
        bias_a = torch.full((M,), float("nan"), device=dev) if ba else None
        bias_b = torch.full((N,), float("nan"), device=dev) if bb else None
        gi = gather.to(dev) if gath else None
        pl.defer16(A, lda, Bsrc, ldb, C, M if ct else N, M, N, K, ct=ct, bias_a=bias_a, bias_b=bias_b, gather=gi)
        Ad = A[:, :M].double().cpu()
        Bd = (Bsrc[gather.long().to(dev)] if gath else Bsrc)[:, :N].double().cpu()
        ref = Ad.t() @ Bd
        want.append((C, ref.t() if ct else ref, bias_a, Ad.sum(0), bias_b, Bd.sum(0)))
    pl.flush_wgrads_bf16(cache)
    torch.cuda.synchronize()
    first = [c.clone() for c, *_ in want]
    for C, ref, bias_a, ra, bias_b, rb_ in want:
        scale = float(ref.abs().max()) + 1e-6
        assert float((C.double().cpu() - ref).abs().max()) < 2e-5 * scale + 1e-5 * (K ** 0.5), (tuple(C.shape), K)
        if bias_a is not None:
            assert float((bias_a.double().cpu() - ra).abs().max()) < 1e-5 * (float(ra.abs().max()) + K ** 0.5)
        if bias_b is not None:
            assert float((bias_b.double().cpu() - rb_).abs().max()) < 1e-5 * (float(rb_.abs().max()) + K ** 0.5)
    for C, *_ in want:
        C.fill_(float("nan"))
    pl.flush_wgrads_bf16(cache)                                   # same table, slabs and counters: a second launch
    torch.cuda.synchronize()
    assert all(torch.equal(a, c) for a, (c, *_) in zip(first, want))


def test_wgrad_three_term_bf16_split_is_fp32_class():
    """csrc/wgrad.hip MB == 2 (erc_wgrad_table_x3): fp32 operands split into three bf16 terms each, six cross products on the
    bf16 matrix cores, fp32 accumulation.  Against float64 the error must be that of an fp32 product (the exact-fp32 path of
    the same launch is the yardstick), three orders of magnitude below a single bf16 rounding -- on operands whose entries
    span six decades."""
    import torch
    from erc_amd.engine import GemmPlanner
    dev = "cuda:0"
    g = torch.Generator().manual_seed(5)
    K, M, N = 3001, 200, 400
    A = (torch.randn(K, M, generator=g) * torch.logspace(-3, 3, M)).to(dev)       # column scales 1e-3 .. 1e3
    B = (torch.randn(K, N, generator=g) * torch.logspace(2, -2, N)).to(dev)
    ref = A.double().t() @ B.double()
    errs = {}
    for mode in (0, 2, 1):
        pl = GemmPlanner(dev, 64)
        C = torch.full((M, N), float("nan"), device=dev)
        bias = torch.full((M,), float("nan"), device=dev)
        pl.defer(A, M, B, N, C, N, M, N, K, 1, bias, mma_bf16=mode)
        pl.flush_wgrads({})
        torch.cuda.synchronize()
        # error relative to the size of the terms that enter each entry (|A|^T |B|): cancellation does not count against a path
        scale = A.double().abs().t() @ B.double().abs()
        errs[mode] = float(((C.double() - ref).abs() / scale).max())
        assert float((bias.double() - A.double().sum(0)).abs().max()) <= 1e-5 * float(A.double().abs().sum(0).max())
    assert errs[2] < 4 * max(errs[0], 1e-7), errs          # fp32 class ...
    assert errs[2] < 1e-6 and errs[1] > 1e-4, errs        # ... not bf16 class


@pytest.mark.parametrize("M,N,K,split", [(300, 260, 200, 1), (131, 200, 1280, 4), (64, 128, 36, 1), (1, 1, 4, 1), (129, 1, 68, 1)])
def test_gemm_x3_is_fp32_class(capi, M, N, K, split):
    """erc_gemm_x3: C = A B^T on the bf16 matrix cores from a three-term split of both fp32 operands.  Against the float64
    product its error is of the size of the exact-fp32 kernel's (erc_gemm_f32: an fp32 fma chain), orders below one bf16
    rounding; ragged tiles, K not a multiple of the 32-chunk, split-K slabs."""
    torch.manual_seed(M + K)
    A = torch.randn(M, K) * torch.logspace(-3, 2, K)       # wide dynamic range along K
    Bm = torch.randn(N, K)
    ref = A.double() @ Bm.double().t()
    Ad, Bd = A.to(DEV), Bm.to(DEV)
    slab = M * N
    out = torch.full((split * slab,), float("nan"), device=DEV)
    capi.gemm_x3(Ad, K, Bd, K, out, N, M, N, K, split_k=split, c_slab=slab)
    C3 = out.view(split, M, N).sum(0).cpu().double()
    C32 = torch.zeros(M, N, device=DEV)
    capi.gemm_f32(Ad, K, 0, None, Bd, K, 0, None, C32, N, M, N, K)
    scale = (A.double().abs() @ Bm.double().abs().t())       # sum |a||b|: the natural error scale of a dot product
    e3 = float(((C3 - ref).abs() / scale).max())
    e32 = float(((C32.cpu().double() - ref).abs() / scale).max())
    ebf = float((((A.bfloat16().double() @ Bm.bfloat16().double().t()) - ref).abs() / scale).max())
    assert e3 <= max(4 * e32, 3e-7), (e3, e32)
    assert e3 < 1e-3 * ebf, (e3, ebf)


def test_gemm_x3_grouped_equals_the_exact_fp32_grouped_product(capi):
    """erc_gemm_x3_grouped: the per-(dialogue, modality) blocks dg z^T of MMGCN's adjacency gradient (K = planes * 200 contiguous
    per row) against erc_gemm_f32_grouped form 1 with planes, both through split-K slabs: fp32-class agreement, rows / columns
    beyond a dialogue's length untouched."""
    torch.manual_seed(5)
    lens, Mo, FD, NP = [7, 33, 1, 128, 50], 3, 200, 4
    B, N, P = len(lens), sum(lens), 128
    off = torch.tensor([0] + list(torch.tensor(lens).cumsum(0)), dtype=torch.int32, device=DEV)
    A, Z = torch.randn(Mo * N, NP * FD, device=DEV), torch.randn(Mo * N, NP * FD, device=DEV)
    n_adj, S = B * Mo * P * P, 5
    ref_s, got_s = torch.zeros(S * n_adj, device=DEV), torch.full((S * n_adj,), 7.0, device=DEV)
    capi.gemm_grouped(1, A, NP * FD, Z, NP * FD, ref_s, P, FD, off, B, Mo, N, max(lens), P, planes=NP, a_plane=FD, b_plane=FD,
                      split=S, c_slab=n_adj)
    capi.gemm_x3_grouped(A, NP * FD, Z, NP * FD, got_s, P, off, B, Mo, N, max(lens), NP * FD, split_k=S, c_slab=n_adj)
    ref, got = ref_s.view(S, B * Mo, P, P).sum(0), got_s.view(S, B * Mo, P, P)
    for b, L in enumerate(lens):
        for m in range(Mo):
            blk = got[:, b * Mo + m]
            assert bool((blk[:, L:, :] == 7.0).all()) and bool((blk[:, :, L:] == 7.0).all())
            want = (A[m * N + int(off[b]):m * N + int(off[b]) + L].double() @ Z[m * N + int(off[b]):m * N + int(off[b]) + L].double().t())
            e3 = float((blk[:, :L, :L].sum(0).double() - want).abs().max())
            e32 = float((ref[b * Mo + m, :L, :L].double() - want).abs().max())
            assert e3 <= max(4 * e32, 1e-4), (b, m, e3, e32)
