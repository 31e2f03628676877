"""DialogueGCN on the GPU: (a) EdgeAtt + basis RGCNConv pieces directly against golden vectors produced by the
REFERENCE's own dgcn_models.EdgeAtt / batch_graphify / models.rgcn.RGCNConv, (b) the whole module (eval logits,
train-mode loss + every live gradient, one optimizer step) against the reference-pinned CPU oracle."""
import math

import numpy as np
import pytest
import torch
from torch.nn import functional as F

from tests.util_cases import check_grad_digest, fill_params, make_batch, rel_err, to_device

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


@pytest.mark.parametrize("name", ["dgcn_s2", "dgcn_s9"])
def test_edge_att_and_basis_rgcn_vs_reference_golden(golden, name):
    from erc_amd import capi
    from erc_amd.cogmen import build_graph_tensors
    fx = golden(name)
    S, R, NB, Fd, O = int(fx["n_speakers"]), 2 * int(fx["n_speakers"]) ** 2, 30, 200, 100
    lengths, spk = torch.from_numpy(fx["lengths"]).to(DEV), torch.from_numpy(fx["speakers"]).to(DEV)
    feats = torch.from_numpy(fx["features"])
    B, T = feats.shape[:2]
    g, ei, et = build_graph_tensors(lengths, spk, 10, 10, S)
    N, E = g["counts"].cpu().tolist()
    np.testing.assert_array_equal(ei[:, :E].cpu().numpy(), fx["edge_index"])     # bit-exact edges / relation ids
    np.testing.assert_array_equal(et[:E].cpu().numpy(), fx["edge_type"])
    # parameters exactly as make_golden filled them
    att_w = torch.nn.Module(); att_w.weight = torch.nn.Parameter(torch.zeros(Fd, Fd)); fill_params(att_w, int(fx["att_seed"]))
    conv = torch.nn.Module()
    conv.basis, conv.att = torch.nn.Parameter(torch.zeros(NB, Fd, O)), torch.nn.Parameter(torch.zeros(R, NB))
    conv.root, conv.bias = torch.nn.Parameter(torch.zeros(Fd, O)), torch.nn.Parameter(torch.zeros(O))
    fill_params(conv, int(fx["conv_seed"]))
    W, basis, attp, root, bias = [t.detach().to(DEV) for t in (att_w.weight, conv.basis, conv.att, conv.root, conv.bias)]
    x = torch.zeros(N, Fd, device=DEV)
    capi.gather_rows(feats.to(DEV).view(B * T, Fd), Fd, g["node_row"], N, Fd, x, Fd)
    ATT = torch.zeros(N, Fd, device=DEV)
    capi.gemm_f32(x, Fd, 0, None, W, Fd, 0, None, ATT, Fd, N, Fd, Fd)
    norm = torch.zeros(E, device=DEV)
    capi.edge_att_fwd(x, Fd, ATT, Fd, Fd, N, g, norm)
    np.testing.assert_allclose(norm.cpu().numpy(), fx["edge_norm"], atol=2e-6, rtol=2e-5)
    Z = torch.zeros(N, NB * Fd, device=DEV)
    capi.brgcn_agg_fwd(x, Fd, Fd, N, g, norm, attp, NB, Z)
    out = torch.zeros(N, O, device=DEV)
    capi.gemm_f32(Z, NB * Fd, 0, None, basis, O, 1, None, out, O, N, O, NB * Fd, bias=bias)
    capi.gemm_f32(x, Fd, 0, None, root, O, 1, None, out, O, N, O, Fd, accumulate=1)
    np.testing.assert_allclose(out.cpu().numpy(), fx["rgcn_out"], atol=1e-4, rtol=1e-4)
    # backward of the two operators from the fixture's upstream gradient
    gout = torch.from_numpy(fx["gout"]).to(DEV)
    dZ = torch.zeros(N, NB * Fd, device=DEV)
    capi.gemm_f32(gout, O, 0, None, basis, O, 0, None, dZ, NB * Fd, N, NB * Fd, O)
    dnorm, TT, datt = torch.zeros(E, device=DEV), torch.zeros(E, NB, device=DEV), torch.zeros(R, NB, device=DEV)
    capi.brgcn_bwd_edges(x, Fd, Fd, N, R, g, norm, attp, NB, dZ, dnorm, TT, datt)
    dbasis, dbias = torch.zeros(NB * Fd, O, device=DEV), torch.zeros(O, device=DEV)
    capi.gemm_f32(Z, NB * Fd, 1, None, gout, O, 1, None, dbasis, O, NB * Fd, O, N, ones_col=2, bias_out=dbias)
    droot = torch.zeros(Fd, O, device=DEV)
    capi.gemm_f32(x, Fd, 1, None, gout, O, 1, None, droot, O, Fd, O, N)
    U, basisT = torch.zeros(N, NB * O, device=DEV), torch.zeros(NB * O, Fd, device=DEV)
    capi.brgcn_bwd_source(gout, O, O, N, g, norm, attp, NB, U)
    capi.transpose_batched(basis, NB, Fd, O, basisT)
    dx = torch.zeros(N, Fd, device=DEV)
    capi.gemm_f32(U, NB * O, 0, None, basisT, Fd, 1, None, dx, Fd, N, Fd, NB * O)
    capi.gemm_f32(gout, O, 0, None, root, O, 0, None, dx, Fd, N, Fd, O, accumulate=1)
    DATT, dscore = torch.zeros(N, Fd, device=DEV), torch.zeros(E, device=DEV)
    capi.edge_att_bwd(x, Fd, ATT, Fd, Fd, N, g, norm, dnorm, dx, Fd, 1, DATT, Fd, dscore)
    dW = torch.zeros(Fd, Fd, device=DEV)
    capi.gemm_f32(DATT, Fd, 1, None, x, Fd, 1, None, dW, Fd, Fd, Fd, N)
    capi.gemm_f32(DATT, Fd, 0, None, W, Fd, 1, None, dx, Fd, N, Fd, Fd, accumulate=1)
    dfeat = torch.zeros(B * T, Fd, device=DEV)
    capi.gather_rows(dx, Fd, g["node_row"], N, Fd, dfeat, Fd, scatter=1)
    np.testing.assert_allclose(dfeat.cpu().view(B, T, Fd).numpy(), fx["dfeatures"], atol=2e-4, rtol=2e-3)
    check_grad_digest(fx, [("edge_att.weight", dW), ("conv1.basis", dbasis.view(NB, Fd, O)), ("conv1.att", datt),
                           ("conv1.root", droot), ("conv1.bias", dbias)], tol=2e-3)


def test_relation_space_rgcn_vs_reference_golden(golden):
    """The two-speaker fixture (R = 8 < 30 bases) through the relation-space entry points: W_r composed first, Z [N, 8F],
    comp / basis gradients from dW_r -- against the same reference outputs as the basis-space path above."""
    from erc_amd import capi
    from erc_amd.cogmen import build_graph_tensors
    fx = golden("dgcn_s2")
    S, R, NB, Fd, O = 2, 8, 30, 200, 100
    assert R <= capi.rrgcn_max_relations()
    lengths, spk = torch.from_numpy(fx["lengths"]).to(DEV), torch.from_numpy(fx["speakers"]).to(DEV)
    feats = torch.from_numpy(fx["features"])
    B, T = feats.shape[:2]
    g, ei, et = build_graph_tensors(lengths, spk, 10, 10, S)
    N, E = g["counts"].cpu().tolist()
    att_w = torch.nn.Module(); att_w.weight = torch.nn.Parameter(torch.zeros(Fd, Fd)); fill_params(att_w, int(fx["att_seed"]))
    conv = torch.nn.Module()
    conv.basis, conv.att = torch.nn.Parameter(torch.zeros(NB, Fd, O)), torch.nn.Parameter(torch.zeros(R, NB))
    conv.root, conv.bias = torch.nn.Parameter(torch.zeros(Fd, O)), torch.nn.Parameter(torch.zeros(O))
    fill_params(conv, int(fx["conv_seed"]))
    W, basis, comp, root, bias = [t.detach().to(DEV) for t in (att_w.weight, conv.basis, conv.att, conv.root, conv.bias)]
    z = lambda *s: torch.zeros(*s, device=DEV)
    x = z(N, Fd)
    capi.gather_rows(feats.to(DEV).view(B * T, Fd), Fd, g["node_row"], N, Fd, x, Fd)
    ATT, norm = z(N, Fd), z(E)
    capi.gemm_f32(x, Fd, 0, None, W, Fd, 0, None, ATT, Fd, N, Fd, Fd)
    capi.edge_att_fwd(x, Fd, ATT, Fd, Fd, N, g, norm)
    Wr, WrT = z(R, Fd, O), z(R, O, Fd)
    capi.basis_compose(comp, basis, R, NB, Fd, O, Wr, WrT)
    Wr_ref = torch.einsum("rb,bfo->rfo", comp.double(), basis.double())
    assert float((Wr.double() - Wr_ref).abs().max()) < 1e-6
    assert torch.equal(WrT, Wr.transpose(1, 2).contiguous())
    Z = z(N, R * Fd)
    capi.rrgcn_agg_fwd(x, Fd, Fd, N, R, g, norm, Z)
    out = z(N, O)
    capi.gemm_f32(Z, R * Fd, 0, None, Wr, O, 1, None, out, O, N, O, R * Fd, bias=bias)
    capi.gemm_f32(x, Fd, 0, None, root, O, 1, None, out, O, N, O, Fd, accumulate=1)
    np.testing.assert_allclose(out.cpu().numpy(), fx["rgcn_out"], atol=1e-4, rtol=1e-4)
    gout = torch.from_numpy(fx["gout"]).to(DEV)
    dZ, dnorm = z(N, R * Fd), z(E)
    capi.gemm_f32(gout, O, 0, None, Wr, O, 0, None, dZ, R * Fd, N, R * Fd, O)
    capi.rrgcn_bwd_edges(x, Fd, Fd, N, R, g, dZ, dnorm)
    dWr, dbias = z(R * Fd, O), z(O)
    capi.gemm_f32(Z, R * Fd, 1, None, gout, O, 1, None, dWr, O, R * Fd, O, N, ones_col=2, bias_out=dbias)
    dbasis, dcomp = z(NB, Fd, O), z(R, NB)
    capi.basis_decompose(comp, basis, dWr, R, NB, Fd * O, dbasis, dcomp)
    droot = z(Fd, O)
    capi.gemm_f32(x, Fd, 1, None, gout, O, 1, None, droot, O, Fd, O, N)
    U, dx = z(N, R * O), z(N, Fd)
    capi.rrgcn_bwd_source(gout, O, O, N, R, g, norm, U)
    capi.gemm_f32(U, R * O, 0, None, WrT, Fd, 1, None, dx, Fd, N, Fd, R * O)
    capi.gemm_f32(gout, O, 0, None, root, O, 0, None, dx, Fd, N, Fd, O, accumulate=1)
    DATT, dscore = z(N, Fd), z(E)
    capi.edge_att_bwd(x, Fd, ATT, Fd, Fd, N, g, norm, dnorm, dx, Fd, 1, DATT, Fd, dscore)
    dW = z(Fd, Fd)
    capi.gemm_f32(DATT, Fd, 1, None, x, Fd, 1, None, dW, Fd, Fd, Fd, N)
    capi.gemm_f32(DATT, Fd, 0, None, W, Fd, 1, None, dx, Fd, N, Fd, Fd, accumulate=1)
    dfeat = z(B * T, Fd)
    capi.gather_rows(dx, Fd, g["node_row"], N, Fd, dfeat, Fd, scatter=1)
    np.testing.assert_allclose(dfeat.cpu().view(B, T, Fd).numpy(), fx["dfeatures"], atol=2e-4, rtol=2e-3)
    check_grad_digest(fx, [("edge_att.weight", dW), ("conv1.basis", dbasis), ("conv1.att", dcomp),
                           ("conv1.root", droot), ("conv1.bias", dbias)], tol=2e-3)


@pytest.mark.parametrize("name", ["dgcn_s2", "dgcn_s9"])
def test_fused_rgcn_forward_tile_vs_unfused_and_reference_golden(golden, name):
    """erc_brgcn_fwd_tile (aggregate + basis product + root product in one launch, fp32 matrix cores) against the unfused
    kernels on the reference fixture's graph, and against the fixture's own RGCN output."""
    from erc_amd import capi
    from erc_amd.cogmen import build_graph_tensors
    fx = golden(name)
    S, R, NB, Fd, O = int(fx["n_speakers"]), 2 * int(fx["n_speakers"]) ** 2, 30, 200, 100
    lengths, spk = torch.from_numpy(fx["lengths"]).to(DEV), torch.from_numpy(fx["speakers"]).to(DEV)
    feats = torch.from_numpy(fx["features"])
    B, T = feats.shape[:2]
    g, _, _ = build_graph_tensors(lengths, spk, 10, 10, S)
    N, E = g["counts"].cpu().tolist()
    att_w = torch.nn.Module(); att_w.weight = torch.nn.Parameter(torch.zeros(Fd, Fd)); fill_params(att_w, int(fx["att_seed"]))
    conv = torch.nn.Module()
    conv.basis, conv.att = torch.nn.Parameter(torch.zeros(NB, Fd, O)), torch.nn.Parameter(torch.zeros(R, NB))
    conv.root, conv.bias = torch.nn.Parameter(torch.zeros(Fd, O)), torch.nn.Parameter(torch.zeros(O))
    fill_params(conv, int(fx["conv_seed"]))
    W, basis, attp, root, bias = [t.detach().to(DEV) for t in (att_w.weight, conv.basis, conv.att, conv.root, conv.bias)]
    z = lambda *s: torch.zeros(*s, device=DEV)
    XW = Fd + O                                  # the module's row pitch: features next to the graph output
    xw = z(N, XW)
    capi.gather_rows(feats.to(DEV).view(B * T, Fd), Fd, g["node_row"], N, Fd, xw, XW)
    ATT, norm = z(N, Fd), z(E)
    capi.gemm_f32(xw, XW, 0, None, W, Fd, 0, None, ATT, Fd, N, Fd, Fd)
    capi.edge_att_fwd(xw, XW, ATT, Fd, Fd, N, g, norm)
    Z_ref = z(N, NB * Fd)
    capi.brgcn_agg_fwd(xw, XW, Fd, N, g, norm, attp, NB, Z_ref)
    out_ref = z(N, O)
    capi.gemm_f32(Z_ref, NB * Fd, 0, None, basis, O, 1, None, out_ref, O, N, O, NB * Fd, bias=bias)
    capi.gemm_f32(xw, XW, 0, None, root, O, 1, None, out_ref, O, N, O, Fd, accumulate=1)
    Zt, slabs, out = torch.full((N, NB * Fd), float("nan"), device=DEV), torch.full((capi.brgcn_fwd_tile_slabs(), N, O), float("nan"), device=DEV), z(N, O)
    capi.poison_lds()
    capi.brgcn_fwd_tile(xw, XW, Fd, O, N, g, norm, attp, NB, basis, root, Zt, slabs)
    capi.slab_reduce(slabs, capi.brgcn_fwd_tile_slabs(), N * O, bias, O, 0, out, N * O)
    # (comparisons on the host: no device-side reductions over freshly NaN-filled buffers)
    cpu = lambda t: t.detach().cpu()
    assert torch.equal(cpu(Zt), cpu(Z_ref))                # the same per-edge multiply-adds in the same order
    assert float((cpu(out) - cpu(out_ref)).abs().max()) < 2e-5 * max(1.0, float(cpu(out_ref).abs().max()))
    np.testing.assert_allclose(cpu(out).numpy(), fx["rgcn_out"], atol=1e-4, rtol=1e-4)
    # node side of the backward: dx += sum_b U_b basis_b^T + gout root^T, fused vs the separate kernels
    gout = torch.from_numpy(fx["gout"]).to(DEV)
    U, basisT = z(N, NB * O), z(NB * O, Fd)
    dx0 = torch.randn(N, XW, device=DEV)
    dx_ref = dx0.clone()
    capi.brgcn_bwd_source(gout, O, O, N, g, norm, attp, NB, U)
    capi.transpose_batched(basis, NB, Fd, O, basisT)
    capi.gemm_f32(U, NB * O, 0, None, basisT, Fd, 1, None, dx_ref, XW, N, Fd, NB * O, accumulate=1)
    capi.gemm_f32(gout, O, 0, None, root, O, 0, None, dx_ref, XW, N, Fd, O, accumulate=1)
    S = capi.brgcn_fwd_tile_slabs()
    dslabs, dx = torch.full((S, N, Fd), float("nan"), device=DEV), dx0.clone()
    capi.poison_lds()
    capi.brgcn_bwd_source_tile(gout, O, Fd, O, N, g, norm, attp, NB, basis, root, dslabs)
    capi.slab_reduce(dslabs, S, N * Fd, None, Fd, 4, dx, N * Fd, ld_out=XW)
    # edge side of the backward: d norm, TT and d att, fused (dZ in LDS) vs GEMM + erc_brgcn_bwd_edges
    dZ = z(N, NB * Fd)
    capi.gemm_f32(gout, O, 0, None, basis, O, 0, None, dZ, NB * Fd, N, NB * Fd, O)
    dn_ref, TT_ref, datt_ref = z(E), z(E, NB), z(R, NB)
    capi.brgcn_bwd_edges(xw, XW, Fd, N, R, g, norm, attp, NB, dZ, dn_ref, TT_ref, datt_ref)
    dn_sl, TTf, dattf, dn = torch.full((S, E), float("nan"), device=DEV), torch.full((E, NB), float("nan"), device=DEV), z(R, NB), z(E)
    capi.poison_lds()
    capi.brgcn_bwd_edges_tile(xw, XW, Fd, O, N, R, g, norm, attp, NB, basis, gout, O, TTf, dn_sl, E, dattf)
    capi.slab_reduce(dn_sl, S, E, None, 0, 0, dn, E)
    for got, want, what in ((dn, dn_ref, "dnorm"), (TTf, TT_ref, "TT"), (dattf, datt_ref, "datt")):
        got, want = cpu(got), cpu(want)
        e = float((got - want).abs().max()) / max(1e-6, float(want.abs().max()))
        assert e < 2e-5, (what, e)
    dx, dx0, dx_ref = cpu(dx), cpu(dx0), cpu(dx_ref)
    assert torch.equal(dx[:, Fd:], dx0[:, Fd:])           # the columns next to the features are not touched
    sc = max(1.0, float((dx_ref - dx0).abs().max()))
    assert float((dx - dx_ref).abs().max()) < 2e-5 * sc, float((dx - dx_ref).abs().max())
    # a second run of the three launches gives the same bits (no ordering left to chance)
    Zt2, slabs2, dsl2, TT2, dnsl2, datt2 = (torch.full_like(t, float("nan")) for t in (Zt, slabs, dslabs, TTf, dn_sl, dattf))
    capi.brgcn_fwd_tile(xw, XW, Fd, O, N, g, norm, attp, NB, basis, root, Zt2, slabs2)
    capi.brgcn_bwd_source_tile(gout, O, Fd, O, N, g, norm, attp, NB, basis, root, dsl2)
    capi.brgcn_bwd_edges_tile(xw, XW, Fd, O, N, R, g, norm, attp, NB, basis, gout, O, TT2, dnsl2, E, datt2)
    for a, b2, what in ((Zt, Zt2, "Z"), (slabs, slabs2, "slabs"), (dslabs, dsl2, "dx slabs"), (TTf, TT2, "TT"), (dn_sl, dnsl2, "dn"),
                        (dattf, datt2, "datt")):
        assert torch.equal(cpu(a), cpu(b2)), what


@pytest.mark.parametrize("S", [2, 9])
def test_rgcn_kernels_with_windows_wider_than_a_wavefront(S):
    """Context +-40 over dialogues of up to 110 utterances: in / out degrees up to 81, i.e. more edges per node than the 64
    lanes that hold a window's metadata -- the second pass of the window loops of the tile kernels (and, for two speakers,
    of the relation-space kernels) against the separate basis-space kernels."""
    from erc_amd import capi
    from erc_amd.cogmen import build_graph_tensors
    torch.manual_seed(11 + S)
    B, T, NB, Fd, O = 5, 110, 30, 200, 100
    R = 2 * S * S
    lengths = torch.tensor([110, 3, 97, 66, 81], device=DEV)
    spk = torch.randint(0, S, (B, T), device=DEV)
    g, _, _ = build_graph_tensors(lengths, spk, 40, 40, S)
    N, E = g["counts"].cpu().tolist()
    assert int((g["in_ptr"][1:N + 1] - g["in_ptr"][:N]).max()) > 64
    z = lambda *s: torch.zeros(*s, device=DEV)
    XW = Fd + O
    xw, norm = torch.randn(N, XW, device=DEV), torch.rand(E, device=DEV)
    comp, basis, root = torch.randn(R, NB, device=DEV) * 0.3, torch.randn(NB, Fd, O, device=DEV) * 0.1, torch.randn(Fd, O, device=DEV) * 0.1
    bias, gout = torch.randn(O, device=DEV) * 0.1, torch.randn(N, O, device=DEV) * 0.1
    cpu = lambda t: t.detach().cpu()
    close = lambda a, b, tol=3e-5: float((cpu(a) - cpu(b)).abs().max()) <= tol * max(1.0, float(cpu(b).abs().max()))
    # ---- separate basis-space kernels (the reference of this test)
    Z = z(N, NB * Fd)
    capi.brgcn_agg_fwd(xw, XW, Fd, N, g, norm, comp, NB, Z)
    out = z(N, O)
    capi.gemm_f32(Z, NB * Fd, 0, None, basis, O, 1, None, out, O, N, O, NB * Fd, bias=bias)
    capi.gemm_f32(xw, XW, 0, None, root, O, 1, None, out, O, N, O, Fd, accumulate=1)
    dZ = z(N, NB * Fd)
    capi.gemm_f32(gout, O, 0, None, basis, O, 0, None, dZ, NB * Fd, N, NB * Fd, O)
    dn, TT, datt = z(E), z(E, NB), z(R, NB)
    capi.brgcn_bwd_edges(xw, XW, Fd, N, R, g, norm, comp, NB, dZ, dn, TT, datt)
    U, basisT, dx = z(N, NB * O), z(NB * O, Fd), z(N, Fd)
    capi.brgcn_bwd_source(gout, O, O, N, g, norm, comp, NB, U)
    capi.transpose_batched(basis, NB, Fd, O, basisT)
    capi.gemm_f32(U, NB * O, 0, None, basisT, Fd, 1, None, dx, Fd, N, Fd, NB * O)
    capi.gemm_f32(gout, O, 0, None, root, O, 0, None, dx, Fd, N, Fd, O, accumulate=1)
    # ---- tile kernels
    Sg = capi.brgcn_fwd_tile_slabs()
    Zt, sl, out_t = z(N, NB * Fd), z(Sg, N, O), z(N, O)
    capi.poison_lds()
    capi.brgcn_fwd_tile(xw, XW, Fd, O, N, g, norm, comp, NB, basis, root, Zt, sl)
    capi.slab_reduce(sl, Sg, N * O, bias, O, 0, out_t, N * O)
    assert torch.equal(cpu(Zt), cpu(Z)) and close(out_t, out)
    dsl, dx_t = z(Sg, N, Fd), z(N, Fd)
    capi.brgcn_bwd_source_tile(gout, O, Fd, O, N, g, norm, comp, NB, basis, root, dsl)
    capi.slab_reduce(dsl, Sg, N * Fd, None, Fd, 4, dx_t, N * Fd)
    assert close(dx_t, dx)
    dnsl, TT_t, datt_t, dn_t = z(Sg, E), z(E, NB), z(R, NB), z(E)
    capi.brgcn_bwd_edges_tile(xw, XW, Fd, O, N, R, g, norm, comp, NB, basis, gout, O, TT_t, dnsl, E, datt_t)
    capi.slab_reduce(dnsl, Sg, E, None, 0, 0, dn_t, E)
    assert close(dn_t, dn) and close(TT_t, TT) and close(datt_t, datt, 1e-4)
    # ---- relation space (two speakers)
    if R <= capi.rrgcn_max_relations():
        Wr, WrT = z(R, Fd, O), z(R, O, Fd)
        capi.basis_compose(comp, basis, R, NB, Fd, O, Wr, WrT)
        Zr, out_r = z(N, R * Fd), z(N, O)
        capi.rrgcn_agg_fwd(xw, XW, Fd, N, R, g, norm, Zr)
        capi.gemm_f32(Zr, R * Fd, 0, None, Wr, O, 1, None, out_r, O, N, O, R * Fd, bias=bias)
        capi.gemm_f32(xw, XW, 0, None, root, O, 1, None, out_r, O, N, O, Fd, accumulate=1)
        assert close(out_r, out, 1e-4)
        dZr, dn_r = z(N, R * Fd), z(E)
        capi.gemm_f32(gout, O, 0, None, Wr, O, 0, None, dZr, R * Fd, N, R * Fd, O)
        capi.rrgcn_bwd_edges(xw, XW, Fd, N, R, g, dZr, dn_r)
        assert close(dn_r, dn, 1e-4)
        Ur, dx_r = z(N, R * O), z(N, Fd)
        capi.rrgcn_bwd_source(gout, O, O, N, R, g, norm, Ur)
        capi.gemm_f32(Ur, R * O, 0, None, WrT, Fd, 1, None, dx_r, Fd, N, Fd, R * O)
        capi.gemm_f32(gout, O, 0, None, root, O, 0, None, dx_r, Fd, N, Fd, O, accumulate=1)
        assert close(dx_r, dx, 1e-4)


def test_dgcn_relation_space_equals_basis_space():
    """Same module, same batch: RGCNConv in relation space (the default for two speakers) vs basis space."""
    from erc_amd.dgcn import DGCNModule
    batch = to_device(make_batch(8, dict(a=100, t=100, v=512), n_speakers=2, n_classes=6, min_len=5, max_len=60, seed=31), DEV)
    outs = []
    for rel in (True, False):
        torch.manual_seed(5)
        m = DGCNModule(2, input_size=712, hidden_size=200, n_classes=6)
        m.relation_space = rel
        m.finalize(DEV)
        m.train()
        m.drop_p, m.lstm.drop_p = 0.0, 0.0
        stats = m.loss_and_grads(batch).clone()
        outs.append((stats, m._last_ws["logits"].clone(), m.flat.grad.clone(), m))
    assert outs[0][3].relation_space and not outs[1][3].relation_space
    assert float((outs[0][1] - outs[1][1]).abs().max()) < 2e-5
    assert abs(float(outs[0][0][0] - outs[1][0][0])) < 1e-5
    for n in outs[0][3].flat.params:
        e = rel_err(outs[0][3].flat.g(n), outs[1][3].flat.g(n))
        assert e < 1e-3, (n, e)


def test_dgcn_fused_rgcn_equals_separate_kernels():
    """Same module, same MELD-shaped batch (9 speakers: basis space): the three RGCN tile launches + the BiLSTM on compact
    rows vs the separate kernels + the BiLSTM on padded rows."""
    from erc_amd.dgcn import DGCNModule
    batch = to_device(make_batch(8, MELD, n_speakers=9, n_classes=7, min_len=1, max_len=33, seed=41, force_max=True), DEV)
    outs = []
    for fused in (True, False):
        torch.manual_seed(6)
        m = DGCNModule(9, input_size=1242, hidden_size=200, n_classes=7)
        m.fused_rgcn_fwd = fused
        m.compact_lstm = fused          # and the BiLSTM on compact rows vs padded rows + gather / scatter
        m.finalize(DEV)
        assert not m.relation_space
        m.train()
        m.drop_p, m.lstm.drop_p = 0.0, 0.0
        stats = m.loss_and_grads(batch).clone().cpu()
        outs.append((stats, m._last_ws["logits"].clone().cpu(), m))
    assert float((outs[0][1] - outs[1][1]).abs().max()) < 2e-5
    assert abs(float(outs[0][0][0] - outs[1][0][0])) < 1e-5
    for n in outs[0][2].flat.params:
        e = rel_err(outs[0][2].flat.g(n).cpu(), outs[1][2].flat.g(n).cpu())
        assert e < 1e-3, (n, e)


@pytest.mark.parametrize("speakers,dims,C,B,max_len,weights,drop", [
    (9, dict(a=300, t=600, v=342), 7, 8, 33, False, 0.0),      # MELD shape, basis space (the tile launches' slabs)
    (9, dict(a=300, t=600, v=342), 7, 32, 33, False, 0.4),     # the benched batch, dropout on (same counter-based masks)
    (2, dict(a=100, t=100, v=512), 6, 6, 60, True, 0.0),       # two speakers: relation space (GEMM slabs), class weights
    (2, dict(a=100, t=100, v=512), 6, 3, 17, True, 0.4),       # N not a multiple of 16, a tile that spans three dialogues
    (2, dict(a=100, t=100, v=512), 4, 1, 1, False, 0.0),       # one utterance
], ids=["meld-b8", "meld-b32-dropout", "iemocap-weighted", "ragged-weighted-dropout", "one-utterance"])
def test_dgcn_fused_tail_equals_separate_kernels(speakers, dims, C, B, max_len, weights, drop, monkeypatch):
    """erc_dgcn_tail (RGCN slab sum .. GraphConv .. classifier .. cross entropy .. dXc / dAGG / dHc in one launch) against the
    nine launches it replaces: same module, same batch, same dropout counters."""
    from erc_amd import capi
    from erc_amd.dgcn import DGCNModule, IEMOCAP6_WEIGHTS
    from tests.util_cases import poison_lds_before
    poison_lds_before(monkeypatch, "dgcn_tail")
    batch = to_device(make_batch(B, dims, n_speakers=speakers, n_classes=C, min_len=1, max_len=max_len, seed=43,
                                 force_max=max_len > 1), DEV)
    cw = torch.tensor(IEMOCAP6_WEIGHTS, dtype=torch.float32, device=DEV) if weights else None
    outs = []
    for fused in (True, False):
        torch.manual_seed(8)
        m = DGCNModule(speakers, input_size=sum(dims.values()), hidden_size=200, n_classes=C)
        m.fused_tail = fused
        m.fused_edge_bwd = fused     # + the RGCN backward's slab sum and relation sums inside EdgeAtt's backward launch
        m.finalize(DEV)
        m.train()
        m.drop_p, m.lstm.drop_p = drop, 0.0
        calls = []
        orig = capi.dgcn_tail
        monkeypatch.setattr(capi, "dgcn_tail", lambda *a, **k: (calls.append(1), orig(*a, **k))[1])
        stats = m.loss_and_grads(batch, cw).clone().cpu()
        monkeypatch.setattr(capi, "dgcn_tail", orig)
        assert len(calls) == (1 if fused else 0)
        ws = m._last_ws
        outs.append((stats, {k: ws[k].clone().cpu() for k in ("logits", "Xc", "Hc", "AGG", "Zc", "dlogits", "dZc", "dXc", "dHc")}, m))
    (sa, a, ma), (sb, b, mb) = outs
    assert abs(float(sa[0] - sb[0])) < 1e-5 and float(sa[1]) == float(sb[1]) and abs(float(sa[2] - sb[2])) < 1e-4 * float(sb[2])
    assert torch.equal(a["Zc"] > 0, b["Zc"] > 0) or float(((a["Zc"] > 0) != (b["Zc"] > 0)).float().mean()) < 1e-3
    for k in a:
        assert rel_err(a[k], b[k]) < 2e-5, (k, rel_err(a[k], b[k]))
    for n in ma.flat.params:
        e = rel_err(ma.flat.g(n).cpu(), mb.flat.g(n).cpu())
        assert e < 1e-4, (n, e)


def _pair(case, compute="f32"):
    from oracle.dgcn import DGCNOracle
    from erc_amd.dgcn import DGCNModule
    torch.manual_seed(case["seed"])
    ctx = case.get("context", (10, 10))
    ref = DGCNOracle(case["S"], input_size=case["D"], hidden_size=200, n_classes=case["C"], context=ctx)
    mine = DGCNModule(case["S"], input_size=case["D"], hidden_size=200, n_classes=case["C"], compute=compute, context=ctx)
    mine.load_state_dict(ref.state_dict())
    mine.finalize(DEV)
    return ref, mine


MELD = dict(a=300, t=600, v=342)      # meld-mmgcn-7 (mmbase.py:80-88)


@pytest.mark.parametrize("case", [
    dict(B=4, lens=(2, 14), dims=dict(a=10, t=14, v=12), S=2, C=6, seed=1, weights=True),
    dict(B=6, lens=(1, 33), dims=MELD, S=9, C=7, seed=2, weights=False),                        # MELD dims, D=1242, R=162
    dict(B=8, lens=(20, 60), dims=dict(a=100, t=100, v=512), S=2, C=6, seed=3, weights=True),   # IEMOCAP dims
    # BASELINE.json configs[4]: the modality ablation of scripts/baseline.py:27-58 (mmbase.py:31,408-415) at MELD dims:
    # D = 300 / 600 / 342 / 900 -- other vector-width / alignment paths of the input GEMMs than atv's 1242
    dict(B=6, lens=(1, 33), dims=MELD, S=9, C=7, seed=12, weights=False, modality="a"),
    dict(B=6, lens=(1, 33), dims=MELD, S=9, C=7, seed=13, weights=False, modality="t"),
    dict(B=6, lens=(1, 33), dims=MELD, S=9, C=7, seed=14, weights=False, modality="v"),
    dict(B=6, lens=(1, 33), dims=MELD, S=9, C=7, seed=15, weights=False, modality="at"),
    # context wider than a wavefront has lanes: in / out degrees up to 66 (window loops of the graph kernels take a second pass)
    dict(B=3, lens=(80, 110), dims=dict(a=10, t=14, v=12), S=2, C=6, seed=22, weights=True, context=(40, 25)),
    dict(B=3, lens=(80, 110), dims=dict(a=10, t=14, v=12), S=9, C=7, seed=23, weights=False, context=(40, 25)),
    # the BENCHED shape (bench.py --module dgcn: BASELINE.json configs[4]): B = 32 dialogues, lengths 1-33, MELD dims, atv
    dict(B=32, lens=(1, 33), dims=MELD, S=9, C=7, seed=31, weights=False),
], ids=["tiny", "meld-1242", "iemocap-712", "meld-a-300", "meld-t-600", "meld-v-342", "meld-at-900", "wide-context-s2",
        "wide-context-s9", "meld-benched-b32"])
def test_dgcn_module_parity_vs_oracle(case):
    from oracle.dgcn import IEMOCAP6_WEIGHTS
    modality = case.get("modality", "atv")
    batch = make_batch(case["B"], case["dims"], n_speakers=case["S"], n_classes=case["C"], min_len=case["lens"][0],
                       max_len=case["lens"][1], seed=case["seed"], force_max=True, modality=modality)
    case = dict(case, D=sum(case["dims"][m] for m in modality))
    assert batch["input_tensor"].shape[2] == case["D"]
    ref, mine = _pair(case)
    w = torch.tensor(IEMOCAP6_WEIGHTS) if case["weights"] else None
    dbatch = to_device(batch, DEV)
    ref.eval(), mine.eval()
    with torch.no_grad():
        want, want_g = ref(**batch)
    got, got_g = mine(**dbatch)
    assert float((got.cpu() - want).abs().max()) < 1e-4
    assert float((got_g.cpu() - want_g).abs().max()) < 1e-4
    ref.train(), mine.train()
    for m in ref.modules():
        if isinstance(m, torch.nn.Dropout):
            m.p = 0.0
    ref.rnn.rnn.dropout = 0.0
    mine.drop_p, mine.lstm.drop_p = 0.0, 0.0
    logits, _ = ref(**batch)
    loss = F.cross_entropy(logits, batch["label"], weight=w)
    loss.backward()
    stats = mine.loss_and_grads(dbatch, w.to(DEV) if w is not None else None).cpu()
    assert abs(float(stats[0]) - float(loss.detach())) < 2e-5
    refp = dict(ref.named_parameters())
    errs = {n: rel_err(mine.flat.g(n).cpu(), refp[n].grad) for n in mine.flat.params}
    assert max(errs.values()) < 3e-3, sorted(errs.items(), key=lambda kv: -kv[1])[:5]
    assert [n for n, p in ref.named_parameters() if p.grad is None] == ["clf.emotion_att.lin.weight", "clf.emotion_att.lin.bias"]


def test_dgcn_train_steps_with_dropout_run():
    from erc_amd.dgcn import DGCNTrainer
    from erc_amd.params import ERCParams, Group
    p = ERCParams().from_args(["--dataset=meld-mmgcn-7", "--modality=atv", "--loss_weights=False"])
    p.optim = Group(name="Adam", lr=3e-4, weight_decay=0.0)
    tr = DGCNTrainer(p, DEV)
    batch = make_batch(8, p.dims(), n_speakers=9, n_classes=7, min_len=1, max_len=33, seed=4)
    losses = [float(tr.train_step(tr.prepare_batch(batch)).cpu()[0]) for _ in range(4)]
    assert all(math.isfinite(l) for l in losses) and losses[-1] < losses[0] + 0.5


@pytest.mark.parametrize("modality,B", [("atv", 6), ("a", 6), ("v", 6), ("atv", 32)], ids=["atv", "a", "v", "atv-benched-b32"])
def test_dgcn_bf16_feature_mode_vs_rounded_oracle(modality, B):
    """``--compute=bf16`` (what bench.py --module dgcn --dtype bf16 runs): the feature block is stored in bf16 and the
    layer-0 input weights of the BiLSTM are rounded to bf16 while they are staged.  The oracle is fed the SAME rounded
    operands, so what is left is accumulation order plus the bf16 rounding of the gate gradients inside the
    weight_ih_l0 weight-gradient product (8 significant bits): logits within 1e-3, gradients within 2 % of their scale
    (weight_ih_l0 itself: 3 %).  Tolerances stated here are those of the MODE, not of fp32 parity (1e-4, test above)."""
    D = sum(MELD[m] for m in modality)
    case = dict(B=B, lens=(1, 33), dims=MELD, S=9, C=7, seed=21, D=D)      # (B = 32: the shape bench.py --module dgcn --dtype bf16 runs)
    batch = make_batch(B, MELD, n_speakers=9, n_classes=7, min_len=1, max_len=33, seed=21, force_max=True, modality=modality)
    ref, mine = _pair(case, compute="bf16")
    with torch.no_grad():
        for n in ("weight_ih_l0", "weight_ih_l0_reverse"):
            w = getattr(ref.rnn.rnn, n)
            w.copy_(w.to(torch.bfloat16).float())
            mine.flat.w("rnn.rnn." + n).copy_(w.to(DEV))
    dbatch = to_device(batch, DEV)
    dbatch["input_tensor"] = dbatch["input_tensor"].to(torch.bfloat16)
    batch = dict(batch, input_tensor=batch["input_tensor"].to(torch.bfloat16).float())
    ref.eval(), mine.eval()
    with torch.no_grad():
        want, _ = ref(**batch)
    got, _ = mine(**dbatch)
    assert float((got.cpu() - want).abs().max()) < 1e-3
    ref.train(), mine.train()
    for m in ref.modules():
        if isinstance(m, torch.nn.Dropout):
            m.p = 0.0
    ref.rnn.rnn.dropout = 0.0
    mine.drop_p, mine.lstm.drop_p = 0.0, 0.0
    logits, _ = ref(**batch)
    loss = F.cross_entropy(logits, batch["label"])
    loss.backward()
    stats = mine.loss_and_grads(dbatch, None).cpu()
    assert abs(float(stats[0]) - float(loss.detach())) < 1e-3
    refp = dict(ref.named_parameters())
    errs = {n: rel_err(mine.flat.g(n).cpu(), refp[n].grad) for n in mine.flat.params}
    for n, e in errs.items():
        assert e < (3e-2 if "weight_ih_l0" in n else 2e-2), sorted(errs.items(), key=lambda kv: -kv[1])[:5]


@pytest.mark.parametrize("name", ["classifier_c6", "classifier_c7"])
def test_classifier_ops_vs_reference_golden(golden, name):
    """The classifier's launch sequence (DGCNModule: Linear+ReLU -> Linear, and its backward) against the REFERENCE's
    own dgcn_models.Classifier (golden vectors; eval mode: dropout off)."""
    from erc_amd import capi
    fx = golden(name)
    C, N = int(fx["n_classes"]), fx["h"].shape[0]
    clf = torch.nn.Module()
    clf.emotion_att = torch.nn.Module(); clf.emotion_att.lin = torch.nn.Linear(300, 300)
    clf.lin1, clf.lin2 = torch.nn.Linear(300, 100), torch.nn.Linear(100, C)
    fill_params(clf, int(fx["param_seed"]))
    W1, b1, W2, b2 = [t.detach().to(DEV) for t in (clf.lin1.weight, clf.lin1.bias, clf.lin2.weight, clf.lin2.bias)]
    h = torch.from_numpy(fx["h"]).to(DEV)
    Z, logits = torch.zeros(N, 100, device=DEV), torch.zeros(N, C, device=DEV)
    capi.gemm_f32(h, 300, 0, None, W1, 300, 0, None, Z, 100, N, 100, 300, bias=b1, act=1)
    capi.gemm_f32(Z, 100, 0, None, W2, 100, 0, None, logits, C, N, C, 100, bias=b2)
    np.testing.assert_allclose(logits.cpu().numpy(), fx["logits"], atol=2e-5, rtol=1e-5)
    dl = torch.from_numpy(fx["w"]).to(DEV)
    dZ, dh = torch.zeros(N, 100, device=DEV), torch.zeros(N, 300, device=DEV)
    capi.gemm_f32(dl, C, 0, None, W2, 100, 1, None, dZ, 100, N, 100, C, act=2, aux=Z, ldaux=100, act_scale=1.0)
    capi.gemm_f32(dZ, 100, 0, None, W1, 300, 1, None, dh, 300, N, 300, 100)
    np.testing.assert_allclose(dh.cpu().numpy(), fx["dh"], atol=2e-5, rtol=1e-4)
    dW1, db1 = torch.zeros(100, 300, device=DEV), torch.zeros(100, device=DEV)
    dW2, db2 = torch.zeros(C, 100, device=DEV), torch.zeros(C, device=DEV)
    capi.gemm_f32(dZ, 100, 1, None, h, 300, 1, None, dW1, 300, 100, 300, N, ones_col=1, bias_out=db1)
    capi.gemm_f32(dl, C, 1, None, Z, 100, 1, None, dW2, 100, C, 100, N, ones_col=1, bias_out=db2)
    check_grad_digest(fx, [("clf.lin1.weight", dW1), ("clf.lin1.bias", db1), ("clf.lin2.weight", dW2),
                           ("clf.lin2.bias", db2)], tol=1e-4)


def test_training_loop_exact_shape_graphs_capture_on_second_occurrence():
    """trainer.run for a module without capacity buckets (DialogueGCN), default sampling: a shape is captured when it shows
    up the second time -- its first step runs eagerly ON the static buffers the later graph binds to, so nothing (workspace,
    weight-gradient table) is allocated or uploaded while capturing -- and replayed from then on; per-step losses are
    identical to the eager loop.  Equal-length synthetic dialogues make the shapes repeat."""
    from tests.test_gpu_cogmen import _run_cli
    args = ["--module=dgcn", "--dataset=meld-mmgcn-7", "--loss_weights=False", "--epoch=2", "--n_train=20", "--n_test=4",
            "--train.batch_size=4", "--test.batch_size=4", "--syn_min_len=12", "--syn_max_len=12"]
    g_loss, g_ep = _run_cli(args)
    e_loss, e_ep = _run_cli(args + ["--graph_replay=False"])
    assert len(g_loss) == 10 and g_loss == e_loss
    assert g_ep[1]["graphs_captured"] == 1 and g_ep[1]["eager_steps"] == 2 and g_ep[1]["graph_replays"] == 8
    assert e_ep[1]["graph_replays"] == 0
