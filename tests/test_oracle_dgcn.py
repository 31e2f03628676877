"""CPU: the DialogueGCN oracle pieces reproduce the reference's own EdgeAtt + batch_graphify + vendored RGCNConv
(golden vectors)."""
import numpy as np
import pytest
import torch

from oracle import graph as og
from oracle.dgcn import EdgeAtt, RGCNConvBasis, dgcn_graphify
from tests.util_cases import check_grad_digest, fill_params


@pytest.mark.parametrize("name", ["dgcn_s2", "dgcn_s9"])
def test_dgcn_oracle_matches_reference(golden, name):
    fx = golden(name)
    S = int(fx["n_speakers"])
    lengths, spk = torch.from_numpy(fx["lengths"]), torch.from_numpy(fx["speakers"])
    feats = torch.from_numpy(fx["features"]).requires_grad_()
    att = EdgeAtt(200, 10, 10)
    fill_params(att, int(fx["att_seed"]))
    x, ei, en, et = dgcn_graphify(feats, lengths, spk, 10, 10, S, att)
    ei_s, et_s, en_s = og.canonical_edges(ei.numpy(), et.numpy(), en.detach().numpy())
    np.testing.assert_array_equal(ei_s, fx["edge_index"])
    np.testing.assert_array_equal(et_s, fx["edge_type"])
    np.testing.assert_allclose(en_s, fx["edge_norm"], atol=1e-6, rtol=1e-5)
    conv = RGCNConvBasis(200, 100, 2 * S * S, 30)
    fill_params(conv, int(fx["conv_seed"]))
    out = conv(x, ei, et, en)
    np.testing.assert_allclose(out.detach().numpy(), fx["rgcn_out"], atol=2e-5, rtol=1e-5)
    out.backward(torch.from_numpy(fx["gout"]))
    np.testing.assert_allclose(feats.grad.numpy(), fx["dfeatures"], atol=2e-5, rtol=1e-4)
    check_grad_digest(fx, [("edge_att.weight", att.weight.grad)] + [("conv1." + n, p.grad) for n, p in conv.named_parameters()],
                      tol=1e-4)
