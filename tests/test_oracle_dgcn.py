"""CPU: the DialogueGCN oracle pieces reproduce the reference's own EdgeAtt + batch_graphify + vendored RGCNConv
(golden vectors)."""
import numpy as np
import pytest
import torch

from oracle import graph as og
from oracle.dgcn import Classifier, EdgeAtt, RGCNConvBasis, SeqContext, dgcn_graphify
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


@pytest.mark.parametrize("name", ["seqcontext_d30", "seqcontext_d1242"])
def test_seqcontext_oracle_matches_reference(golden, name):
    """oracle.dgcn.SeqContext against the reference's own SeqContext (dgcn_models.py:10-33; golden vectors)."""
    fx = golden(name)
    x = torch.from_numpy(fx["x"]).requires_grad_()
    rnn = SeqContext(x.shape[2], 200).eval()
    fill_params(rnn, int(fx["param_seed"]))
    out = rnn(torch.from_numpy(fx["lengths"]), x)
    np.testing.assert_allclose(out.detach().numpy(), fx["out"], atol=1e-6, rtol=1e-5)
    (out * torch.from_numpy(fx["w"])).sum().backward()
    np.testing.assert_allclose(x.grad.numpy(), fx["dx"], atol=1e-5, rtol=1e-4)
    check_grad_digest(fx, [("rnn." + n, p.grad) for n, p in rnn.named_parameters()], tol=1e-4)


@pytest.mark.parametrize("name", ["classifier_c6", "classifier_c7"])
def test_classifier_oracle_matches_reference(golden, name):
    """oracle.dgcn.Classifier against the reference's own Classifier (dgcn_models.py:155-170; golden vectors),
    including the set of parameters that never receive a gradient (emotion_att)."""
    fx = golden(name)
    h = torch.from_numpy(fx["h"]).requires_grad_()
    clf = Classifier(300, 100, int(fx["n_classes"]), 0.4).eval()
    fill_params(clf, int(fx["param_seed"]))
    logits = clf(h)
    np.testing.assert_allclose(logits.detach().numpy(), fx["logits"], atol=1e-6, rtol=1e-5)
    (logits * torch.from_numpy(fx["w"])).sum().backward()
    np.testing.assert_allclose(h.grad.numpy(), fx["dh"], atol=1e-6, rtol=1e-5)
    none = sorted(n for n, p in clf.named_parameters() if p.grad is None)
    assert none == sorted(str(n) for n in fx["grad_none"])
    check_grad_digest(fx, [("clf." + n, p.grad) for n, p in clf.named_parameters() if p.grad is not None], tol=1e-5)
