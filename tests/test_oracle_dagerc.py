"""CPU: the DAG-ERC oracle reproduces the reference's own DAGERCModule (golden vectors), and the closed-form
predecessor rule that the HIP kernels implement reproduces the reference adjacency."""
import numpy as np
import pytest
import torch

from oracle import graph as og
from oracle.dagerc import DAGERCOracle, dagerc_loss
from tests.util_cases import check_grad_digest, fill_params


@pytest.mark.parametrize("name", ["dagerc_small", "dagerc_s3"])
def test_dagerc_oracle_matches_reference(golden, name):
    fx = golden(name)
    batch = {k[3:]: torch.from_numpy(fx[k]) for k in fx.files if k.startswith("in_")}
    D, C = int(fx["dims"].sum()), int(fx["n_classes"])
    model = DAGERCOracle(emb_dim=D, dropout=0.0, n_classes=C, gnn_layers=4)
    fill_params(model, int(fx["param_seed"]))
    model.train()
    loss, _ = dagerc_loss(model, batch)
    loss.backward()
    logits, _ = model(**batch)
    np.testing.assert_array_equal(model.last_adj.numpy(), fx["adj"])
    np.testing.assert_array_equal(model.last_s_mask.numpy(), fx["s_mask"])
    np.testing.assert_allclose(logits.detach().numpy(), fx["logits"], atol=2e-6, rtol=1e-5)
    assert abs(float(loss) - float(fx["loss"])) < 1e-6
    none = sorted(n for n, p in model.named_parameters() if p.grad is None)
    assert none == sorted(fx["grad_none"].tolist())
    assert all(n.startswith(("fcs.", "attentive_node_features.")) for n in none) and len(none) == 10
    check_grad_digest(fx, [(n, p.grad) for n, p in model.named_parameters() if p.grad is not None], tol=1e-4)


@pytest.mark.parametrize("name", ["dagerc_small", "dagerc_s3"])
def test_dag_predecessor_closed_form(golden, name):
    """adj row i = ones on [max(p_i,0), i-1] with p_i the last earlier utterance of the same speaker
    (SURVEY.md Appendix C); padded positions count as speaker 0."""
    fx = golden(name)
    spk = fx["in_speaker_tensor"].argmax(-1)
    p = og.dag_pred_closed_form(spk)
    B, T = spk.shape
    adj = np.zeros((B, T, T), dtype=np.float32)
    for b in range(B):
        for i in range(T):
            adj[b, i, max(p[b, i], 0):i] = 1
    np.testing.assert_array_equal(adj, fx["adj"])
    np.testing.assert_array_equal((spk[:, :, None] == spk[:, None, :]).astype(np.int64), fx["s_mask"])
