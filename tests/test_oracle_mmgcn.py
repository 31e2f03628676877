"""CPU: the MMGCN oracle reproduces the reference's own MMGCNModule (golden vectors): normalised adjacency,
logits, loss, gradients, and the set of never-trained parameters."""
import numpy as np
import pytest
import torch
from torch.nn import functional as F

from oracle.mmgcn import MMGCNOracle
from tests.util_cases import check_grad_digest, fill_params


@pytest.mark.parametrize("name", ["mmgcn_atv", "mmgcn_tv_s3"])
def test_mmgcn_oracle_matches_reference(golden, name):
    fx = golden(name)
    batch = {k[3:]: torch.from_numpy(fx[k]) for k in fx.files if k.startswith("in_")}
    for k in ("text_feature", "audio_feature", "visual_feature"):
        batch.setdefault(k, None)
    da, dt, dv = [int(v) for v in fx["dims"]]
    model = MMGCNOracle(hidden_text=dt, hidden_visual=dv, hidden_audio=da, n_speakers=int(fx["n_speakers"]),
                        n_classes=int(fx["n_classes"]), modals=str(fx["modality"]))
    fill_params(model, int(fx["param_seed"]))
    model.eval()
    logits, _ = model(**batch)
    loss = F.cross_entropy(logits, batch["label"])
    loss.backward()
    np.testing.assert_allclose(model.graph_model.last_adj.detach().numpy(), fx["adj"], atol=2e-6, rtol=1e-5)
    np.testing.assert_allclose(logits.detach().numpy(), fx["logits"], atol=5e-6, rtol=1e-4)
    assert abs(float(loss) - float(fx["loss"])) < 1e-6
    none = sorted(n for n, p in model.named_parameters() if p.grad is None)
    assert none == sorted(fx["grad_none"].tolist())
    check_grad_digest(fx, [(n, p.grad) for n, p in model.named_parameters() if p.grad is not None], tol=2e-4)
