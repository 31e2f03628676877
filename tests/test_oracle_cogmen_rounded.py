"""The bf16-product restatement of the COGMEN graph part (oracle/pyg.py RoundedLinear / RGCNMeanRounded) carries
hand-written backward formulas: with the rounding switched off they must reproduce autograd of the plain restatement
exactly, and with it on they must stay within bf16 distance of it."""
import pytest
import torch
from torch.nn import functional as F

from tests.util_cases import ZERO_GRAD, cogmen_case, rel_err


def _run(ref, batch):
    ref.train()
    for m in ref.modules():
        if isinstance(m, torch.nn.Dropout):
            m.p = 0.0
    ref.zero_grad()
    logits, _ = ref(**batch)
    F.cross_entropy(logits, batch["label"]).backward()
    return logits.detach(), {n: p.grad.clone() for n, p in ref.named_parameters() if p.grad is not None}


@pytest.mark.parametrize("identity", [True, False])
def test_rounded_graph_part_matches_autograd(monkeypatch, identity):
    from oracle import pyg
    from oracle.cogmen import COGMENOracle
    case = cogmen_case(B=5, min_len=3, max_len=30, dims=dict(a=12, t=20, v=16), seed=11)
    torch.manual_seed(4)
    plain = COGMENOracle(case["D"], 100, 17, 2, 6, dead_encoder=False)
    rounded = COGMENOracle(case["D"], 100, 17, 2, 6, dead_encoder=False, bf16_products=True)
    rounded.load_state_dict(plain.state_dict())
    if identity:
        monkeypatch.setattr(pyg, "rb", lambda t: t)
    want, gw = _run(plain, case["batch"])
    got, gg = _run(rounded, case["batch"])
    tol = 1e-5 if identity else 3e-2
    assert float((got - want).abs().max()) < tol
    assert set(gw) == set(gg)
    for n in gw:
        if n in ZERO_GRAD:      # mathematically zero (softmax shift invariance / constant shift in front of BatchNorm)
            continue
        if identity:
            assert rel_err(gg[n], gw[n]) < 2e-4, n
        else:   # quantisation of the mode itself: a norm-wise sanity bound, not a parity statement
            assert float((gg[n] - gw[n]).norm() / (gw[n].norm() + 1e-9)) < 0.1, n
