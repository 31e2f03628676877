"""Data-parallel SEMANTICS on one GPU (SURVEY.md 8e, Appendix B "Loss reduction under DP"): what N ranks compute is
the average over ranks of each rank's own mean-loss gradient (DDP), with BatchNorm batch statistics taken per rank
(plain nn.BatchNorm1d under DDP, cogmen.py:67) and DAG-ERC's clip_grad_norm_(5) applied to the AVERAGED gradient
(dagerc.py:230 runs after accelerate.backward).  Two shards are run one after the other through the HIP path, their flat
gradients are summed (the all-reduce) and scaled by 1/2 inside the optimizer (``grad_scale``) exactly as
``engine.all_reduce_grads`` + ``FusedAdam.step`` do across ranks, and the result is held against the oracle doing the
same thing with autograd.  (The collective itself is covered by tests/test_dp_gloo.py on CPU.)"""
import pytest
import torch
from torch.nn import functional as F

from tests.util_cases import ZERO_GRAD, cogmen_case, make_batch, rel_err, to_device

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def test_cogmen_two_shards_equal_ddp_average():
    from oracle.cogmen import COGMENOracle
    from erc_amd.cogmen import COGMENModule
    from erc_amd.engine import FusedAdam
    shards = [cogmen_case(B=5, min_len=3, max_len=30, dims=dict(a=100, t=100, v=512), seed=60 + r)["batch"] for r in range(2)]
    torch.manual_seed(2)
    ref = COGMENOracle(712, 100, 17, 2, 6, dead_encoder=False)
    with torch.no_grad():
        ref.gcn.bn.weight.uniform_(0.5, 1.5), ref.gcn.bn.bias.uniform_(-0.3, 0.3)
    for m in ref.modules():
        if isinstance(m, torch.nn.Dropout):
            m.p = 0.0
    mine = COGMENModule(712, 100, 17, 2, 6)
    mine.load_state_dict(ref.state_dict())
    mine.finalize(DEV)
    mine.drop_p = 0.0
    ref.train(), mine.train()
    # oracle: each "rank" forwards its own shard (own BatchNorm batch statistics, own mean loss); DDP averages gradients
    ref.zero_grad()
    losses = []
    for b in shards:
        logits, _ = ref(**b)
        loss = F.cross_entropy(logits, b["label"])
        (loss / len(shards)).backward()
        losses.append(float(loss))
    # HIP: shard gradients summed (= all-reduce), 1/world folded into the optimizer
    total = torch.zeros_like(mine.flat.grad)
    for r, b in enumerate(shards):
        stats = mine.loss_and_grads(to_device(b, DEV)).cpu()
        assert abs(float(stats[0]) - losses[r]) < 2e-5
        total += mine.flat.grad
    refp = dict(ref.named_parameters())
    mine.flat.grad.copy_(total)
    for n in mine.flat.params:
        want = refp[n].grad
        if float(want.abs().max()) < 1e-6:
            continue
        assert rel_err(0.5 * mine.flat.g(n).cpu(), want) < 2e-3, n
    # one optimizer step on the averaged gradient == torch.optim.Adam on the oracle's averaged gradient
    opt_ref = torch.optim.Adam(ref.parameters(), lr=1e-3, weight_decay=1e-8)
    opt_ref.step()
    FusedAdam(mine.flat, lr=1e-3, weight_decay=1e-8).step(grad_scale=0.5)
    for n in mine.flat.params:
        if n in ZERO_GRAD:      # mathematically zero gradients (shift in front of softmax / BatchNorm): Adam turns their
            continue            # rounding noise into +-lr steps on both sides
        assert float((mine.flat.w(n).cpu() - refp[n].detach()).abs().max()) < 2e-5, n


def test_dagerc_two_shards_clip_on_averaged_gradient():
    from oracle.dagerc import DAGERCOracle, dagerc_loss
    from erc_amd.dagerc import DAGERCModule
    from erc_amd.engine import FusedAdam
    dims = dict(a=20, t=30, v=14)
    shards = [make_batch(3, dims, n_speakers=2, n_classes=6, min_len=2, max_len=15, seed=70 + r, speaker_onehot=True,
                         force_max=True) for r in range(2)]
    torch.manual_seed(4)
    ref = DAGERCOracle(emb_dim=64, dropout=0.0, n_classes=6, gnn_layers=4)
    with torch.no_grad():                       # large enough weights that the global norm exceeds the clip threshold
        for q in ref.out_mlp.parameters():
            q.mul_(6.0)
    mine = DAGERCModule(emb_dim=64, dropout=0.0, n_classes=6, gnn_layers=4)
    mine.load_state_dict(ref.state_dict())
    mine.finalize(DEV)
    ref.train(), mine.train()
    ref.zero_grad()
    for b in shards:
        loss, _ = dagerc_loss(ref, b)
        (loss / 2).backward()
    live = [q for q in ref.parameters() if q.grad is not None]
    norm = float(torch.nn.utils.clip_grad_norm_(live, 5.0))
    assert norm > 5.0, "the case must actually clip (norm %.2f)" % norm
    torch.optim.AdamW(live, lr=1e-3, weight_decay=1e-2).step()
    total = torch.zeros_like(mine.flat.grad)
    for b in shards:
        mine.loss_and_grads(to_device(b, DEV))
        total += mine.flat.grad
    mine.flat.grad.copy_(total)
    opt = FusedAdam(mine.flat, lr=1e-3, weight_decay=1e-2, decoupled=True, clip_norm=5.0)
    opt.step(grad_scale=0.5)
    assert abs(float(opt.gnorm.cpu()) - norm) < 1e-3 * norm      # the norm of the AVERAGED gradient
    refp = dict(ref.named_parameters())
    for n in mine.flat.params:
        # the first AdamW step moves every element by lr * g / (|g| + eps'): where |g| is at rounding-noise level the
        # step's SIZE is noise on both sides, so parameters are compared where the gradient is well above that level
        g = refp[n].grad
        if n.endswith("linear.bias") and n.startswith("gather."):
            assert float(g.abs().max()) < 1e-6      # softmax is shift invariant: this gradient is mathematically zero
            continue
        sure = g.abs() > 1e-3 * g.abs().max()
        d = (mine.flat.w(n).cpu() - refp[n].detach()).abs()
        assert float(d[sure].max()) < 2e-5, n
        assert float(d.max()) < 2.1e-3, n            # nowhere more than the two opposite full steps
    mine.check_cluster()
