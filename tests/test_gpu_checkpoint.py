"""Checkpoint envelope (erc_amd/checkpoint.py) for all four trainers: save after two steps, load into a fresh trainer,
identical state and an identical third step (the path is deterministic, so equality is exact)."""
import importlib

import pytest
import torch

pytestmark = pytest.mark.gpu

CASES = {
    "cogmen": ("iemocap-cogmen-6", "COGMENTrainer", []),
    "dagerc": ("iemocap-cogmen-6", "DAGERCTrainer", ["--reimplement"]),
    "dgcn": ("meld-mmgcn-7", "DGCNTrainer", ["--loss_weights=False"]),
    "mmgcn": ("iemocap-cogmen-6", "MMGCNTrainer", []),
}


@pytest.mark.parametrize("module", sorted(CASES))
def test_save_load_continue(module, tmp_path):
    from bench import synthetic_batch
    from erc_amd import checkpoint
    ds, cls, extra = CASES[module]
    plugin = importlib.import_module("track_mm." + module)

    def make():
        params = plugin.ParamsType().from_args(["--dataset=" + ds, "--modality=atv"] + extra)
        return params, getattr(plugin, cls)(params, torch.device("cuda:0"))

    params, a = make()
    batches = [a.prepare_batch(synthetic_batch(params, 3, 12, seed=s)) for s in (1, 2, 3)]
    for b in batches[:2]:
        a.train_step(b)
    path = checkpoint.save(a, str(tmp_path / "m.ckpt"))
    _, b_tr = make()
    with torch.no_grad():      # make sure the load really overwrites
        for p in b_tr.model.flat.params.values():
            p.add_(1.0)
    checkpoint.load(b_tr, path)
    sa, sb = a.model.state_dict(), b_tr.model.state_dict()
    assert set(sa) == set(sb)
    for k in sa:
        if "num_batches_tracked" not in k:
            assert torch.equal(sa[k].cpu(), sb[k].cpu()), k
    assert torch.equal(a.model.flat.exp_avg, b_tr.model.flat.exp_avg)
    assert torch.equal(a.model.flat.exp_avg_sq, b_tr.model.flat.exp_avg_sq)
    assert int(a.optim.state[0]) == int(b_tr.optim.state[0]) == 2
    b_tr.optim.state[1] = a.optim.state[1]          # dropout stream position is not part of the reference's file
    b_tr.optim.state[2] = a.optim.state[2]
    la = a.train_step(batches[2]).cpu()[0]
    lb = b_tr.train_step(batches[2]).cpu()[0]
    assert float(la) == float(lb)
