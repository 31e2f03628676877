"""MMGCN on the GPU vs (a) golden vectors produced by the REFERENCE's own MMGCNModule and (b) the reference-pinned
CPU oracle at larger shapes (eval mode: every dropout off)."""
import numpy as np
import pytest
import torch
from torch.nn import functional as F

from tests.util_cases import poison_lds_before, check_grad_digest, fill_params, make_batch, rel_err, to_device

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _dense_adj(ws, B, Mo, N, lens):
    """expand blocks + cross entries to the reference's (M*N)^2 matrix."""
    P = ws["P"]
    ADJ, CR = ws["ADJ"].cpu(), ws["CR"].cpu()
    A = torch.zeros(Mo * N, Mo * N)
    off = 0
    for b, L in enumerate(lens):
        for m in range(Mo):
            A[m * N + off:m * N + off + L, m * N + off:m * N + off + L] = ADJ[b * Mo + m, :L, :L]
            for n in range(Mo):
                if n != m:
                    idx = torch.arange(L)
                    A[m * N + off + idx, n * N + off + idx] = CR[b, m * Mo + n, :L]
        off += L
    return A


@pytest.mark.parametrize("name", ["mmgcn_atv", "mmgcn_tv_s3"])
def test_mmgcn_matches_reference_golden(golden, name):
    from erc_amd.mmgcn import MMGCNModule
    fx = golden(name)
    batch = {k[3:]: torch.from_numpy(fx[k]) for k in fx.files if k.startswith("in_")}
    da, dt, dv = [int(v) for v in fx["dims"]]
    mods = str(fx["modality"])
    model = MMGCNModule(hidden_text=dt, hidden_visual=dv, hidden_audio=da, n_speakers=int(fx["n_speakers"]),
                        n_classes=int(fx["n_classes"]), modals=mods)
    fill_params(model, int(fx["param_seed"]))
    model.finalize(DEV)
    model.eval()
    dbatch = to_device(batch, DEV)
    for k in ("text_feature", "audio_feature", "visual_feature"):
        dbatch.setdefault(k, None)
    stats = model.loss_and_grads(dbatch).cpu()
    T, B = batch["speaker_tensor"].shape[:2]
    N = int(batch["label"].shape[0])
    ws = model._last_ws
    A = _dense_adj(ws, B, len(mods), N, batch["text_length"].tolist())
    np.testing.assert_allclose(A.numpy(), fx["adj"], atol=2e-5, rtol=1e-4)
    assert float((ws["logits"].cpu() - torch.from_numpy(fx["logits"])).abs().max()) < 1e-4
    assert abs(float(stats[0]) - float(fx["loss"])) < 1e-5
    check_grad_digest(fx, [(n, model.flat.g(n)) for n in model.flat.params], tol=5e-3)
    assert set(model.flat.params).isdisjoint(set(fx["grad_none"].tolist()))
    assert len(model.flat.params) + len(fx["grad_none"]) == len(list(model.named_parameters()))


@pytest.mark.parametrize("B,lens,dims,S,C,mods", [(16, (20, 110), dict(a=100, t=768, v=512), 2, 6, "atv"),
                                                  (5, (1, 30), dict(a=30, t=60, v=34), 9, 7, "at")])
def test_mmgcn_parity_vs_oracle_large(B, lens, dims, S, C, mods, monkeypatch):
    """BASELINE config-3 shape (iemocap-cogmen-sbert-6 atv, B=16, T=110) and a ragged two-modality MELD-like case."""
    from oracle.mmgcn import MMGCNOracle
    from erc_amd.mmgcn import MMGCNModule
    batch = make_batch(B, dims, n_speakers=S, n_classes=C, min_len=lens[0], max_len=lens[1], seed=9, modality=mods,
                       batch_first=False, speaker_onehot=True, force_max=True)
    torch.manual_seed(4)
    ref = MMGCNOracle(hidden_text=dims["t"], hidden_visual=dims["v"], hidden_audio=dims["a"], n_speakers=S, n_classes=C,
                      modals=mods)
    mine = MMGCNModule(hidden_text=dims["t"], hidden_visual=dims["v"], hidden_audio=dims["a"], n_speakers=S, n_classes=C,
                       modals=mods)
    mine.load_state_dict(ref.state_dict())
    mine.finalize(DEV)
    ref.eval(), mine.eval()
    torch.set_num_threads(8)
    logits, _ = ref(**batch)
    loss = F.cross_entropy(logits, batch["label"])
    loss.backward()
    poison_lds_before(monkeypatch, "gcnii_chain_fwd", "gcnii_chain_bwd")       # uninitialised LDS shows up as NaN, on every box
    stats = mine.loss_and_grads(to_device(batch, DEV)).cpu()
    T = batch["speaker_tensor"].shape[0]
    got = mine._last_ws["logits"].cpu()
    assert float((got - logits.detach()).abs().max()) < 1e-4
    assert abs(float(stats[0]) - float(loss.detach())) < 1e-5
    refp = dict(ref.named_parameters())
    errs = {n: rel_err(mine.flat.g(n).cpu(), refp[n].grad) for n in mine.flat.params}
    assert max(errs.values()) < 5e-3, sorted(errs.items(), key=lambda kv: -kv[1])[:6]


def test_mmgcn_train_steps_with_dropout_run():
    import math
    from erc_amd.mmgcn import MMGCNTrainer
    from erc_amd.params import ERCParams, Group
    p = ERCParams().from_args(["--dataset=iemocap-cogmen-6", "--modality=atv"])
    p.optim = Group(name="Adam", lr=3e-4, weight_decay=3e-5)
    p.batch_first, p.speaker_onehot = False, True
    tr = MMGCNTrainer(p, DEV)
    batch = make_batch(4, p.dims(), n_classes=6, min_len=5, max_len=20, seed=6, batch_first=False, speaker_onehot=True)
    losses = [float(tr.train_step(tr.prepare_batch(batch)).cpu()[0]) for _ in range(3)]
    assert all(math.isfinite(l) for l in losses)


def _mmgcn_trainer():
    from erc_amd.mmgcn import MMGCNTrainer
    from erc_amd.params import ERCParams, Group
    p = ERCParams().from_args(["--dataset=iemocap-cogmen-6", "--modality=atv"])
    p.optim = Group(name="Adam", lr=3e-4, weight_decay=3e-5)
    p.batch_first, p.speaker_onehot = False, True
    tr = MMGCNTrainer(p, DEV)
    batch = tr.prepare_batch(make_batch(4, p.dims(), n_classes=6, min_len=17, max_len=40, seed=6, batch_first=False,
                                        speaker_onehot=True))
    return tr, batch


def test_chain_timeout_protocol():
    """health word raised in the middle of a step -> update skipped on the device, next step counts + clears, reported once"""
    from tests.test_gpu_dagerc import _timeout_protocol
    _timeout_protocol(*_mmgcn_trainer())


def test_chain_poll_timeout_fails_the_step_on_the_device():
    """The real thing: with the poll bound at 1 the GCNII chain kernels (several workgroups per dialogue and modality that
    exchange rows every layer) run into it, raise the health word themselves and drain; the optimizer launch of that step
    leaves parameters, moments and the step count alone; check_cluster() reports it; the next step, with the default
    bound, trains."""
    from erc_amd import capi
    tr, batch = _mmgcn_trainer()
    tr.train_step(batch)
    before, m_before, step = tr.model.flat.data.clone(), tr.model.flat.exp_avg.clone(), int(tr.optim.state[0])
    assert tr.model._last_ws["chain"] and step == 1
    capi.gcnii_chain_set_spin_limit(1)
    try:
        tr.train_step(batch)
        torch.cuda.synchronize()
    finally:
        capi.gcnii_chain_set_spin_limit(0)
    assert int(tr.model.flat.health[0]) == capi.HEALTH_RAISED
    assert torch.equal(tr.model.flat.data, before) and torch.equal(tr.model.flat.exp_avg, m_before)
    assert int(tr.optim.state[0]) == step
    with pytest.raises(capi.ErcGraftError, match="GCNII chain"):
        tr.model.check_cluster()
    tr.train_step(batch)
    assert int(tr.optim.state[0]) == step + 1 and int(tr.model.flat.health[0]) == 0
    tr.model.check_cluster()
