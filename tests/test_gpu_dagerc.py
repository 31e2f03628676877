"""DAG-ERC on the GPU vs (a) golden vectors produced by the REFERENCE's own DAGERCModule and (b) the CPU oracle
at larger shapes.  Integer structure (predecessors / speaker ids) bit-exact; logits within 1e-4."""
import numpy as np
import pytest
import torch

from tests.util_cases import poison_lds_before, check_grad_digest, fill_params, make_batch, rel_err, to_device

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _load(golden, name, compute="f32"):
    from erc_amd.dagerc import DAGERCModule
    fx = golden(name)
    batch = {k[3:]: torch.from_numpy(fx[k]) for k in fx.files if k.startswith("in_")}
    D, C = int(fx["dims"].sum()), int(fx["n_classes"])
    model = DAGERCModule(emb_dim=D, dropout=0.0, n_classes=C, gnn_layers=4, compute=compute)
    fill_params(model, int(fx["param_seed"]))
    model.finalize(DEV)
    return fx, batch, model


@pytest.mark.parametrize("name", ["dagerc_small", "dagerc_s3"])
def test_dag_structure_bit_exact_vs_reference(golden, name):
    fx, batch, model = _load(golden, name)
    model.eval()
    model(**to_device(batch, DEV))
    B, T = batch["input_tensor"].shape[:2]
    ws = model._last_ws
    spk, pred = ws["spk"].cpu().numpy(), ws["pred"].cpu().numpy()
    adj = np.zeros((B, T, T), dtype=np.float32)
    for b in range(B):
        for i in range(T):
            adj[b, i, max(pred[b, i], 0):i] = 1
    np.testing.assert_array_equal(adj, fx["adj"])                                  # get_adj_v1
    np.testing.assert_array_equal((spk[:, :, None] == spk[:, None, :]).astype(np.int64), fx["s_mask"])  # get_s_mask
    rows = np.concatenate([b * T + np.arange(L) for b, L in enumerate(batch["text_length"].tolist())])
    np.testing.assert_array_equal(ws["node_row"].cpu().numpy(), rows)


@pytest.mark.parametrize("name", ["dagerc_small", "dagerc_s3"])
def test_dagerc_matches_reference_golden(golden, name):
    """logits (all B*T padded rows), masked CE loss and every live gradient vs the reference's own module."""
    fx, batch, model = _load(golden, name)
    model.train()
    stats = model.loss_and_grads(to_device(batch, DEV)).cpu()
    B, T = batch["input_tensor"].shape[:2]
    logits = model._last_ws["logits"].view(B, T, -1).cpu()
    assert float((logits - torch.from_numpy(fx["logits"])).abs().max()) < 1e-4
    assert abs(float(stats[0]) - float(fx["loss"])) < 1e-5
    worst = check_grad_digest(fx, [(n, model.flat.g(n)) for n in model.flat.params], tol=2e-3)
    live = set(model.flat.params)
    assert live.isdisjoint(set(fx["grad_none"].tolist()))            # never-trained parameters stay out of the flat buffer
    assert len(live) + len(fx["grad_none"]) == len(list(model.named_parameters()))
    print("worst relative gradient error", worst)


@pytest.mark.parametrize("B,lens,dims,S,C", [(16, (20, 110), dict(a=100, t=100, v=512), 2, 6),
                                             (5, (1, 40), dict(a=30, t=60, v=34), 9, 7)])
def test_dagerc_parity_vs_oracle_large(B, lens, dims, S, C, monkeypatch):
    """BASELINE config-4 shape (B=16, T=110, D=712) and a multi-speaker ragged case vs the (reference-pinned) oracle."""
    from oracle.dagerc import DAGERCOracle, dagerc_loss
    from erc_amd.dagerc import DAGERCModule
    batch = make_batch(B, dims, n_speakers=S, n_classes=C, min_len=lens[0], max_len=lens[1], seed=8,
                       speaker_onehot=True, force_max=True)
    D = sum(dims.values())
    torch.manual_seed(5)
    ref = DAGERCOracle(emb_dim=D, dropout=0.0, n_classes=C, gnn_layers=4)
    mine = DAGERCModule(emb_dim=D, dropout=0.0, n_classes=C, gnn_layers=4)
    mine.load_state_dict(ref.state_dict())
    mine.finalize(DEV)
    ref.train(), mine.train()
    torch.set_num_threads(8)
    loss, _ = dagerc_loss(ref, batch)
    loss.backward()
    want, _ = ref(**batch)
    poison_lds_before(monkeypatch, "dag_rec_fwd", "dag_rec_bwd")       # uninitialised LDS shows up as NaN, on every box
    stats = mine.loss_and_grads(to_device(batch, DEV)).cpu()
    T = batch["input_tensor"].shape[1]
    got = mine._last_ws["logits"].view(B, T, -1).cpu()
    valid = batch["attention_mask"].bool()
    assert float((got[valid] - want.detach()[valid]).abs().max()) < 1e-4
    assert float((got - want.detach()).abs().max()) < 1e-4              # padded rows too (finite garbage, same garbage)
    assert abs(float(stats[0]) - float(loss)) < 1e-5
    refp = dict(ref.named_parameters())
    for n in mine.flat.params:
        assert rel_err(mine.flat.g(n).cpu(), refp[n].grad) < 2e-3, n
    mine.check_cluster()      # no member of a cluster-mode recurrence kernel timed out


def test_recurrence_configurations_agree(monkeypatch):
    """The weight-stationary recurrence kernels under different partitions -- elements per workgroup (EPC: 5 / 4 / 2 ->
    60 / 75 / 150 workgroups per group and layer), dialogues per group (DG: the MFMA M rows in use; a ragged last group;
    several launches when the groups do not fit the device at once) and, in the forward, layers per launch (the layer
    pipeline: 1 = one layer per launch, 3 = layers {0,1,2} then {3}, 4 = all four concurrently): hidden states and
    gradients equal up to the association order of the cross-workgroup / cross-wavefront partial sums."""
    from erc_amd.dagerc import DAGERCModule
    dims = dict(a=30, t=60, v=34)
    batch = make_batch(6, dims, n_speakers=3, n_classes=5, min_len=2, max_len=37, seed=4, speaker_onehot=True, force_max=True)
    res = {}
    # (forward epc, dg, layers per launch | backward epc, dg, layers per launch); 0 = let erc_dag_rec_config choose
    configs = [(0, 0, 0, 0, 0, 0), (5, 1, 1, 5, 1, 1), (5, 4, 4, 5, 4, 4), (4, 3, 3, 4, 3, 3), (4, 6, 2, 4, 6, 2),
               (2, 6, 1, 2, 6, 1), (2, 16, 1, 2, 16, 1), (5, 16, 4, 5, 16, 4), (5, 6, 2, 5, 2, 3)]
    for cf in configs:
        for k, v in zip(("ERC_DAG_EPC", "ERC_DAG_DG", "ERC_DAG_LPL", "ERC_DAG_BEPC", "ERC_DAG_BDG", "ERC_DAG_BLPL"), cf):
            monkeypatch.setenv(k, str(v))
        torch.manual_seed(9)
        m = DAGERCModule(emb_dim=sum(dims.values()), dropout=0.0, n_classes=5, gnn_layers=4).finalize(DEV)
        m.train()
        stats = m.loss_and_grads(to_device(batch, DEV)).cpu()
        ws = m._last_ws
        if cf[0]:
            assert ws["cfg"][0][:2] == cf[:2] and ws["cfg"][0][3] == cf[2], ws["cfg"]
            assert ws["cfg"][1][:2] == cf[3:5] and ws["cfg"][1][3] == cf[5], ws["cfg"]
        m.check_cluster()
        res[cf] = (float(stats[0]), ws["Hall"].cpu().clone(), m.flat.grad.cpu().clone(), ws["cfg"])
    base = res[configs[0]]
    print("default configuration (epc, dg, groups per launch, layers per launch) forward | backward:", base[3])
    for key in configs[1:]:
        assert abs(res[key][0] - base[0]) < 1e-6, key
        assert float((res[key][1] - base[1]).abs().max()) <= 2e-6 * max(1.0, float(base[1].abs().max())), key
        assert float((res[key][2] - base[2]).abs().max()) <= 2e-6 * max(1.0, float(base[2].abs().max())), key


def _timeout_protocol(tr, batch):
    """What the trainer sees when a persistent kernel's bounded poll times out in the middle of a step (the kernels store
    ERC_HEALTH_RAISED into the health word, which lives behind the flat gradient): that step's update is skipped on the
    device -- parameters, moments and the step counter untouched, no host synchronisation involved --, the NEXT step
    counts the event, clears the word and trains normally, and check_cluster() reports the count once."""
    from erc_amd import capi
    tr.train_step(batch)
    before, step = tr.model.flat.data.clone(), int(tr.optim.state[0])
    assert step == 1
    tr.model.train()
    tr.model.loss_and_grads(batch)
    tr.model.flat.health.fill_(capi.HEALTH_RAISED)         # raised between the step's first launch and its optimizer launch
    tr.optim.step()
    assert torch.equal(tr.model.flat.data, before) and int(tr.optim.state[0]) == step
    tr.train_step(batch)                                   # event counted and word cleared on the device: training continues
    assert not torch.equal(tr.model.flat.data, before) and int(tr.optim.state[0]) == step + 1
    assert int(tr.model.flat.health[0]) == 0 and int(tr.model.flat.events[0]) == 1
    with pytest.raises(capi.ErcGraftError, match="1 optimizer step"):
        tr.model.check_cluster()
    tr.model.check_cluster()                               # reported once
    tr.model.flat.health.fill_(capi.HEALTH_RAISED)         # raised by an evaluation pass: reported without a roll
    with pytest.raises(capi.ErcGraftError):
        tr.model.check_cluster()


def test_recurrence_timeout_fails_the_step_on_the_device():
    from erc_amd.dagerc import DAGERCTrainer
    from erc_amd.params import ERCParams
    p = ERCParams().from_args(["--dataset=iemocap-cogmen-6", "--modality=a"])
    p.speaker_onehot, p.dropout = True, 0.0
    tr = DAGERCTrainer(p, DEV)
    batch = tr.prepare_batch(make_batch(3, p.dims(), n_classes=6, min_len=2, max_len=9, seed=3, modality="a", speaker_onehot=True))
    _timeout_protocol(tr, batch)


def test_dagerc_train_step_clip_adamw():
    from oracle.dagerc import DAGERCOracle, dagerc_train_step
    from erc_amd.dagerc import DAGERCTrainer
    from erc_amd.params import ERCParams, Group
    p = ERCParams().from_args(["--dataset=iemocap-cogmen-6", "--modality=at"])
    p.optim = Group(name="AdamW", lr=1e-3, weight_decay=1e-2)
    p.speaker_onehot, p.dropout = True, 0.0
    tr = DAGERCTrainer(p, DEV)
    ref = DAGERCOracle(emb_dim=p.hidden_all, dropout=0.0, n_classes=p.n_classes)
    ref.load_state_dict({k: v.cpu() for k, v in tr.model.state_dict().items()})
    opt = torch.optim.AdamW([q for q in ref.parameters()], lr=1e-3)
    ref.train()
    for step in range(2):
        batch = make_batch(3, p.dims(), n_classes=6, min_len=2, max_len=12, seed=30 + step, modality="at",
                           speaker_onehot=True)
        loss, _ = dagerc_train_step(ref, opt, batch)
        stats = tr.train_step(tr.prepare_batch(batch)).cpu()
        assert abs(float(stats[0]) - float(loss)) < 2e-5
    refp = dict(ref.named_parameters())
    for n in tr.model.flat.params:
        assert float((tr.model.flat.w(n).cpu() - refp[n].detach()).abs().max()) < 2e-4, n


@pytest.mark.parametrize("B,lens,dims,S,C", [(4, (5, 40), dict(a=100, t=100, v=512), 2, 6),
                                             # the BENCHED shape (bench.py --module dagerc: BASELINE.json configs[3]): B = 16, T = 110, D = 712
                                             (16, (20, 110), dict(a=100, t=100, v=512), 2, 6)], ids=["b4", "benched-b16-t110"])
def test_dagerc_bf16_feature_mode_vs_rounded_oracle(B, lens, dims, S, C):
    """``--compute=bf16`` (what bench.py --module dagerc --dtype bf16 runs): the feature block is stored in bf16; the two
    weights that multiply it (fc1.weight, the raw-feature columns of out_mlp.0.weight) are rounded to bf16 while staged.
    The oracle gets the SAME rounded operands; left over: accumulation order and the bf16 rounding of the upstream
    gradient inside those two weight-gradient products.  Logits within 1e-3, gradients within 2 % of their scale
    (3 % for the two bf16-side weight gradients) -- tolerances of the MODE; fp32 parity (1e-4) is tested above."""
    from oracle.dagerc import DAGERCOracle, dagerc_loss
    from erc_amd.dagerc import DAGERCModule, HID
    batch = make_batch(B, dims, n_speakers=S, n_classes=C, min_len=lens[0], max_len=lens[1], seed=18,
                       speaker_onehot=True, force_max=True)
    D = sum(dims.values())
    torch.manual_seed(6)
    ref = DAGERCOracle(emb_dim=D, dropout=0.0, n_classes=C, gnn_layers=4)
    W5 = HID * 5
    with torch.no_grad():
        ref.fc1.weight.copy_(ref.fc1.weight.to(torch.bfloat16).float())
        w0 = ref.out_mlp[0].weight
        w0[:, W5:] = w0[:, W5:].to(torch.bfloat16).float()
    mine = DAGERCModule(emb_dim=D, dropout=0.0, n_classes=C, gnn_layers=4, compute="bf16")
    mine.load_state_dict(ref.state_dict())
    mine.finalize(DEV)
    dbatch = to_device(batch, DEV)
    dbatch["input_tensor"] = dbatch["input_tensor"].to(torch.bfloat16)
    batch = dict(batch, input_tensor=batch["input_tensor"].to(torch.bfloat16).float())
    ref.train(), mine.train()
    loss, _ = dagerc_loss(ref, batch)
    loss.backward()
    want, _ = ref(**batch)
    stats = mine.loss_and_grads(dbatch).cpu()
    T = batch["input_tensor"].shape[1]
    got = mine._last_ws["logits"].view(B, T, -1).cpu()
    assert float((got - want.detach()).abs().max()) < 1e-3
    assert abs(float(stats[0]) - float(loss)) < 1e-3
    refp = dict(ref.named_parameters())
    errs = {n: rel_err(mine.flat.g(n).cpu(), refp[n].grad) for n in mine.flat.params}
    for n, e in errs.items():
        assert e < (3e-2 if n in ("fc1.weight", "out_mlp.0.weight") else 2e-2), sorted(errs.items(), key=lambda kv: -kv[1])[:5]
    mine.check_cluster()
