"""BiLSTM (2 layers, hidden 100/direction) on libercgraft vs torch.nn.LSTM on the CPU: packed (DialogueGCN
SeqContext) and unpacked-over-padding (MMGCN text branch) runs, outputs and every gradient."""
import pytest
import torch
from torch.nn.utils.rnn import pack_padded_sequence, pad_packed_sequence

from tests.util_cases import check_grad_digest, fill_params, rel_err

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _runner(lstm, d_in):
    from erc_amd.engine import FlatParams, GemmPlanner
    from erc_amd.rnn import BiLSTM2, lstm_groups
    import copy
    holder = copy.deepcopy(lstm)
    flat = FlatParams(lstm_groups("rnn.", holder), DEV)
    return BiLSTM2(flat, "rnn.", d_in, drop_p=0.4), flat, GemmPlanner(DEV, 1 << 24)


@pytest.mark.parametrize("B,T,D,lens", [(4, 9, 20, [9, 1, 5, 7]), (32, 33, 1242, None), (3, 110, 712, [110, 20, 64]),
                                        # lengths around the 8-step chunks of the recurrence kernels
                                        (9, 25, 36, [25, 8, 16, 17, 15, 24, 1, 2, 9])])
def test_packed_bilstm_matches_torch(B, T, D, lens):
    torch.manual_seed(B + T)
    lens = torch.tensor(lens) if lens else torch.randint(1, T + 1, (B,))
    lens[0] = T
    x = torch.randn(B, T, D) * 0.5
    for b in range(B):
        x[b, lens[b]:] = 0
    lstm = torch.nn.LSTM(D, 100, dropout=0.4, bidirectional=True, num_layers=2, batch_first=True)
    lstm.eval()  # dropout off (the mask cannot match across implementations)
    packed = pack_padded_sequence(x, lens, batch_first=True, enforce_sorted=False)
    want, _ = pad_packed_sequence(lstm(packed, None)[0], batch_first=True)
    gout = torch.randn(B, T, 200)
    for b in range(B):
        gout[b, lens[b]:] = 0
    want.backward(gout)
    run, flat, pl = _runner(lstm, D)
    out = torch.zeros(B * T, 200, device=DEV)
    run.forward(pl, x.to(DEV).view(B * T, D), D, B * T, B, T, T, 1, lens.to(DEV), False, None, out, 200)
    assert float((out.cpu().view(B, T, 200) - want.detach()).abs().max()) < 2e-5
    run.backward(pl, gout.to(DEV).view(B * T, 200), 200)
    from erc_amd import capi
    capi.slab_reduce_batched(pl.ws, flat.grad, pl.job_table(), len(pl.jobs), pl.max_numel)
    for name, p in lstm.named_parameters():
        assert rel_err(flat.g("rnn." + name).cpu(), p.grad) < 1e-3, name


@pytest.mark.parametrize("B,T,D", [(6, 20, 36), (32, 33, 1242)])
def test_packed_bilstm_on_compact_rows_matches_torch(B, T, D):
    """DialogueGCN layout: every LSTM buffer holds the sum(lengths) valid positions in node order (row = node_off[b] + t),
    the input is read through the node -> padded-row map, the output lands in a wider block (row pitch 300)."""
    from erc_amd import capi
    torch.manual_seed(B * T)
    lens = torch.randint(1, T + 1, (B,))
    lens[0], lens[-1] = T, 8
    x = torch.randn(B, T, D) * 0.5
    lstm = torch.nn.LSTM(D, 100, dropout=0.4, bidirectional=True, num_layers=2, batch_first=True)
    lstm.eval()
    packed = pack_padded_sequence(x, lens, batch_first=True, enforce_sorted=False)
    want, _ = pad_packed_sequence(lstm(packed, None)[0], batch_first=True, total_length=T)
    gout = torch.randn(B, T, 200)
    mask = torch.arange(T)[None, :] < lens[:, None]
    gout = gout * mask[:, :, None]
    want.backward(gout)
    N = int(lens.sum())
    node_off = torch.zeros(B + 1, dtype=torch.int32)
    node_off[1:] = torch.cumsum(lens, 0)
    node_row = torch.cat([b * T + torch.arange(int(lens[b])) for b in range(B)]).to(torch.int32)
    run, flat, pl = _runner(lstm, D)
    XW = 300
    out = torch.full((N, XW), 7.0, device=DEV)
    run.forward(pl, x.to(DEV).view(B * T, D), D, N, B, T, T, 1, lens.to(DEV), False, None, out, XW,
                node_off=node_off.to(DEV), node_row=node_row.to(DEV))
    got = out.cpu()
    assert float((got[:, :200] - want.detach()[mask]).abs().max()) < 2e-5
    assert bool((got[:, 200:] == 7.0).all())               # the neighbouring columns are not touched
    dout = torch.zeros(N, XW, device=DEV)
    dout[:, :200] = gout[mask].to(DEV)
    run.backward(pl, dout, XW)
    capi.slab_reduce_batched(pl.ws, flat.grad, pl.job_table(), len(pl.jobs), pl.max_numel)
    for name, p in lstm.named_parameters():
        assert rel_err(flat.g("rnn." + name).cpu(), p.grad) < 1e-3, name


def test_unpacked_time_major_bilstm_with_input_grad():
    """MMGCN layout: [T,B,200] input, every dialogue runs all T steps (reverse direction starts in the padding)."""
    T, B, D = 12, 5, 200
    torch.manual_seed(0)
    x = (torch.randn(T, B, D) * 0.5).requires_grad_()
    lstm = torch.nn.LSTM(D, 100, 2, bidirectional=True, dropout=0.4)
    lstm.eval()
    want, _ = lstm(x)
    gout = torch.randn(T, B, 200)
    want.backward(gout)
    run, flat, pl = _runner(lstm, D)
    out = torch.zeros(T * B, 200, device=DEV)
    run.forward(pl, x.detach().to(DEV).view(T * B, D), D, T * B, B, T, 1, B, None, False, None, out, 200)
    assert float((out.cpu().view(T, B, 200) - want.detach()).abs().max()) < 2e-5
    dx = torch.zeros(T * B, D, device=DEV)
    run.backward(pl, gout.to(DEV).view(T * B, 200), 200, dx=dx, lddx=D)
    assert rel_err(dx.cpu().view(T, B, D), x.grad) < 1e-3
    from erc_amd import capi
    capi.slab_reduce_batched(pl.ws, flat.grad, pl.job_table(), len(pl.jobs), pl.max_numel)
    for name, p in lstm.named_parameters():
        assert rel_err(flat.g("rnn." + name).cpu(), p.grad) < 1e-3, name


def test_interlayer_dropout_is_consistent_between_forward_and_backward():
    """train mode: finite-difference-free check -- with dropout on, backward must use the SAME mask as forward:
    d(sum(out * g))/d(bias_ih_l0) from the kernels equals a torch run that is fed the kernels' own dropped layer-0
    output."""
    B, T, D = 3, 7, 16
    torch.manual_seed(1)
    lens = torch.tensor([7, 3, 5])
    x = torch.randn(B, T, D)
    for b in range(B):
        x[b, lens[b]:] = 0
    lstm = torch.nn.LSTM(D, 100, dropout=0.4, bidirectional=True, num_layers=2, batch_first=True)
    run, flat, pl = _runner(lstm, D)
    rng = torch.tensor([3, 99], dtype=torch.int64, device=DEV)
    out = torch.zeros(B * T, 200, device=DEV)
    run.forward(pl, x.to(DEV).view(B * T, D), D, B * T, B, T, T, 1, lens.to(DEV), True, rng, out, 200)
    ws = run._own["lstm:" + run.prefix]
    h0, h0d = ws["H0"].cpu(), ws["H0d"].cpu()
    kept = h0d != 0
    assert 0.4 < float(kept.float().sum() / (h0 != 0).float().sum()) < 0.8
    assert torch.allclose(h0d[kept], h0[kept] / 0.6, atol=1e-6)
    # torch layer 1 on the kernels' dropped layer-0 output
    l1 = torch.nn.LSTM(200, 100, bidirectional=True, batch_first=True)
    with torch.no_grad():
        for n in ("weight_ih_l0", "weight_hh_l0", "bias_ih_l0", "bias_hh_l0"):
            getattr(l1, n).copy_(getattr(lstm, n.replace("l0", "l1")))
            getattr(l1, n + "_reverse").copy_(getattr(lstm, n.replace("l0", "l1") + "_reverse"))
    inp = h0d.view(B, T, 200).clone().requires_grad_()
    packed = pack_padded_sequence(inp, lens, batch_first=True, enforce_sorted=False)
    want, _ = pad_packed_sequence(l1(packed)[0], batch_first=True)
    assert float((out.cpu().view(B, T, 200) - want.detach()).abs().max()) < 2e-5
    gout = torch.randn(B, T, 200)
    for b in range(B):
        gout[b, lens[b]:] = 0
    want.backward(gout)
    run.backward(pl, gout.to(DEV).view(B * T, 200), 200)
    assert rel_err(ws["dH0d"].cpu().view(B, T, 200), inp.grad) < 1e-3


@pytest.mark.parametrize("name", ["seqcontext_d30", "seqcontext_d1242"])
def test_packed_bilstm_vs_reference_seqcontext(golden, name):
    """lstm.hip (packed run) against the REFERENCE's own SeqContext (track_mm/dgcn_models.py:10-33; golden vectors
    written by tests/golden/make_golden.py gen_dgcn_leaves): outputs, input gradient, every weight gradient."""
    fx = golden(name)
    x, lens = torch.from_numpy(fx["x"]), torch.from_numpy(fx["lengths"])
    B, T, D = x.shape
    holder = torch.nn.Module()
    holder.rnn = torch.nn.LSTM(D, 100, dropout=0.4, bidirectional=True, num_layers=2, batch_first=True)
    fill_params(holder, int(fx["param_seed"]))          # names "rnn.<param>", as the generator's SeqContext
    run, flat, pl = _runner(holder.rnn, D)
    out = torch.zeros(B * T, 200, device=DEV)
    run.forward(pl, x.to(DEV).view(B * T, D), D, B * T, B, T, T, 1, lens.to(DEV), False, None, out, 200)
    got = out.cpu().view(B, T, 200)
    want = torch.from_numpy(fx["out"])                  # pad_packed_sequence: zeros behind each dialogue's length
    assert float((got - want).abs().max()) < 2e-5
    gout = torch.from_numpy(fx["w"]).clone()
    for b in range(B):
        gout[b, lens[b]:] = 0                           # padded outputs are constants: no gradient flows through them
    dx = torch.zeros(B * T, D, device=DEV)
    run.backward(pl, gout.to(DEV).view(B * T, 200), 200, dx=dx, lddx=D)
    want_dx = torch.from_numpy(fx["dx"])
    assert rel_err(dx.cpu().view(B, T, D), want_dx) < 1e-3
    from erc_amd import capi
    capi.slab_reduce_batched(pl.ws, flat.grad, pl.job_table(), len(pl.jobs), pl.max_numel)
    # the generator's names: "rnn." + SeqContext.named_parameters() = "rnn.rnn.<param>"
    check_grad_digest(fx, [("rnn.rnn." + n, flat.g("rnn." + n)) for n, _ in holder.rnn.named_parameters()], tol=1e-3)
