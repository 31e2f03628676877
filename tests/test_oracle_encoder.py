"""Pins oracle/encoder.py (the functional encoder layer with explicit dropout masks and a rounding hook) to
torch.nn.TransformerEncoder -- the class the reference instantiates (track_mm/cogmen.py:94-102; contrib/nn.py is a
vendored copy of torch.nn's layer) -- in eval mode, with and without the key-padding mask, values and gradients."""
import pytest
import torch
from torch import nn

from oracle.cogmen import COGMENOracle, pick_heads
from oracle.encoder import encoder, round_bf16
from tests.util_cases import check_grad_digest, fill_params


def encoder_from_fixture(fx, D):
    """torch.nn.TransformerEncoder carrying the parameters the fixture's generator gave the reference's vendored
    layers (tests/golden/make_golden.py gen_encoder: same names, same filler, LayerNorm gains + 1)."""
    layer = nn.TransformerEncoderLayer(d_model=D, nhead=int(fx["nhead"]), dropout=0.5, batch_first=True)
    enc = nn.TransformerEncoder(layer, num_layers=2, enable_nested_tensor=False).eval()
    fill_params(enc, int(fx["param_seed"]))
    with torch.no_grad():
        for lyr in enc.layers:
            lyr.norm1.weight.add_(1.0), lyr.norm2.weight.add_(1.0)
    return enc


@pytest.mark.parametrize("name", ["encoder_d24", "encoder_d24_mask", "encoder_d48_h8", "encoder_d712",
                                  "encoder_d712_nomask", "encoder_d1380"])
def test_functional_encoder_equals_reference_layers(golden, name):
    """oracle/encoder.py against the REFERENCE's own contrib/nn.py:206-305 layers (golden vectors): outputs on the
    valid positions, input gradient, every parameter gradient."""
    fx = golden(name)
    x = torch.from_numpy(fx["x"]).requires_grad_()
    B, T, D = x.shape
    enc = encoder_from_fixture(fx, D)
    lengths = torch.from_numpy(fx["lengths"])
    pad = (torch.arange(T)[None, :] >= lengths[:, None]) if bool(fx["masked"]) else None
    valid = ~pad if pad is not None else torch.ones(B, T, dtype=torch.bool)
    got = encoder(x, enc, pad)
    want = torch.from_numpy(fx["out"])
    assert float((got - want)[valid].abs().max()) < 3e-5
    (got * torch.from_numpy(fx["w"])).sum().backward()
    dx = torch.from_numpy(fx["dx"])
    assert float((x.grad - dx).abs().max()) < 1e-4 * max(1.0, float(dx.abs().max()))
    check_grad_digest(fx, [(n, p.grad) for n, p in enc.named_parameters()], tol=2e-4)


@pytest.mark.parametrize("B,T,D,masked", [(3, 13, 24, True), (2, 20, 712, True), (2, 9, 48, False)])
def test_functional_encoder_equals_torch_module(B, T, D, masked):
    torch.manual_seed(B + T)
    layer = nn.TransformerEncoderLayer(d_model=D, nhead=pick_heads(D), dropout=0.5, batch_first=True)
    enc = nn.TransformerEncoder(layer, num_layers=2, enable_nested_tensor=False).eval()
    with torch.no_grad():
        for lyr in enc.layers:
            lyr.norm1.weight.uniform_(0.5, 1.5), lyr.norm1.bias.uniform_(-0.3, 0.3)
    x = torch.randn(B, T, D, requires_grad=True)
    lengths = torch.randint(1, T + 1, (B,))
    lengths[0] = T
    pad = (torch.arange(T)[None, :] >= lengths[:, None]) if masked else None
    want = enc(x, src_key_padding_mask=pad)
    got = encoder(x, enc, pad)
    valid = ~pad if masked else torch.ones(B, T, dtype=torch.bool)
    assert float((got - want)[valid].abs().max()) < 2e-5
    w = torch.randn(B, T, D) * valid[..., None]
    g_want = torch.autograd.grad((want * w).sum(), [x] + list(enc.parameters()), retain_graph=True)
    g_got = torch.autograd.grad((got * w).sum(), [x] + list(enc.parameters()))
    for a, b in zip(g_got, g_want):
        assert float((a - b).abs().max()) < 1e-4 * max(1.0, float(b.abs().max()))


def test_explicit_keeps_and_rounding_hook():
    torch.manual_seed(0)
    B, T, D = 2, 7, 24
    layer = nn.TransformerEncoderLayer(d_model=D, nhead=6, dropout=0.5, batch_first=True)
    enc = nn.TransformerEncoder(layer, num_layers=2, enable_nested_tensor=False)
    x = torch.randn(B, T, D)
    ones = {(l, s): torch.ones(shape) for l in range(2) for s, shape in
            enumerate([(B, 6, T, T), (B * T, D), (B * T, 2048), (B * T, D)])}
    assert torch.equal(encoder(x, enc, keeps=ones), encoder(x, enc))
    zero_ffn = dict(ones)
    zero_ffn[(1, 3)] = torch.zeros(B * T, D)           # dropping the whole FFN branch of layer 1 changes the output
    assert not torch.equal(encoder(x, enc, keeps=zero_ffn), encoder(x, enc))
    r = encoder(x, enc, rnd=round_bf16)
    assert 0 < float((r - encoder(x, enc)).abs().max()) < 0.1
    xr = x.clone().requires_grad_(True)                # straight-through: gradients flow through the rounding
    encoder(xr, enc, rnd=round_bf16).sum().backward()
    assert float(xr.grad.abs().max()) > 0


def test_chained_oracle_paths_agree():
    torch.manual_seed(1)
    from bench import synthetic_batch
    from erc_amd.params import ERCParams
    m = COGMENOracle(712, 100, 17, 2, 6, chained=True).eval()
    batch = synthetic_batch(ERCParams().from_args(["--dataset=iemocap-cogmen-6"]), 3, 15, seed=1)
    with torch.no_grad():
        a, _ = m(**batch)
        m.enc_rnd = lambda t: t
        b, _ = m(**batch)
    assert float((a - b).abs().max()) < 1e-4
