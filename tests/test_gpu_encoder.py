"""K9 (faithful-cost mode of the dead COGMEN encoder): the HIP encoder block against torch.nn.TransformerEncoder with
the same parameters (post-norm, ReLU, ffn 2048, batch_first, no padding mask -- the configuration of
track_mm/cogmen.py:94-102) in inference mode.  bf16 operands / fp32 accumulate: tolerance is that of the bf16 rounding
of activations and weights, measured relative to the output scale (LayerNorm outputs are O(1))."""
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


@pytest.mark.parametrize("B,T,D", [(3, 13, 24), (2, 37, 712), (4, 110, 1380)])
def test_encoder_block_matches_torch(B, T, D):
    from erc_amd.cogmen import COGMENModule
    from erc_amd.encoder import EncoderBlock
    torch.manual_seed(B + T)
    m = COGMENModule(D, 100, 17, 2, 6)
    enc = m.rnn[0].eval()
    with torch.no_grad():     # non-trivial LayerNorm affine parameters
        for lyr in enc.layers:
            lyr.norm1.weight.uniform_(0.5, 1.5), lyr.norm1.bias.uniform_(-0.3, 0.3)
            lyr.norm2.weight.uniform_(0.5, 1.5), lyr.norm2.bias.uniform_(-0.3, 0.3)
    x = torch.randn(B, T, D)
    blk = EncoderBlock(enc, DEV)
    for dt in (torch.float32, torch.bfloat16):
        xin = x.to(dt)
        got = blk.forward(xin.to(DEV)).cpu().clone()
        with torch.no_grad():
            want = enc(xin.float())
        err = float((got - want).abs().max())
        assert err < 6e-2, (dt, err)
        assert float((got - want).abs().mean()) < 8e-3
    assert blk.flops(B, T) > 0


@pytest.mark.parametrize("name", ["encoder_d24", "encoder_d712_nomask", "encoder_d1380"])
def test_encoder_block_vs_reference_layers(golden, name):
    """K9 against the REFERENCE's own contrib/nn.py:206-305 layers (golden vectors written by
    tests/golden/make_golden.py gen_encoder: two layers, eval mode, no padding mask -- what cogmen.py:146 runs).
    (a) vs the fp32 fixture: bf16 operands bound the error to the mode's quantisation -- mean |err| < 4e-3, 99.9th
        percentile < 3e-2 on O(1) LayerNorm outputs;
    (b) vs oracle/encoder.py (itself pinned to the same fixture at 3e-5 on the CPU) fed the SAME bf16 roundings the
        kernels apply: what is left is accumulation order and single-ulp rounding flips -- mean |err| < 1e-3.
    A wrong head split / q-k-v layout moves individual outputs by O(1): mean |err| would be ~0.5, far outside both."""
    from erc_amd.encoder import EncoderBlock
    from oracle.encoder import encoder, round_bf16
    from tests.test_oracle_encoder import encoder_from_fixture
    fx = golden(name)
    x = torch.from_numpy(fx["x"])
    B, T, D = x.shape
    enc = encoder_from_fixture(fx, D)
    blk = EncoderBlock(enc, DEV)
    got = blk.forward(x.to(DEV)).cpu().clone()
    want = torch.from_numpy(fx["out"])
    e = (got - want).abs().flatten()
    print(name, "vs reference fp32: mean %.2e p99.9 %.2e max %.2e" % (float(e.mean()), float(e.quantile(0.999)), float(e.max())))
    assert float(e.mean()) < 4e-3 and float(e.quantile(0.999)) < 3e-2 and float(e.max()) < 8e-2
    with torch.no_grad():
        rounded = encoder(x, enc, None, rnd=round_bf16)
    r = (got - rounded).abs().flatten()
    print(name, "vs bf16-rounded oracle: mean %.2e p99.9 %.2e max %.2e" % (float(r.mean()), float(r.quantile(0.999)), float(r.max())))
    assert float(r.mean()) < 1e-3 and float(r.quantile(0.999)) < 1.5e-2


def test_faithful_mode_changes_cost_not_results():
    """--faithful_dead_encoder runs the dead encoder every step and discards its output: losses are bit-identical to the
    default mode (the live path does not read anything the encoder writes)."""
    from bench import synthetic_batch
    from erc_amd.cogmen import COGMENTrainer
    from erc_amd.params import ERCParams
    losses = []
    for extra in ([], ["--faithful_dead_encoder"]):
        p = ERCParams().from_args(["--dataset=iemocap-cogmen-6", "--compute=bf16"] + extra)
        tr = COGMENTrainer(p, DEV)
        assert (tr.encoder is not None) == bool(extra)
        b = tr.prepare_batch(synthetic_batch(p, 4, 30, seed=2))
        losses.append([float(tr.train_step(b).cpu()[0]) for _ in range(3)])
    assert losses[0] == losses[1]
