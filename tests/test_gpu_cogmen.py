"""End-to-end COGMEN parity on the GPU: eval logits within 1e-4 (fp32) of the CPU oracle, loss and every
live gradient in train mode with dropout forced to 0, BatchNorm running statistics, one optimizer step."""
import numpy as np
import pytest
import torch

from tests.util_cases import ZERO_GRAD, cogmen_case, run_cogmen_parity, to_device

pytestmark = pytest.mark.gpu
LOGIT_TOL = 1e-4   # north_star: logits within 1e-4 fp32
GRAD_TOL = 2e-3    # relative to the largest entry of each gradient tensor


@pytest.mark.parametrize("case", [
    dict(B=4, min_len=3, max_len=14, dims=dict(a=12, t=20, v=16), seed=3),
    dict(B=1, min_len=1, max_len=1, dims=dict(a=4, t=4, v=4), seed=4),          # single utterance, N=1... BN needs >1
    dict(B=9, min_len=1, max_len=30, dims=dict(a=100, t=100, v=512), seed=5),   # iemocap-cogmen dims (D=712)
    dict(B=8, min_len=20, max_len=60, dims=dict(a=100, t=768, v=512), seed=6),  # sbert dims (D=1380)
], ids=["tiny", "one-utt", "d712", "d1380"])
def test_cogmen_fp32_parity(case):
    if case["max_len"] == 1:
        case = dict(case, B=3)  # three one-utterance dialogues: ragged minimum that still has batch statistics
    # one-utterance dialogues have self loops only: TransformerConv is linear in H1 there, so conv1.bias is a constant
    # shift in front of BatchNorm and its gradient is mathematically zero (compared in absolute terms)
    res = run_cogmen_parity(cogmen_case(**case), zero_grad=("gcn.conv1.bias",) if case["max_len"] == 1 else ())
    assert res["logit_err"] < LOGIT_TOL, res
    assert res["feat_err"] < LOGIT_TOL, res
    assert res["loss_err"] < 1e-5, res
    assert res["acc_match"], res
    assert res["grad_err"] < GRAD_TOL, sorted(res["grad_errs"].items(), key=lambda kv: -kv[1])[:6]
    assert res["bn_mean_err"] < 1e-5 and res["bn_var_err"] < 1e-5, res
    assert res["dead_ok"]


def test_cogmen_config2_shape_parity():
    """BASELINE config 2 shape (B=32, T=110, D=1380, 6 classes), fp32: the headline batch."""
    res = run_cogmen_parity(cogmen_case(B=32, min_len=20, max_len=110, dims=dict(a=100, t=768, v=512), seed=1))
    assert res["logit_err"] < LOGIT_TOL, res
    assert res["grad_err"] < GRAD_TOL, sorted(res["grad_errs"].items(), key=lambda kv: -kv[1])[:6]


def test_cogmen_bf16_feature_mode():
    """bf16 compute mode (config 2's throughput mode): bf16 feature block, bf16 matrix-core products in the input
    projection and the graph part (csrc/cogmen_fused.hip).  The oracle is fed the same bf16-rounded features and
    weights and rounds the same product operands (oracle/pyg.py RoundedLinear / RGCNMeanRounded), so what is left is
    fp32 accumulation order -- and its second-order effect: an fp32 difference in a value that sits on a bf16 rounding
    boundary rounds the other way (measured at config 2: 0.1 % of the H1 entries, one bf16 step each), which moves
    single logits by up to ~1e-3 and single gradient entries by ~1 % of the tensor's scale (0.5 % norm-wise).
    tests/test_gpu_cogmen_fused.py pins every intermediate of the two fused kernels at fp32 tolerance.
    Versus the unrounded fp32 oracle the mode itself moves O(1) logits by up to ~3e-2: quantisation, not
    implementation, error."""
    res = run_cogmen_parity(cogmen_case(B=8, min_len=20, max_len=60, dims=dict(a=100, t=768, v=512), seed=6),
                            compute="bf16")
    assert res["logit_err"] < 1e-3, res
    assert res["grad_err"] < 4e-2, sorted(res["grad_errs"].items(), key=lambda kv: -kv[1])[:8]
    # (bias gradients are column sums of bf16-STORED gradients since round 3: a stored value on a rounding boundary that
    #  goes the other way than in the oracle weighs 2^-8 of itself in a sum with cancellation -- conv1.bias 1.8e-2)
    assert res["grad_norm_err"] < 2.5e-2, res["grad_norm_err"]


def test_cogmen_bf16_feature_mode_full_config2():
    """The same check at the BENCHED shape: BASELINE.json configs[1], B=32, T=110, D=1380 (d_a=100 d_t=768 d_v=512),
    bf16 feature block -- the persistent projection kernel (M >= 1024) and the gathered bf16 weight-gradient path."""
    res = run_cogmen_parity(cogmen_case(B=32, min_len=20, max_len=110, dims=dict(a=100, t=768, v=512), seed=16),
                            compute="bf16")
    assert res["logit_err"] < 1e-3, res
    assert res["grad_err"] < 4e-2, sorted(res["grad_errs"].items(), key=lambda kv: -kv[1])[:8]
    # (bias gradients are column sums of bf16-STORED gradients since round 3: a stored value on a rounding boundary that
    #  goes the other way than in the oracle weighs 2^-8 of itself in a sum with cancellation -- conv1.bias 1.8e-2)
    assert res["grad_norm_err"] < 2.5e-2, res["grad_norm_err"]


# The bf16 compute mode against the UNROUNDED fp32 oracle at the benched shape: the mode's tolerance against the reference
# as a pinned figure (BASELINE.md quotes these bounds next to the bf16 throughput; the 1e-4 path is --compute=f32).
# measured on MI355X (round 3): max 2.5e-3, mean 3.8e-4 at a logit scale of 0.59; gradients 7.6e-2 norm-wise (worst tensor)
BF16_MODE_LOGIT_MAX, BF16_MODE_LOGIT_MEAN, BF16_MODE_GRAD_NORM = 1e-2, 1.5e-3, 0.12


def test_cogmen_bf16_mode_vs_unrounded_fp32_reference_config2():
    """BASELINE.json configs[1] (B=32, T=110, D=1380) in the benched bf16 compute mode versus the oracle WITHOUT any
    rounding (fp32 features, weights and products = the reference's PyTorch-CPU path).  What is bounded here is the
    mode's quantisation (bf16 feature block, bf16 operands of the projection / RGCN / QKVS / weight-gradient products,
    fp32 accumulation): logits of O(1) magnitude deviate by at most BF16_MODE_LOGIT_MAX (max) / BF16_MODE_LOGIT_MEAN
    (mean), every live gradient by at most BF16_MODE_GRAD_NORM norm-wise.  north_star's 1e-4 is met by --compute=f32
    (test_cogmen_config2_shape_parity), whose throughput bench.py reports next to this mode's."""
    res = run_cogmen_parity(cogmen_case(B=32, min_len=20, max_len=110, dims=dict(a=100, t=768, v=512), seed=16),
                            compute="bf16", ref_rounding=False)
    print("bf16 mode vs fp32 reference: max|dlogit| %.3e mean %.3e (logit scale %.2f), grad norm-wise %.3e entry-wise %.3e"
          % (res["logit_err"], res["logit_err_mean"], res["logit_scale"], res["grad_norm_err"], res["grad_err"]))
    assert res["logit_err"] < BF16_MODE_LOGIT_MAX, res
    assert res["logit_err_mean"] < BF16_MODE_LOGIT_MEAN, res
    assert res["grad_norm_err"] < BF16_MODE_GRAD_NORM, res["grad_errs"]


def test_cogmen_bf16_many_nodes_path():
    """N > BN_FUSED_MAX_N (8 192): the path the B=512 figure of BASELINE.md runs on -- BatchNorm batch statistics in their
    own launch (erc_bn_batch_stats), the head kernel reducing its own records, several rounds of weight-gradient work
    items -- against the oracle with the mode's operand rounding.  Small D keeps the oracle's CPU time in seconds."""
    from erc_amd.cogmen import COGMENModule
    case = cogmen_case(B=160, min_len=40, max_len=70, dims=dict(a=12, t=20, v=16), seed=21)
    assert int(case["batch"]["label"].shape[0]) > COGMENModule.BN_FUSED_MAX_N
    res = run_cogmen_parity(case, compute="bf16")
    assert res["logit_err"] < 1e-3, res
    assert res["loss_err"] < 1e-4, res
    assert res["grad_err"] < 5e-2, sorted(res["grad_errs"].items(), key=lambda kv: -kv[1])[:8]
    assert res["grad_norm_err"] < 2e-2, res["grad_norm_err"]
    assert res["bn_mean_err"] < 1e-4 and res["bn_var_err"] < 1e-4, res


@pytest.mark.parametrize("case", [
    dict(B=4, min_len=3, max_len=14, dims=dict(a=12, t=20, v=16), seed=3),
    dict(B=3, min_len=1, max_len=1, dims=dict(a=4, t=4, v=4), seed=4),          # one-utterance dialogues: self loops only
    dict(B=9, min_len=1, max_len=30, dims=dict(a=100, t=100, v=512), seed=5),   # ragged, tiles straddle dialogues
    dict(B=2, min_len=16, max_len=16, dims=dict(a=12, t=20, v=16), seed=8),     # dialogue length == tile height
    dict(B=5, min_len=33, max_len=47, dims=dict(a=12, t=20, v=16), seed=9),
], ids=["tiny", "one-utt", "ragged", "len16", "mid"])
def test_cogmen_bf16_fused_graph_kernels(case, monkeypatch):
    """bf16 compute mode runs the graph part as the two row-tile kernels of csrc/cogmen_fused.hip (halo tiles, bf16
    matrix cores).  Checked against the oracle with the same operand rounding (oracle/pyg.py RoundedLinear /
    RGCNMeanRounded) on shapes that exercise the tile / halo / dialogue-boundary logic; LDS is poisoned with NaN
    patterns before both kernels so that a read of an unwritten pad shows."""
    from tests.util_cases import poison_lds_before
    poison_lds_before(monkeypatch, "cogmen_fwd_tile", "cogmen_bwd_tile")
    # (one-utterance dialogues: conv1.bias is a constant shift in front of BatchNorm only up to the bf16 rounding of H1)
    res = run_cogmen_parity(cogmen_case(**case), compute="bf16",
                            zero_grad=("gcn.conv1.bias",) if case["max_len"] == 1 else (), zero_tol=2e-2)
    # (small N: a single bf16 rounding that goes the other way than in the oracle -- whose window sums run in edge order in
    #  fp32, the kernel's as fp64 prefix differences -- weighs more in a gradient entry than at the benched shape)
    assert res["logit_err"] < 1e-3, res
    assert res["loss_err"] < 1e-4, res
    assert res["grad_err"] < 5e-2, sorted(res["grad_errs"].items(), key=lambda kv: -kv[1])[:8]
    assert res["grad_norm_err"] < 3e-2, res["grad_norm_err"]    # (bias sums of bf16-stored gradients: see the config-2 test)
    assert res["bn_mean_err"] < 1e-4 and res["bn_var_err"] < 1e-4, res


def test_cogmen_bf16_fused_equals_unfused_bf16_storage_mode():
    """The fused graph kernels against the SAME module with them switched off (fp32 kernels on the bf16 feature block):
    the two differ only by the bf16 rounding of the graph part's product operands."""
    from erc_amd.cogmen import COGMENModule
    case = cogmen_case(B=8, min_len=20, max_len=60, dims=dict(a=100, t=768, v=512), seed=6)
    torch.manual_seed(2)
    outs = []
    for fused in (True, False):
        torch.manual_seed(5)
        m = COGMENModule(case["D"], 100, 17, 2, 6, compute="bf16").finalize("cuda:0")
        m.use_fused_graph = fused
        m.train()
        m.drop_p = 0.0
        b = to_device(case["batch"], "cuda:0")
        b["input_tensor"] = b["input_tensor"].to(torch.bfloat16)
        stats = m.loss_and_grads(b).cpu()
        outs.append((float(stats[0]), m.flat.grad.clone(), m._last_ws["H2"].clone()))
    assert abs(outs[0][0] - outs[1][0]) < 2e-2
    assert float((outs[0][2] - outs[1][2]).abs().max()) < 5e-2
    g0, g1 = outs[0][1], outs[1][1]
    assert float((g0 - g1).norm() / g1.norm()) < 0.12    # quantisation of the mode (a sanity bound, not parity)


def test_bf16_shadow_table_tracks_master_weights():
    """Every bf16 weight copy of the bf16 mode (ErcShadowTab: W1, WcatT, Wb, Wq, WqT) is written by the optimizer launch
    and stays bit-identical to packing the rounded fp32 masters."""
    from erc_amd.cogmen import COGMENTrainer
    from erc_amd.params import ERCParams
    p = ERCParams().from_args(["--dataset=iemocap-cogmen-sbert-6", "--compute=bf16", "--optim.lr=0.01"])
    tr = COGMENTrainer(p, "cuda:0")
    m = tr.model
    assert m.shadows is not None and not m._shadow_auto
    case = cogmen_case(B=6, min_len=5, max_len=25, dims=dict(a=100, t=768, v=512), seed=2)
    b = tr.prepare_batch(case["batch"])
    for _ in range(3):
        tr.train_step(b)
    after_steps = m.shadows.buf.clone()
    m.refresh_shadows()
    assert torch.equal(after_steps.view(torch.int16), m.shadows.buf.view(torch.int16))
    from erc_amd.capi import mfma_b_fragment_order as frag
    F = 100
    eq = lambda got, want: torch.equal(got.view(torch.int16), want.contiguous().view(torch.int16))
    wcat = torch.cat([m.flat.w("gcn.conv1.weight").reshape(8 * F, F), m.flat.w("gcn.conv1.root")], 0).to(torch.bfloat16)
    assert eq(m._sh["catT"], frag(wcat.t(), 29))                       # WcatT[o][r*100+c]
    wb = torch.zeros(F, 960, dtype=torch.bfloat16, device=wcat.device)   # Wb[c][r*104+o] = W_r[c][o]
    for r in range(9):
        wb[:, r * 104:r * 104 + F] = wcat[r * F:(r + 1) * F]
    assert eq(m._sh["wb"], frag(wb, 30))
    wq = torch.cat([m.flat.w("gcn.conv2.lin_%s.weight" % n) for n in ("query", "key", "value", "skip")], 0).to(torch.bfloat16)
    assert eq(m._sh["q"], frag(wq, 4))
    assert eq(m._sh["qT"], frag(wq.t(), 13))
    assert eq(m._sh["w1"], m.flat.w("rnn.1.weight").to(torch.bfloat16))


def test_cogmen_train_step_matches_torch_adam():
    """three full train steps (dropout off) == oracle + torch.optim.Adam on the same batches."""
    from oracle.cogmen import COGMENOracle, cogmen_train_step
    from erc_amd.cogmen import COGMENTrainer
    from erc_amd.params import ERCParams
    p = ERCParams().from_args(["--dataset=iemocap-cogmen-4", "--optim.lr=0.001", "--optim.weight_decay=1e-8"])
    tr = COGMENTrainer(p, "cuda:0")
    tr.model.drop_p = 0.0
    ref = COGMENOracle(p.hidden_all, 100, 17, p.n_speakers, p.n_classes, dead_encoder=False)
    ref.load_state_dict({k: v.cpu() for k, v in tr.model.state_dict().items()})
    for m in ref.modules():
        if isinstance(m, torch.nn.Dropout):
            m.p = 0.0
    opt = torch.optim.Adam(ref.parameters(), lr=1e-3, weight_decay=1e-8)
    ref.train()
    for step in range(3):
        case = cogmen_case(B=5, min_len=2, max_len=20, dims=dict(a=100, t=100, v=512), seed=20 + step, n_classes=4)
        loss, acc = cogmen_train_step(ref, opt, case["batch"])
        stats = tr.train_step(tr.prepare_batch(case["batch"])).cpu()
        assert abs(float(stats[0]) - float(loss)) < 2e-5, (step, float(stats[0]), float(loss))
    refp = dict(ref.named_parameters())
    for name in tr.model.flat.params:
        if name in ZERO_GRAD:
            continue  # true gradient is 0: Adam normalises pure rounding noise to steps of +-lr
        got, want = tr.model.flat.w(name).cpu(), refp[name].detach()
        assert float((got - want).abs().max()) < 2e-4, name  # 3 steps of lr=1e-3; Adam divides by sqrt(v)
    # dead encoder untouched
    for k, v in tr.model.state_dict().items():
        if k.startswith("rnn.0."):
            assert torch.equal(v.cpu(), ref.state_dict()[k])


def test_cogmen_dropout_train_mode_runs_and_is_reproducible():
    from erc_amd.cogmen import COGMENTrainer
    from erc_amd.params import ERCParams
    p = ERCParams().from_args(["--dataset=iemocap-cogmen-6"])
    case = cogmen_case(B=6, min_len=5, max_len=25, dims=dict(a=100, t=100, v=512), seed=2)
    losses = []
    for _ in range(2):
        tr = COGMENTrainer(p, "cuda:0")
        b = tr.prepare_batch(case["batch"])
        losses.append([float(tr.train_step(b).cpu()[0]) for _ in range(3)])
    assert losses[0] == losses[1]            # bit-reproducible run to run (no atomics, counter RNG)
    assert losses[0][2] < losses[0][0] + 1.0  # and training does not blow up


def test_train_mm_cli_plugin_surface():
    """``python train_mm.py --module=cogmen --dataset=iemocap-cogmen-4 --modality=atv`` (BASELINE configs[0] shape)
    runs end to end on the GPU: dispatcher -> plugin main() -> collate -> train steps -> test metrics."""
    import json
    import os
    import subprocess
    import sys
    repo = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    res = subprocess.run([sys.executable, "train_mm.py", "--module=cogmen", "--dataset=iemocap-cogmen-4",
                          "--modality=atv", "--epoch=2", "--n_train=12", "--n_test=5", "--train.batch_size=4",
                          "--test.batch_size=4"], cwd=repo, capture_output=True, text=True, timeout=300)
    assert res.returncode == 0, res.stderr[-2000:]
    lines = [json.loads(l) for l in res.stdout.splitlines() if l.startswith("{")]
    steps = [l for l in lines if "Lall" in l]
    epochs = [l for l in lines if "test" in l]
    assert len(steps) == 6 and len(epochs) == 2
    assert all(np.isfinite(l["Lall"]) for l in steps)
    assert set(epochs[-1]["test"]) >= {"acc", "wa", "f1", "mif1", "maf1"}
    bad = subprocess.run([sys.executable, "train_mm.py", "--module=nope"], cwd=repo, capture_output=True, text=True)
    assert bad.returncode == 1 and "cogmen" in bad.stdout


def test_bf16_weight_shadow_tracks_master_weights():
    """bf16 mode: the optimizer kernel keeps a bf16 copy of rnn.1.weight in sync (it is the W operand of the input
    projection), bit-identical to rounding the fp32 master after every step."""
    from erc_amd.cogmen import COGMENTrainer
    from erc_amd.params import ERCParams
    p = ERCParams().from_args(["--dataset=iemocap-cogmen-sbert-6", "--compute=bf16", "--optim.lr=0.01"])
    tr = COGMENTrainer(p, "cuda:0")
    assert tr.model.w1_shadow is not None
    case = cogmen_case(B=6, min_len=5, max_len=25, dims=dict(a=100, t=768, v=512), seed=2)
    b = tr.prepare_batch(case["batch"])
    before = tr.model.flat.w("rnn.1.weight").clone()
    for _ in range(3):
        tr.train_step(b)
    w = tr.model.flat.w("rnn.1.weight")
    assert not torch.equal(w, before)
    assert torch.equal(tr.model.w1_shadow.view(torch.int16), w.to(torch.bfloat16).view(torch.int16))
    # and the projection through the shadow equals the projection through the fp32 weights rounded on the fly
    tr.model.eval()
    with_shadow = tr.model(**b)[0].clone()
    tr.model.w1_shadow = None
    without = tr.model(**b)[0]
    assert torch.equal(with_shadow, without)


def test_fused_head_equals_separate_launches_with_dropout():
    """csrc/head.hip (one launch) against the unfused kernel sequence, train mode with dropout 0.5 and class weights:
    same counter-RNG mask, so losses, every gradient and the BatchNorm running statistics must agree to rounding."""
    from erc_amd.cogmen import COGMENModule
    case = cogmen_case(B=7, min_len=3, max_len=40, dims=dict(a=100, t=100, v=512), seed=5)
    torch.manual_seed(3)
    outs = []
    cw = torch.tensor([0.5, 1.0, 2.0, 1.5, 0.7, 1.2], device="cuda:0")
    sd = None
    for fuse in (True, False):
        m = COGMENModule(case["D"], 100, 17, case["n_speakers"], case["n_classes"])
        if sd is None:
            sd = {k: v.clone() for k, v in m.state_dict().items()}
        m.load_state_dict(sd)
        m.finalize("cuda:0")
        m.rng_state = torch.tensor([11, 1234], dtype=torch.int64, device="cuda:0")
        m.fuse_head, m.drop_p = fuse, 0.5
        m.train()
        batch = {k: (v.to("cuda:0") if torch.is_tensor(v) else v) for k, v in case["batch"].items()}
        stats = m.loss_and_grads(batch, cw).cpu().clone()
        outs.append((stats[:3], m.flat.grad.cpu().clone(), m.gcn.bn.running_mean.cpu().clone(),
                     m.gcn.bn.running_var.cpu().clone()))
    (s0, g0, rm0, rv0), (s1, g1, rm1, rv1) = outs
    assert float(s1[0]) > 0 and abs(float(s0[0]) - float(s1[0])) < 1e-5 * max(1.0, abs(float(s1[0])))
    assert float(s0[1]) == float(s1[1]) and abs(float(s0[2]) - float(s1[2])) < 1e-4
    assert float((g0 - g1).abs().max()) <= 2e-5 * max(1.0, float(g1.abs().max()))
    assert float((rm0 - rm1).abs().max()) < 1e-6 and float((rv0 - rv1).abs().max()) < 1e-6


def test_real_data_path_with_device_collate(tmp_path):
    """``--synthetic=False --data_root=... --device_collate``: the reference's IEMOCAP pickle layout (synthetic content)
    is read by datasets.py, kept resident in HBM and batched on the device; loss curve identical to the DataLoader path
    with the same shuffle seed is not required -- both must train and report the metric set."""
    import json
    import os
    import subprocess
    import sys
    from tests.test_datasets import _write_iemocap
    _write_iemocap(str(tmp_path), 6, np.random.default_rng(5), with_maps=False)
    repo = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    ck = str(tmp_path / "last_model.ckpt")
    for extra in (["--device_collate", "--save=" + ck], ["--load=" + ck]):
        res = subprocess.run([sys.executable, "train_mm.py", "--module=cogmen", "--dataset=iemocap-cogmen-6", "--epoch=1",
                              "--synthetic=False", "--data_root=" + str(tmp_path), "--train.batch_size=2",
                              "--test.batch_size=2", "--compute=bf16"] + extra,
                             cwd=repo, capture_output=True, text=True, timeout=300)
        assert res.returncode == 0, res.stderr[-2000:]
        lines = [json.loads(l) for l in res.stdout.splitlines() if l.startswith("{")]
        assert len([l for l in lines if "Lall" in l]) == 2 and len([l for l in lines if "test" in l]) == 1
        assert all(np.isfinite(l["Lall"]) for l in lines if "Lall" in l)
        assert os.path.exists(ck)


def test_checkpoint_envelope_round_trip_with_reference_style_trainer(tmp_path):
    """Checkpoints in the reference's envelope ({'models': {'model': ...}, 'optims': {'optim': Adam state_dict}, ...},
    mmbase.py:325-333): (1) a file written the way the reference trainer writes it (oracle module + torch.optim.Adam)
    loads into the HIP trainer and training continues identically; (2) a file written here loads into the reference-style
    pair with plain load_state_dict calls and continues identically."""
    from oracle.cogmen import COGMENOracle, cogmen_train_step
    from erc_amd import checkpoint
    from erc_amd.cogmen import COGMENTrainer
    from erc_amd.params import ERCParams
    p = ERCParams().from_args(["--dataset=iemocap-cogmen-4", "--optim.lr=0.001", "--optim.weight_decay=1e-8"])
    batches = [cogmen_case(B=5, min_len=2, max_len=20, dims=dict(a=100, t=100, v=512), seed=40 + s, n_classes=4)["batch"]
               for s in range(4)]

    def make_ref():
        ref = COGMENOracle(p.hidden_all, 100, 17, p.n_speakers, p.n_classes, dead_encoder=False)
        for m in ref.modules():
            if isinstance(m, torch.nn.Dropout):
                m.p = 0.0
        return ref.train(), torch.optim.Adam(ref.parameters(), lr=1e-3, weight_decay=1e-8)

    def close(tr, ref, tol):
        refp = dict(ref.named_parameters())
        for name in tr.model.flat.params:
            if name not in ZERO_GRAD:
                assert float((tr.model.flat.w(name).cpu() - refp[name].detach()).abs().max()) < tol, name

    # (1) reference-style file -> HIP trainer
    torch.manual_seed(11)
    ref, opt = make_ref()
    for b in batches[:2]:
        cogmen_train_step(ref, opt, b)
    f1 = str(tmp_path / "best_model.ckpt")
    torch.save({"optims": {"optim": opt.state_dict()}, "models": {"model": ref.state_dict()}, "others": {},
                "thtensor": {}, "nptensor": {}}, f1)
    tr = COGMENTrainer(p, "cuda:0")
    tr.model.drop_p = 0.0
    checkpoint.load(tr, f1)
    assert int(tr.optim.state[0]) == 2
    close(tr, ref, 1e-6)
    l_ref, _ = cogmen_train_step(ref, opt, batches[2])
    l_hip = float(tr.train_step(tr.prepare_batch(batches[2])).cpu()[0])
    assert abs(l_hip - float(l_ref)) < 2e-5
    close(tr, ref, 1e-4)
    # (2) HIP trainer file -> reference-style pair
    f2 = checkpoint.save(tr, str(tmp_path / "last_model.ckpt"))
    ck = torch.load(f2, map_location="cpu", weights_only=True)
    assert set(ck) == {"optims", "models", "others", "thtensor", "nptensor"}
    assert int(ck["models"]["model"]["gcn.bn.num_batches_tracked"]) == 3
    ref2, opt2 = make_ref()
    ref2.load_state_dict(ck["models"]["model"])
    opt2.load_state_dict(ck["optims"]["optim"])
    l_ref2, _ = cogmen_train_step(ref2, opt2, batches[3])
    l_hip2 = float(tr.train_step(tr.prepare_batch(batches[3])).cpu()[0])
    assert abs(l_hip2 - float(l_ref2)) < 2e-5
    close(tr, ref2, 1e-4)


def test_training_loop_graph_replay_equals_eager():
    """trainer.run (what ``train_mm.py --module=cogmen`` executes): with ``--fixed_batches`` the batches repeat every epoch,
    so from epoch 1 on every step is a replay of the shape's captured HIP graph.  Per-step losses must be IDENTICAL to the
    same loop run eagerly (``--graph_replay=False``): the first occurrence of a shape runs eagerly and is then captured
    (capture records, it does not execute), the dropout RNG offset and the optimizer step live on the device."""
    import json
    import os
    import subprocess
    import sys
    repo = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    runs = {}
    for tag, extra in (("graph", []), ("eager", ["--graph_replay=False"])):
        res = subprocess.run([sys.executable, "train_mm.py", "--module=cogmen", "--dataset=iemocap-cogmen-6", "--epoch=3",
                              "--n_train=20", "--n_test=6", "--train.batch_size=8", "--test.batch_size=8", "--fixed_batches",
                              "--compute=bf16"] + extra, cwd=repo, capture_output=True, text=True, timeout=300)
        assert res.returncode == 0, res.stderr[-2000:]
        lines = [json.loads(l) for l in res.stdout.splitlines() if l.startswith("{")]
        runs[tag] = ([l["Lall"] for l in lines if "Lall" in l], [l for l in lines if "test" in l])
    assert len(runs["graph"][0]) == 9 and runs["graph"][0] == runs["eager"][0]
    ep = runs["graph"][1]
    assert ep[0]["graph_replays"] == 0 and ep[0]["eager_steps"] == 3          # epoch 0: every shape is new
    assert ep[2]["graph_replays"] == 6 and ep[2]["eager_steps"] == 3          # epochs 1, 2: replays only
    assert runs["eager"][1][2]["graph_replays"] == 0


def test_capacity_mode_step_equals_exact_step():
    """COGMEN bf16, capacity mode (COGMENModule.dynamic_n): the batch sits in capacity-sized static buffers -- more dialogue
    slots than dialogues (length 0), a longer T, a label buffer of N_cap > N entries -- and every kernel of the step reads
    the true node count from the device.  Loss, statistics, every gradient and BatchNorm's running statistics equal the
    exact-shape step (gradients up to the fp32 summation order of the weight-gradient splits, which are cut for the
    capacity); the rows beyond N hold NaN-free garbage of an earlier, larger batch."""
    from erc_amd.cogmen import COGMENTrainer
    from erc_amd.params import ERCParams
    outs = []
    for cap in (False, True):
        torch.manual_seed(0)
        p = ERCParams().from_args(["--dataset=iemocap-cogmen-sbert-6", "--compute=bf16", "--train.batch_size=12"])
        tr = COGMENTrainer(p, "cuda:0")
        tr.model.drop_p = 0.0
        big = tr.prepare_batch(cogmen_case(B=12, min_len=30, max_len=70, dims=dict(a=100, t=768, v=512), seed=3)["batch"])
        small = tr.prepare_batch(cogmen_case(B=9, min_len=5, max_len=40, dims=dict(a=100, t=768, v=512), seed=4)["batch"])
        lr, tr.optim.lr = tr.optim.lr, 0.0   # the first step only fills the buffers: the weights of the compared step are the same
        if cap:
            tr.t_cap = 80
            key, make, fill = tr.capacity_bucket(big)
            assert key == ("capacity", 12, 80, -(-int(big["label"].shape[0]) // 256) * 256)
            static = make()
            tr.model.dynamic_n = True
            fill(static, big)
            tr.train_step(static)                 # leaves rows of a LARGER batch behind in every buffer of the bucket
            fill(static, small)
            tr.optim.lr = lr
            stats = tr.train_step(static).cpu()
            tr.model.dynamic_n = False
        else:
            tr.train_step(big)
            tr.optim.lr = lr
            stats = tr.train_step(small).cpu()
        n = int(small["label"].shape[0])
        outs.append((stats, tr.model.flat.grad.clone(), tr.model._last_ws["logits"][:n].clone(), tr.model.gcn.bn.running_mean.clone(),
                     tr.model.flat.data.clone()))
    (s0, g0, l0, m0, w0), (s1, g1, l1, m1, w1) = outs
    assert abs(float(s0[0]) - float(s1[0])) < 1e-6 and float(s0[1]) == float(s1[1])
    assert float((l0 - l1).abs().max()) < 1e-5
    assert float((g0 - g1).abs().max()) <= 2e-5 * float(g0.abs().max())
    assert float((m0 - m1).abs().max()) < 1e-6
    assert float((w0 - w1).abs().max()) < 2.5e-3    # one Adam step of lr 1e-3 (a near-zero gradient entry may change sign)


def _run_cli(args, timeout=300):
    import json
    import os
    import subprocess
    import sys
    repo = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    res = subprocess.run([sys.executable, "train_mm.py"] + args, cwd=repo, capture_output=True, text=True, timeout=timeout)
    assert res.returncode == 0, res.stderr[-2000:]
    lines = [json.loads(l) for l in res.stdout.splitlines() if l.startswith("{")]
    return [l["Lall"] for l in lines if "Lall" in l], [l for l in lines if "test" in l]


@pytest.mark.parametrize("compute", ["bf16", "f32x32"])
def test_training_loop_default_sampling_replays_capacity_buckets(compute):
    """trainer.run with the REFERENCE'S sampling (dialogues reshuffled every epoch, smaller last batch:
    lumo/trainer/trainer.py:429-442, mmbase.py:468; no --fixed_batches): batch shapes never repeat, the capacity buckets
    do -- after the first epoch almost every step is a replay of one of a handful of graphs, and the per-step losses are
    IDENTICAL to the same loop with capture switched off (same buckets, every step eager on the static buffers)."""
    args = ["--module=cogmen", "--dataset=iemocap-cogmen-6", "--epoch=3", "--n_train=44", "--n_test=6", "--train.batch_size=8",
            "--test.batch_size=8", "--compute=" + compute]      # (f32x32: the 1e-4 parity path runs the same capacity-mode step)
    g_loss, g_ep = _run_cli(args)
    e_loss, e_ep = _run_cli(args + ["--graph_capture=False"])
    assert len(g_loss) == 18 and g_loss == e_loss
    assert e_ep[2]["graph_replays"] == 0 and e_ep[2]["graphs_captured"] == 0
    assert g_ep[2]["graphs_captured"] <= 8
    assert g_ep[2]["graph_replays"] + g_ep[2]["eager_steps"] == 18
    assert g_ep[2]["eager_steps"] == g_ep[2]["graphs_captured"]            # one eager step per bucket, everything else replayed
    assert g_ep[2]["graph_replays"] >= 10


@pytest.mark.parametrize("compute", ["bf16", "f32x32"])
def test_resident_epochs_equal_the_collated_loop(compute):
    """``--resident``: the dialogues stay in the HBM store and a step's input is 2 B int32 (lengths | first store rows of
    the batch's dialogues) -- the projection launch reads features, speakers and labels straight from the store.  Same
    seed, same permutations, same batches as the device-collated loop (which pads every batch into a [B, T, D] block and
    copies it into the bucket's static buffers): the epoch mean losses are equal and every step after a bucket's first
    is a graph replay."""
    args = ["--module=cogmen", "--dataset=iemocap-cogmen-6", "--epoch=3", "--n_train=44", "--n_test=6", "--train.batch_size=8",
            "--test.batch_size=8", "--compute=" + compute, "--device_collate"]
    c_loss, c_ep = _run_cli(args)
    r_loss, r_ep = _run_cli(args + ["--resident"])
    assert len(c_loss) == 18 and len(r_loss) == 3
    for e in range(3):
        want = sum(c_loss[6 * e:6 * e + 6]) / 6
        # (the two loops round their node counts up to different capacity edges, so the weight-gradient launch may cut K
        #  into different split counts: fp32 sums in another order, a few 1e-6 of the loss after an epoch)
        assert abs(r_loss[e] - want) < 1e-5 * max(1.0, abs(want)), (e, r_loss[e], want)
    assert r_ep[2]["graphs_captured"] <= 8 and r_ep[2]["graph_replays"] + r_ep[2]["eager_steps"] == 18
    assert r_ep[2]["eager_steps"] == r_ep[2]["graphs_captured"]
    # (same reason: a logit pair closer than that rounding may flip its argmax -- a few utterances of the ~400 at most)
    assert all(abs(a["test"]["acc"] - b["test"]["acc"]) <= 0.01 for a, b in zip(r_ep, c_ep))


def test_optimizer_fused_into_the_weight_gradient_launch_equals_the_separate_launch():
    """COGMEN bf16, one rank: the weight-gradient launch's last arrivers apply Adam themselves (csrc/wgrad_bf16.hip W2Adam; no
    optimizer launch).  Same gradients, same element-wise update: parameters, both moments, the bf16 shadows, the step count
    and the dropout offset are BIT-IDENTICAL to the step with the separate optimizer launch, over several steps with
    dropout on; a health word left raised by the previous step is rolled into the event count by the step's first launch."""
    from erc_amd import capi
    from erc_amd.cogmen import COGMENTrainer
    from erc_amd.params import ERCParams
    outs = []
    for fused in (True, False):
        torch.manual_seed(0)
        p = ERCParams().from_args(["--dataset=iemocap-cogmen-sbert-6", "--compute=bf16", "--optim.lr=0.003", "--optim.weight_decay=1e-4"])
        tr = COGMENTrainer(p, "cuda:0")
        assert tr.model.fused_optim is tr.optim
        if not fused:
            tr.model.fused_optim = None
        losses = []
        for step in range(4):
            b = tr.prepare_batch(cogmen_case(B=7, min_len=4, max_len=33, dims=dict(a=100, t=768, v=512), seed=30 + step)["batch"])
            if step == 2:
                tr.model.flat.health.fill_(capi.HEALTH_RAISED)     # as a timed-out step 1 would have left it: rolled by step 2's first launch
            losses.append(float(tr.train_step(b).cpu()[0]))
            assert tr.model._last_ws["planner"].adam_fused == fused
            assert int(tr.model.flat.health[0].item()) == 0 and int(tr.model.flat.events[0].item()) == (1 if step >= 2 else 0)
        f = tr.model.flat
        outs.append((losses, f.data.clone(), f.exp_avg.clone(), f.exp_avg_sq.clone(), tr.model.shadows.buf.clone(),
                     tr.optim.state[:2].clone(), tr.optim.state[4:].clone()))
    a, b = outs
    assert a[0] == b[0]
    for x, y in zip(a[1:], b[1:]):
        assert torch.equal(x.view(torch.int16) if x.dtype == torch.bfloat16 else x, y.view(torch.int16) if y.dtype == torch.bfloat16 else y)
    assert int(a[5][0]) == 4 and bool((a[6] == 4).all())          # 4 steps; every private step copy agrees


def test_fused_optimizer_timeout_is_reported_as_a_partial_update_and_the_run_falls_back():
    """The weight-gradient launch with the optimizer inside waits (bounded) for the splits of a tile.  With every tile's arrival
    counter set back by one (no tile can ever complete) and a small bound (erc_wgrad_bf16_set_spin_limit) every wait times out:
    the health word is raised and `check_cluster` reports a PARTIAL update (this launch gives up per tile: what does not wait --
    BatchNorm's scale / shift records, the step count -- was updated, the tiles were not), marks the parameter buffer tainted
    and checkpoint.save refuses.  From then on the planner keeps the two-launch form (its optimizer skips a step as a whole); the
    NEXT step's first launch (erc_cogmen_fwd_tile: the health roll folded in) counts the event and clears the word, and -- from
    a restored state -- equals the same step of a trainer that never saw the timeout."""
    from erc_amd import capi, checkpoint
    from erc_amd.cogmen import COGMENTrainer
    from erc_amd.params import ERCParams

    def fresh():
        torch.manual_seed(0)
        p = ERCParams().from_args(["--dataset=iemocap-cogmen-sbert-6", "--compute=bf16"])
        tr = COGMENTrainer(p, "cuda:0")
        tr.model.drop_p = 0.0
        b = tr.prepare_batch(cogmen_case(B=32, min_len=20, max_len=110, dims=dict(a=100, t=768, v=512), seed=41)["batch"])
        tr.train_step(b)                                     # step 0: builds the launch's table, slabs and counters
        torch.cuda.synchronize()
        return tr, b

    tr, b = fresh()
    ref, rb = fresh()
    ws = tr.model._last_ws
    assert ws["planner"].adam_fused and ws["w16_fused"]
    tiles = ws["w16_tiles"]
    ws["w16_counters"][:tiles] -= 1                          # a split short on every tile
    capi.wgrad_bf16_set_spin_limit(64)
    try:
        tr.train_step(b)
        torch.cuda.synchronize()
    finally:
        capi.wgrad_bf16_set_spin_limit(0)
        ws["w16_counters"][:tiles] += 1
    f, g = tr.model.flat, ref.model.flat
    assert int(f.health[0].item()) == capi.HEALTH_RAISED
    with pytest.raises(capi.ErcGraftError, match="gives up per gradient tile"):
        tr.model.check_cluster()                                              # reports, clears, taints
    assert f.tainted
    with pytest.raises(capi.ErcGraftError, match="not saving"):
        checkpoint.save(tr, "/tmp/never_written.pt")
    # back to the state after step 0, as a restart from a checkpoint would (the timed-out launch updated what does not wait)
    f.data.copy_(g.data), f.exp_avg.copy_(g.exp_avg), f.exp_avg_sq.copy_(g.exp_avg_sq)
    tr.optim.state.copy_(ref.optim.state)
    tr.model.refresh_shadows()
    tr.model.gcn.bn.running_mean.copy_(ref.model.gcn.bn.running_mean), tr.model.gcn.bn.running_var.copy_(ref.model.gcn.bn.running_var)
    f.health.fill_(capi.HEALTH_RAISED)                   # as the timed-out step left it
    s1, s0 = tr.train_step(b).cpu(), ref.train_step(rb).cpu()
    torch.cuda.synchronize()
    assert not tr.model._last_ws["planner"].adam_fused       # the tainted run keeps the two-launch form
    assert int(f.events[0].item()) == 1 and int(f.health[0].item()) == 0
    assert torch.equal(s1[:3], s0[:3]) and torch.equal(f.data, g.data)
    with pytest.raises(capi.ErcGraftError, match="timed out"):
        tr.model.check_cluster()
    ref.model.check_cluster()                            # nothing to report


@pytest.mark.parametrize("compute", ["bf16", "f32x32"])
def test_precapture_leaves_the_training_state_untouched(compute):
    """trainer.StepGraphs.precapture (data parallel: every capacity bucket captured up front) runs one REAL warm-up step per
    bucket on a made-up batch.  Whatever those steps change must be back afterwards, bit for bit: parameters, both Adam
    moments, the optimizer's step count / dropout offset / per-workgroup step copies, the bf16 weight shadows, BatchNorm's
    running statistics, the health word and its event counter.  (Round 3 relied on a raised health word to make the optimizer
    skip the warm-ups; the step's first launch rolls that word, so up to 32 Adam steps on zero features ran silently.)"""
    import types
    import track_mm.cogmen as plugin
    from erc_amd.trainer import StepGraphs
    params = plugin.ParamsType().from_args(["--dataset=iemocap-cogmen-4", "--modality=atv", "--compute=" + compute])
    params.train.batch_size = 8
    tr = plugin.COGMENTrainer(params, "cuda:0")
    tr.t_cap = 70                            # buckets of 256 and 512 nodes
    host = cogmen_case(B=8, min_len=20, max_len=40, dims=params.dims(), seed=2, n_classes=params.n_classes)["batch"]
    batch = tr.prepare_batch(host)
    for _ in range(3):                       # a state worth protecting: non-zero moments, step count 3
        tr.train_step(batch)
    torch.cuda.synchronize()
    fl, opt, bn = tr.model.flat, tr.optim, tr.model.gcn.bn
    before = [t.clone() for t in (fl.data, fl.exp_avg, fl.exp_avg_sq, opt.state, tr.model.shadows.buf, bn.running_mean, bn.running_var)]
    graphs = StepGraphs(tr)
    graphs.precapture(batch)
    torch.cuda.synchronize()
    assert graphs.captures == 2
    after = (fl.data, fl.exp_avg, fl.exp_avg_sq, opt.state, tr.model.shadows.buf, bn.running_mean, bn.running_var)
    for name, a, b in zip(("parameters", "exp_avg", "exp_avg_sq", "optimizer state", "shadows", "running_mean", "running_var"), before, after):
        assert torch.equal(a, b), name
    assert int(fl.health[0]) == 0 and int(fl.events[0]) == 0
    # and the captured graphs train: one replayed step equals the same step of a trainer that never precaptured
    ref = plugin.COGMENTrainer(params, "cuda:0")
    ref.t_cap = 70
    for _ in range(3):
        ref.train_step(batch)
    graphs.lazy = False
    s1 = graphs.step(batch).clone()
    s2 = ref.train_step(batch)
    torch.cuda.synchronize()
    assert graphs.replays == 1
    # (the bucket's launches are sized for its capacity: other split counts than the exact-shape step, sums in another order)
    assert float((fl.data - ref.model.flat.data).abs().max()) < 2e-6 and torch.allclose(s1[:3], s2[:3], rtol=1e-5, atol=1e-6)
