"""Shared builders for parity tests: synthetic batches, oracle <-> HIP model pairs."""
import types

import numpy as np
import torch
from torch.nn import functional as F

from erc_amd.collate import ERCCollate
from erc_amd.synthetic import make_dialogues


def make_batch(B, dims, n_speakers=2, n_classes=6, min_len=3, max_len=14, seed=1, modality="atv",
               batch_first=True, speaker_onehot=False, force_max=False):
    dialogs = make_dialogues(B, dims, n_speakers=n_speakers, n_classes=n_classes, min_len=min_len,
                             max_len=max_len, seed=seed, force_max=force_max)
    p = types.SimpleNamespace(batch_first=batch_first, speaker_onehot=speaker_onehot, n_classes=n_classes,
                              n_speakers=n_speakers, modality=modality)
    batch = ERCCollate(p)([[d] for d in dialogs])
    batch.pop("utterance_texts", None)
    return batch


def cogmen_case(B=4, min_len=3, max_len=14, dims=None, seed=3, n_classes=6, n_speakers=2):
    dims = dims or dict(a=12, t=20, v=16)
    return dict(batch=make_batch(B, dims, n_speakers, n_classes, min_len, max_len, seed), D=sum(dims.values()),
                n_classes=n_classes, n_speakers=n_speakers, seed=seed)


def to_device(batch, device):
    return {k: (v.to(device) if torch.is_tensor(v) else v) for k, v in batch.items()}


ZERO_GRAD = ("gcn.conv2.lin_key.bias", "gcn.conv2.lin_value.bias", "gcn.conv2.lin_skip.bias")


def rel_err(a, b, floor=1e-5):
    """max |a-b| relative to the scale of the reference tensor.  ``floor`` keeps gradients that are
    mathematically zero (TransformerConv key bias: softmax is shift invariant) from dividing noise by noise."""
    a, b = a.double(), b.double()
    return float((a - b).abs().max() / (b.abs().max() + floor))


def run_cogmen_parity(case, device="cuda:0", compute="f32", zero_grad=(), zero_tol=1e-5, ref_rounding=True, kink_aware=False,
                      kink_tol=2e-5):
    """eval-mode logits and train-mode (dropout p=0) loss/gradients: HIP path vs oracle.
    ``ref_rounding=False`` with compute="bf16": the oracle stays the UNROUNDED fp32 restatement of the reference (fp32
    features, fp32 weights, fp32 products) -- what is measured is the bf16 compute mode's deviation from the reference,
    quantisation included."""
    from oracle.cogmen import COGMENOracle
    from erc_amd.cogmen import COGMENModule
    torch.manual_seed(case["seed"])
    D, C, S = case["D"], case["n_classes"], case["n_speakers"]
    # bf16 mode runs the graph part's dense products on bf16 matrix cores: the oracle rounds the same operands
    ref = COGMENOracle(D, 100, 17, S, C, dead_encoder=False, bf16_products=(compute == "bf16" and ref_rounding))
    with torch.no_grad():  # make BN affine / running stats non-trivial
        ref.gcn.bn.weight.uniform_(0.5, 1.5)
        ref.gcn.bn.bias.uniform_(-0.3, 0.3)
        ref.gcn.bn.running_mean.uniform_(-0.2, 0.2)
        ref.gcn.bn.running_var.uniform_(0.5, 1.5)
        ref.gcn.conv1.bias.uniform_(-0.1, 0.1)
    mine = COGMENModule(D, 100, 17, S, C, compute=compute)
    mine.load_state_dict(ref.state_dict())
    mine.finalize(device)
    batch = case["batch"]
    dbatch = to_device(batch, device)
    if compute == "bf16":
        # bf16 mode stores the feature block in bf16 and rounds rnn.1.weight to bf16 while staging it:
        # give the oracle the SAME rounded operands so that the test isolates implementation error
        # (fp32 accumulate) from the quantisation error of the mode itself.
        dbatch["input_tensor"] = dbatch["input_tensor"].to(torch.bfloat16)
        if ref_rounding:
            batch = dict(batch, input_tensor=batch["input_tensor"].to(torch.bfloat16).float())
            with torch.no_grad():
                ref.rnn[1].weight.copy_(ref.rnn[1].weight.to(torch.bfloat16).float())
                mine.rnn[1].weight.copy_(ref.rnn[1].weight.to(device))
    out = {}
    # --- eval logits
    ref.eval(), mine.eval()
    with torch.no_grad():
        want, want_feat = ref(**batch)
    got, got_feat = mine(**dbatch)
    out["logit_err"] = float((got.cpu() - want).abs().max())
    out["logit_err_mean"] = float((got.cpu() - want).abs().mean())
    out["logit_scale"] = float(want.abs().max())
    out["feat_err"] = float((got_feat.cpu() - want_feat).abs().max())
    # --- train mode, dropout off: loss + every live gradient + BN running stats
    ref.train(), mine.train()
    for m in ref.modules():
        if isinstance(m, torch.nn.Dropout):
            m.p = 0.0
    mine.drop_p = 0.0
    stats = mine.loss_and_grads(dbatch).cpu()
    if kink_aware:
        # ReLU / LeakyReLU have a derivative JUMP at 0: a unit whose pre-activation lies within the compared path's own
        # forward deviation of 0 may sit on the other side there, and that one unit's whole gradient contribution appears or
        # vanishes (measured at config 2: 3e-3 of cls.0.weight's scale per unit, whatever the size of the deviation).  The
        # oracle's backward is therefore evaluated with the activation PATTERN of the compared path; every unit whose pattern
        # differs must have an oracle pre-activation below ``kink_tol``.
        ws = mine._last_ws
        pat = {"cls": ws["Z"].cpu() > 0, "gcn": ws["H3"].cpu() > 0}
        flips = []

        def masked(slope, key):
            def fwd(x):
                m = pat[key]
                diff = m != (x > 0)
                flips.append((key, int(diff.sum()), float(x.detach()[diff].abs().max()) if bool(diff.any()) else 0.0))
                return torch.where(m, x, x * slope)
            return fwd
        ref.cls[1].forward = masked(0.0, "cls")
        ref.gcn.relu.forward = masked(0.01, "gcn")
    logits, _ = ref(**batch)
    loss = F.cross_entropy(logits, batch["label"])
    ref.zero_grad()
    loss.backward()
    if kink_aware:
        out["kink_flips"] = flips
        assert all(mag < kink_tol for _, _, mag in flips), flips
    out["loss_err"] = abs(float(stats[0]) - float(loss))
    out["acc_match"] = int(stats[1]) == int((logits.argmax(-1) == batch["label"]).sum())
    worst, names, worst_norm = 0.0, {}, 0.0
    ref_params = dict(ref.named_parameters())
    for name in mine.flat.params:
        if name in ZERO_GRAD or name in zero_grad:
            # mathematically zero: softmax shift invariance (key bias) / constant shift in front of BatchNorm
            assert float(mine.flat.g(name).abs().max()) < zero_tol and float(ref_params[name].grad.abs().max()) < zero_tol
            continue
        e = rel_err(mine.flat.g(name).cpu(), ref_params[name].grad)
        names[name] = e
        worst = max(worst, e)
        gr = ref_params[name].grad
        worst_norm = max(worst_norm, float((mine.flat.g(name).cpu().double() - gr.double()).norm() / (gr.double().norm() + 1e-12)))
    out["grad_err"], out["grad_errs"], out["grad_norm_err"] = worst, names, worst_norm
    out["bn_mean_err"] = float((mine.gcn.bn.running_mean.cpu() - ref.gcn.bn.running_mean).abs().max())
    out["bn_var_err"] = float((mine.gcn.bn.running_var.cpu() - ref.gcn.bn.running_var).abs().max())
    dead = [n for n, p in ref.named_parameters() if p.grad is None]
    out["dead_ok"] = all(n.startswith("rnn.0.") for n in dead) and len(dead) > 0
    return out


# ----------------------------------------------------------------------------- golden helpers
def fill_params(model, seed):
    """Same deterministic filler as tests/golden/make_golden.py: fixtures carry the seed, not the weights."""
    g = torch.Generator().manual_seed(seed)
    with torch.no_grad():
        for name, p in sorted(model.named_parameters()):
            bound = 1.0 / (p.shape[-1] ** 0.5) if p.dim() > 1 else 0.1
            p.copy_((torch.rand(p.shape, generator=g) * 2 - 1) * bound)


def check_grad_digest(fix, named_grads, tol, seed=0, keep=512):
    """Compare gradients with a fixture written by make_golden.grad_digest; returns the worst relative error."""
    worst = 0.0
    seen = 0
    for name, gr in named_grads:
        key = name.replace(".", "__")
        flat = gr.detach().cpu().flatten()
        if "grad__" + key in fix.files:
            want = torch.from_numpy(fix["grad__" + key])
            e = rel_err(flat, want)
        elif "gsample__" + key in fix.files:
            g = torch.Generator().manual_seed(seed + flat.numel())
            idx = torch.randint(0, flat.numel(), (keep,), generator=g)
            want = torch.from_numpy(fix["gsample__" + key])
            e = max(rel_err(flat[idx], want),
                    abs(float(flat.double().norm()) - float(fix["gnorm__" + key])) / (float(fix["gnorm__" + key]) + 1e-9))
        else:
            continue
        seen += 1
        assert e < tol, (name, e)
        worst = max(worst, e)
    assert seen > 0
    return worst


def poison_lds_before(monkeypatch, *entry_points):
    """Wrap capi entry points so that every CU's LDS holds NaN bit patterns when their kernel starts: a persistent kernel
    that reads LDS it never wrote (masked operand tails, pad rows) then fails its parity test on every box."""
    from erc_amd import capi

    def wrap(fn):
        def inner(*a, **k):
            capi.poison_lds()
            return fn(*a, **k)
        return inner
    for name in entry_points:
        monkeypatch.setattr(capi, name, wrap(getattr(capi, name)))
