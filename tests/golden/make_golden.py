#!/usr/bin/env python3
"""Generate golden vectors by running the REFERENCE's own importable leaf
modules on CPU fp32 in the build container.

Run once, here (never on the GPU box: /root/reference does not exist there):

    python tests/golden/make_golden.py [--ref /root/reference]

Outputs small ``.npz`` fixtures next to this file.  Fixtures are DATA (inputs,
parameters, expected outputs); no reference source text is stored.  The stub
modules below are this repo's own code: they stand in for third-party /
framework packages the reference imports at module scope but that the hot-path
``nn.Module``s never touch (lumo, fire-based params, dbrecord, dataset
registry) and for two absent third-party numeric packages, for which only a
minimal shim is provided:
  * ``torch_scatter.scatter_add`` -> ``index_add_`` (needed by the vendored
    models/rgcn.py:12,37);
  * ``torch_geometric.nn`` -> placeholder classes (module construction only;
    their forward is never called here - those operators stay "parity
    unpinned", see oracle/pyg.py).
torch-1.11 -> torch-2.10 drift handled here, not in the reference:
  * contrib/nn.py layers are called one after another by this script instead
    of through nn.TransformerEncoder (which now passes ``is_causal``);
  * mmgcn_models.py:634 ``adj[idx] = dia_sim`` relied on legacy
    sequence-as-tuple indexing; the module is loaded with that one statement
    read as ``adj[tuple(idx)] = dia_sim`` (in memory only).
"""
import argparse
import importlib
import importlib.util
import os
import sys
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, REPO)


# ----------------------------------------------------------------------------
# stubs
# ----------------------------------------------------------------------------
class _Anything:
    """Attribute sink: any attribute / call / subclassing works."""

    def __init__(self, *a, **k):
        pass

    def __call__(self, *a, **k):
        return _Anything()

    def __getattr__(self, name):
        if name.startswith("__"):
            raise AttributeError(name)
        return _Anything()


def _stub_module(name, **attrs):
    m = types.ModuleType(name)
    m.__dict__.update(attrs)
    m.__path__ = []  # behave as a package

    def _getattr(attr, _m=m):
        if attr.startswith("__"):
            raise AttributeError(attr)
        cls = type(attr, (_Anything,), {})
        setattr(_m, attr, cls)
        return cls

    m.__getattr__ = _getattr
    sys.modules[name] = m
    return m


def install_stubs(ref):
    class CollateBase:
        def __init__(self, params=None):
            self.params = params

    def _base():
        class _Base:
            def __init__(self, *a, **k):
                pass

            def iparams(self):
                pass
        return _Base

    cb = _stub_module("lumo.callbacks")
    _stub_module("lumo", CollateBase=CollateBase, Trainer=_base(), TrainerParams=_base(), callbacks=cb)
    _stub_module("lumo.data")
    _stub_module("lumo.core")
    _stub_module("lumo.contrib")
    _stub_module("lumo.contrib.torch")
    spec = importlib.util.spec_from_file_location(
        "lumo.contrib.torch.tensor", os.path.join(ref, "lumo/contrib/torch/tensor.py"))
    real = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(real)  # the real onehot (mmbase.py:15)
    sys.modules["lumo.contrib.torch.tensor"] = real
    _stub_module("dbrecord")
    _stub_module("mmdatasets")
    _stub_module("mmdatasets.erc_dataset")
    _stub_module("mmdatasets.dataset_utils", DataParams=_base())
    _stub_module("models.module_utils", ModelParams=_base())
    _stub_module("contrib.make_optim")

    def scatter_add(src, index, dim=0, out=None, dim_size=None):
        assert dim == 0
        res = torch.zeros((dim_size,) + tuple(src.shape[1:]), dtype=src.dtype)
        return res.index_add_(0, index, src)

    _stub_module("torch_scatter", scatter_add=scatter_add)

    class _Placeholder(torch.nn.Module):
        def __init__(self, *a, **k):
            super().__init__()

    _stub_module("torch_geometric")
    _stub_module("torch_geometric.nn", RGCNConv=_Placeholder, TransformerConv=_Placeholder,
                 GraphConv=_Placeholder)


def load_with_patch(ref, modname, relpath, old, new):
    """Load a reference module from its source text with one statement
    rewritten in memory (torch-1.11 indexing semantics, see module docstring)."""
    path = os.path.join(ref, relpath)
    with open(path) as fh:
        text = fh.read()
    assert text.count(old) == 1, (relpath, old)
    text = text.replace(old, new)
    mod = types.ModuleType(modname)
    mod.__file__ = path
    mod.__package__ = modname.rpartition(".")[0]
    sys.modules[modname] = mod
    exec(compile(text, path, "exec"), mod.__dict__)
    return mod


# ----------------------------------------------------------------------------
# synthetic inputs (our generator; shapes of SURVEY.md 8d)
# ----------------------------------------------------------------------------
def t2n(d):
    return {k: (v.detach().cpu().numpy() if torch.is_tensor(v) else np.asarray(v)) for k, v in d.items()}


def save(name, **arrays):
    path = os.path.join(HERE, name + ".npz")
    np.savez_compressed(path, **arrays)
    print("wrote %-34s %7.1f KB" % (name + ".npz", os.path.getsize(path) / 1024))


def gen_collate(ref):
    from erc_amd.synthetic import make_dialogues
    mmbase = importlib.import_module("track_mm.mmbase")
    for tag, S, dims in (("s2", 2, dict(a=3, t=5, v=4)), ("s9", 9, dict(a=2, t=3, v=6))):
        dialogs = make_dialogues(5, dims, n_speakers=S, n_classes=6, min_len=2, max_len=9, seed=11)
        samples = [[d] for d in dialogs]  # lumo DatasetBuilder yields 1-lists (builder.py:100-101)
        for modality in ("atv", "tv", "a"):
            for batch_first in (True, False):
                for onehot in (False, True):
                    p = types.SimpleNamespace(batch_first=batch_first, speaker_onehot=onehot, n_classes=6,
                                              n_speakers=S, modality=modality)
                    out = mmbase.ERCCollate(p)(samples)
                    arrs = {k: v for k, v in t2n({k: v for k, v in out.items() if torch.is_tensor(v)}).items()}
                    save("collate_%s_%s_bf%d_oh%d" % (tag, modality, batch_first, onehot),
                         seed=11, **{"out_" + k: v for k, v in arrs.items()})


def gen_window_graph(ref):
    """cogmen_utils.batch_graphify (wp=wf=5, S=2) and dgcn-style window 10 with S=9."""
    cu = importlib.import_module("track_mm.cogmen_utils")
    from oracle.graph import relation_table, canonical_edges
    g = torch.Generator().manual_seed(5)
    for tag, S, wp, wf, lens in (
            ("cogmen_s2_w5", 2, 5, 5, [1, 2, 3, 6, 11, 12, 17, 30]),
            ("dgcn_s9_w10", 9, 10, 10, [1, 4, 10, 11, 21, 22, 33]),
            ("asym_s3_w2_4", 3, 2, 4, [1, 2, 5, 9, 14])):
        B, T = len(lens), max(lens)
        lengths = torch.tensor(lens)
        spk = torch.randint(0, S, (B, T), generator=g)
        for b, L in enumerate(lens):
            spk[b, L:] = 0
        feats = torch.randn(B, T, 4, generator=g)
        x, ei, et, cnt = cu.batch_graphify(feats, lengths, spk, wp, wf, relation_table(S))
        ei_s, et_s = canonical_edges(ei.numpy(), et.numpy())
        save("graph_" + tag, lengths=lengths.numpy(), speakers=spk.numpy(), wp=wp, wf=wf, n_speakers=S,
             features=feats.numpy(), x=x.numpy(), edge_index=ei_s, edge_type=et_s, edge_count=cnt.numpy())


def fill_params(model, seed):
    """Deterministic parameter values shared by this script and the tests (tests/util_cases.py has the
    same function): a fixture then only needs the seed, not 15 M parameters."""
    g = torch.Generator().manual_seed(seed)
    with torch.no_grad():
        for name, p in sorted(model.named_parameters()):
            bound = 1.0 / (p.shape[-1] ** 0.5) if p.dim() > 1 else 0.1
            p.copy_((torch.rand(p.shape, generator=g) * 2 - 1) * bound)


def grad_digest(named_grads, seed=0, keep=512):
    """Small gradients verbatim; big ones as (norm, values at fixed pseudo-random positions)."""
    out = {}
    for name, gr in named_grads:
        key = name.replace(".", "__")
        if gr is None:
            continue
        flat = gr.detach().flatten()
        if flat.numel() <= 20000:
            out["grad__" + key] = flat.numpy()
        else:
            g = torch.Generator().manual_seed(seed + flat.numel())
            idx = torch.randint(0, flat.numel(), (keep,), generator=g)
            out["gsample__" + key] = flat[idx].numpy()
            out["gnorm__" + key] = np.array(float(flat.double().norm()))
    return out


def gen_dagerc(ref):
    """DAG-ERC end to end through the reference's own DAGERCModule (track_mm/dagerc.py:73-198):
    adjacency, speaker mask, padded logits, masked CE loss, every gradient."""
    from erc_amd.collate import ERCCollate
    from erc_amd.synthetic import make_dialogues
    import torch.nn.functional as F
    dag = importlib.import_module("track_mm.dagerc")
    for tag, B, dims, S, C, lens in (("small", 3, dict(a=6, t=8, v=6), 2, 6, (2, 9)),
                                     ("s3", 4, dict(a=5, t=4, v=3), 3, 7, (1, 12))):
        D = sum(dims.values())
        dialogs = make_dialogues(B, dims, n_speakers=S, n_classes=C, min_len=lens[0], max_len=lens[1], seed=21,
                                 force_max=True)
        p = types.SimpleNamespace(batch_first=True, speaker_onehot=True, n_classes=C, n_speakers=S, modality="atv")
        batch = ERCCollate(p)([[d] for d in dialogs])
        batch.pop("utterance_texts", None)
        model = dag.DAGERCModule(emb_dim=D, dropout=0.0, n_classes=C, gnn_layers=4)
        fill_params(model, 77)
        model.train()
        logits, _ = model(**batch)
        sel = logits[batch["attention_mask"].bool()]
        loss = F.cross_entropy(sel, batch["label"])
        loss.backward()
        spk = batch["speaker_tensor"].tolist()
        mx = int(batch["text_length"].max())
        adj = model.get_adj_v1(spk, mx)
        s_mask, _ = model.get_s_mask(spk, mx)
        none = [n for n, q in model.named_parameters() if q.grad is None]
        save("dagerc_" + tag, param_seed=77, n_classes=C, n_speakers=S, dims=np.array([dims["a"], dims["t"], dims["v"]]),
             **{"in_" + k: v.numpy() for k, v in batch.items() if torch.is_tensor(v)},
             logits=logits.detach().numpy(), loss=np.array(float(loss)), adj=adj.numpy(), s_mask=s_mask.numpy(),
             grad_none=np.array(none), **grad_digest([(n, q.grad) for n, q in model.named_parameters()]))


def gen_dgcn(ref):
    """DialogueGCN pieces that run from the reference's own files: EdgeAtt + batch_graphify
    (track_mm/dgcn_models.py:51-152) and the vendored RGCNConv (models/rgcn.py:264-355) forward/backward."""
    from oracle.graph import relation_table, canonical_edges
    dm = importlib.import_module("track_mm.dgcn_models")
    rg = importlib.import_module("models.rgcn")
    g = torch.Generator().manual_seed(3)
    for tag, S, lens in (("s2", 2, [3, 12, 25, 1]), ("s9", 9, [7, 33, 2, 21, 16])):
        B, T = len(lens), max(lens)
        lengths = torch.tensor(lens)
        spk = torch.randint(0, S, (B, T), generator=g)
        feats = torch.randn(B, T, 200, generator=g) * 0.5
        for b, L in enumerate(lens):
            feats[b, L:] = 0
            spk[b, L:] = 0
        feats.requires_grad_()
        att = dm.EdgeAtt(200, 10, 10)
        fill_params(att, 5)
        x, ei, en, et, _ = dm.batch_graphify(feats, lengths, spk, 10, 10, relation_table(S), att)
        R = 2 * S * S
        conv = rg.RGCNConv(200, 100, R, num_bases=30)
        fill_params(conv, 6)
        out = conv(x, ei, et, edge_norm=en)
        gout = torch.randn(out.shape, generator=g)
        out.backward(gout)
        ei_s, et_s, en_s = canonical_edges(ei.numpy(), et.numpy(), en.detach().numpy())
        save("dgcn_" + tag, lengths=lengths.numpy(), speakers=spk.numpy(), n_speakers=S, features=feats.detach().numpy(),
             edge_index=ei_s, edge_type=et_s, edge_norm=en_s, rgcn_out=out.detach().numpy(), gout=gout.numpy(),
             att_seed=5, conv_seed=6, dfeatures=feats.grad.numpy(),
             **grad_digest([("edge_att.weight", att.weight.grad)] + [("conv1." + n, q.grad) for n, q in conv.named_parameters()]))


def gen_mmgcn(ref):
    """MMGCN end to end through the reference's own MMGCNModule (track_mm/mmgcn.py:56-123), eval mode (every
    dropout off): normalised big adjacency, logits, CE loss, every gradient."""
    from erc_amd.collate import ERCCollate
    from erc_amd.synthetic import make_dialogues
    import torch.nn.functional as F
    load_with_patch(ref, "track_mm.mmgcn_models", "track_mm/mmgcn_models.py", "adj[idx] = dia_sim",
                    "adj[tuple(idx)] = dia_sim")
    mm = importlib.import_module("track_mm.mmgcn")
    for tag, modality, B, dims, S, C, lens in (("atv", "atv", 3, dict(a=10, t=12, v=8), 2, 6, (2, 7)),
                                               ("tv_s3", "tv", 4, dict(a=4, t=9, v=6), 3, 7, (1, 6))):
        dialogs = make_dialogues(B, dims, n_speakers=S, n_classes=C, min_len=lens[0], max_len=lens[1], seed=31,
                                 force_max=True)
        p = types.SimpleNamespace(batch_first=False, speaker_onehot=True, n_classes=C, n_speakers=S, modality=modality)
        batch = ERCCollate(p)([[d] for d in dialogs])
        batch.pop("utterance_texts", None)
        model = mm.MMGCNModule(hidden_text=dims["t"], hidden_visual=dims["v"], hidden_audio=dims["a"], n_speakers=S,
                               n_classes=C, modals=modality)
        fill_params(model, 91)
        model.eval()
        captured = {}
        orig = model.graph_model.create_big_adj

        def spy(*a, **k):
            captured["adj"] = orig(*a, **k)
            return captured["adj"]
        model.graph_model.create_big_adj = spy
        logits, _ = model(**batch)
        loss = F.cross_entropy(logits, batch["label"])
        loss.backward()
        none = [n for n, q in model.named_parameters() if q.grad is None]
        save("mmgcn_" + tag, param_seed=91, n_classes=C, n_speakers=S, dims=np.array([dims["a"], dims["t"], dims["v"]]),
             modality=np.array(modality), adj=captured["adj"].detach().numpy(),
             **{"in_" + k: v.numpy() for k, v in batch.items() if torch.is_tensor(v)},
             logits=logits.detach().numpy(), loss=np.array(float(loss)), grad_none=np.array(none),
             **grad_digest([(n, q.grad) for n, q in model.named_parameters()]))


def gen_encoder(ref):
    """The vendored encoder layer (contrib/nn.py:206-305 over its MultiheadAttention :24-203), two layers called one
    after the other as cogmen.py:99-101 stacks them (not through nn.TransformerEncoder, see module docstring), eval
    mode (dropout off), with and without a key-padding mask: outputs + gradients of a weighted output sum."""
    cn = importlib.import_module("contrib.nn")
    g = torch.Generator().manual_seed(41)
    for tag, B, T, D, nhead, masked in (("d24", 3, 13, 24, 6, False), ("d24_mask", 3, 13, 24, 6, True),
                                        ("d48_h8", 2, 9, 48, 8, True), ("d712", 2, 20, 712, 8, True),
                                        ("d712_nomask", 2, 20, 712, 8, False), ("d1380", 1, 16, 1380, 6, False)):
        class Stack(torch.nn.Module):
            def __init__(self):
                super().__init__()
                self.layers = torch.nn.ModuleList([cn.TransformerEncoderLayer(d_model=D, nhead=nhead, dropout=0.5,
                                                                              batch_first=True) for _ in range(2)])
        enc = Stack()
        fill_params(enc, 43)
        with torch.no_grad():          # LayerNorm gains around one (the filler gives +-0.1)
            for lyr in enc.layers:
                lyr.norm1.weight.add_(1.0), lyr.norm2.weight.add_(1.0)
        enc.eval()
        x = torch.randn(B, T, D, generator=g).requires_grad_()
        lengths = torch.randint(1, T + 1, (B,), generator=g)
        lengths[0] = T
        pad = (torch.arange(T)[None, :] >= lengths[:, None]) if masked else None
        h = x
        for lyr in enc.layers:
            h = lyr(h, src_key_padding_mask=pad)
        valid = ~pad if masked else torch.ones(B, T, dtype=torch.bool)
        w = torch.randn(B, T, D, generator=g) * valid[..., None]
        (h * w).sum().backward()
        save("encoder_" + tag, param_seed=43, nhead=nhead, x=x.detach().numpy(), lengths=lengths.numpy(),
             masked=np.array(masked), out=h.detach().numpy(), w=w.numpy(), dx=x.grad.numpy(),
             **grad_digest([(n, q.grad) for n, q in enc.named_parameters()]))


def gen_dgcn_leaves(ref):
    """DialogueGCN leaf modules that run from the reference's own file: SeqContext (packed 2-layer BiLSTM,
    dgcn_models.py:10-33) and Classifier (dgcn_models.py:155-170), eval mode, outputs + gradients."""
    dm = importlib.import_module("track_mm.dgcn_models")
    g = torch.Generator().manual_seed(51)
    for tag, D, lens in (("d30", 30, [5, 12, 1, 9]), ("d1242", 1242, [7, 33, 2, 16])):
        B, T = len(lens), max(lens)
        lengths = torch.tensor(lens)
        x = torch.randn(B, T, D, generator=g) * 0.5
        for b, L in enumerate(lens):
            x[b, L:] = 0
        x.requires_grad_()
        rnn = dm.SeqContext(D, 200)
        fill_params(rnn, 53)
        rnn.eval()
        out = rnn(lengths, x)
        w = torch.randn(out.shape, generator=g)
        (out * w).sum().backward()
        save("seqcontext_" + tag, param_seed=53, x=x.detach().numpy(), lengths=lengths.numpy(), out=out.detach().numpy(),
             w=w.numpy(), dx=x.grad.numpy(), **grad_digest([("rnn." + n, q.grad) for n, q in rnn.named_parameters()]))
    for tag, C, N in (("c6", 6, 37), ("c7", 7, 11)):
        clf = dm.Classifier(300, 100, C, 0.4)
        fill_params(clf, 57)
        clf.eval()
        h = torch.randn(N, 300, generator=g).requires_grad_()
        logits = clf(h, None)
        w = torch.randn(logits.shape, generator=g)
        (logits * w).sum().backward()
        none = [n for n, q in clf.named_parameters() if q.grad is None]
        save("classifier_" + tag, param_seed=57, n_classes=C, h=h.detach().numpy(), logits=logits.detach().numpy(),
             w=w.numpy(), dh=h.grad.numpy(), grad_none=np.array(none),
             **grad_digest([("clf." + n, q.grad) for n, q in clf.named_parameters()]))


GENERATORS = {"collate": gen_collate, "window_graph": gen_window_graph, "dagerc": gen_dagerc, "dgcn": gen_dgcn,
              "mmgcn": gen_mmgcn, "encoder": gen_encoder, "dgcn_leaves": gen_dgcn_leaves}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--ref", default="/root/reference")
    ap.add_argument("--only", default=None)
    args = ap.parse_args()
    torch.set_num_threads(4)
    install_stubs(args.ref)
    # the repo has its own ``track_mm`` plugin package: bind the reference's
    # packages explicitly so that ``track_mm.*`` / ``contrib.*`` / ``models.*``
    # resolve to /root/reference inside THIS process only.
    for pkg in ("track_mm", "contrib", "models"):
        m = types.ModuleType(pkg)
        m.__path__ = [os.path.join(args.ref, pkg)]
        sys.modules[pkg] = m
    for name, fn in GENERATORS.items():
        if args.only and name not in args.only.split(","):
            continue
        print("==", name)
        fn(args.ref)


if __name__ == "__main__":
    main()
