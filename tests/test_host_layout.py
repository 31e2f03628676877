"""Host logic vs golden vectors produced by the reference's own ERCCollate /
batch_graphify (tests/golden/make_golden.py).  CPU only."""
import glob
import os
import types

import numpy as np
import pytest
import torch

from erc_amd.collate import ERCCollate
from erc_amd.params import ERCParams, DATASETS
from erc_amd.synthetic import make_dialogues
from oracle import graph as og

GOLDEN = os.path.join(os.path.dirname(__file__), "golden")


@pytest.mark.parametrize("path", sorted(glob.glob(os.path.join(GOLDEN, "collate_*.npz"))))
def test_collate_matches_reference(path):
    tag, modality, bf, oh = os.path.basename(path)[len("collate_"):-4].split("_")
    S, dims = (2, dict(a=3, t=5, v=4)) if tag == "s2" else (9, dict(a=2, t=3, v=6))
    dialogs = make_dialogues(5, dims, n_speakers=S, n_classes=6, min_len=2, max_len=9, seed=11)
    p = types.SimpleNamespace(batch_first=bf == "bf1", speaker_onehot=oh == "oh1", n_classes=6,
                              n_speakers=S, modality=modality)
    out = ERCCollate(p)([[d] for d in dialogs])
    ref = np.load(path)
    keys = [k[4:] for k in ref.files if k.startswith("out_")]
    assert keys
    for k in keys:
        got = out[k]
        assert got is not None, k
        assert tuple(got.shape) == ref["out_" + k].shape, k
        assert str(got.dtype).replace("torch.", "") == str(ref["out_" + k].dtype), k
        np.testing.assert_array_equal(got.numpy(), ref["out_" + k])
    for m, key in (("t", "text_feature"), ("a", "audio_feature"), ("v", "visual_feature")):
        assert (out[key] is None) == (m not in modality)


@pytest.mark.parametrize("name", ["graph_cogmen_s2_w5", "graph_dgcn_s9_w10", "graph_asym_s3_w2_4"])
def test_window_graph_oracle_matches_reference(golden, name):
    g = golden(name)
    lengths, spk = torch.from_numpy(g["lengths"]), torch.from_numpy(g["speakers"])
    wp, wf, S = int(g["wp"]), int(g["wf"]), int(g["n_speakers"])
    x, ei, et, cnt = og.window_graph_loop(torch.from_numpy(g["features"]), lengths, spk, wp, wf, S)
    ei_s, et_s = og.canonical_edges(ei.numpy(), et.numpy())
    np.testing.assert_array_equal(ei_s, g["edge_index"])
    np.testing.assert_array_equal(et_s, g["edge_type"])
    np.testing.assert_array_equal(cnt.numpy(), g["edge_count"])
    np.testing.assert_array_equal(x.numpy(), g["x"])
    # closed form (what the HIP builder implements) == reference edges, already canonical
    ei_c, et_c = og.window_graph_closed_form(g["lengths"], g["speakers"], wp, wf, S)
    np.testing.assert_array_equal(ei_c, g["edge_index"])
    np.testing.assert_array_equal(et_c, g["edge_type"])


def test_dataset_dims():
    cases = {
        ("iemocap-cogmen-6", "atv"): (100, 100, 512, 712, 6, 2),
        ("iemocap-cogmen-sbert-6", "atv"): (100, 768, 512, 1380, 6, 2),
        ("iemocap-cogmen-4", "tv"): (100, 100, 512, 612, 4, 2),
        ("iemocap-cogmen-robert-tsn-v+-4", "atv"): (100, 768, 2560, 3428, 4, 2),
        ("iemocap-cogmen-tsn-6", "v"): (100, 100, 2048, 2048, 6, 2),
        ("meld-mmgcn-7", "atv"): (300, 600, 342, 1242, 7, 9),
        ("meld-mmgcn-sbert-7", "at"): (300, 768, 342, 1068, 7, 9),
    }
    for (ds, mod), (a, t, v, D, C, S) in cases.items():
        p = ERCParams().from_args(["--dataset=" + ds, "--modality=" + mod])
        assert (p.hidden_audio, p.hidden_text, p.hidden_visual, p.hidden_all, p.n_classes, p.n_speakers) == \
            (a, t, v, D, C, S), ds
    assert len(DATASETS) == 26


def test_cli_grammar():
    p = ERCParams().from_args(["--train.batch_size=16", "--optim.lr=0.01", "--reimplement", "--device=cpu",
                               "--dataset=iemocap-cogmen-4"])
    assert p.train.batch_size == 16 and p.optim.lr == 0.01 and p.reimplement is True and p.device == "cpu"
    with pytest.raises(ValueError):
        ERCParams().from_args(["--modality=xyz"])
    with pytest.raises(ValueError):
        ERCParams().from_args(["--dataset=nope-3"])
