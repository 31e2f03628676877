"""Readers for the reference's feature-pickle formats, exercised on synthetic pickles written in the
same tuple layouts (no corpus ships with either repository), and the device-resident dialogue store against ERCCollate."""
import os
import pickle

import numpy as np
import pytest
import torch

from erc_amd import datasets
from erc_amd.collate import ERCCollate
from erc_amd.params import ERCParams


def _write_iemocap(root, n_classes, rng, with_maps=True):
    sub = "cogmen/iemocap" if n_classes == 6 else "cogmen/iemocap_4"
    os.makedirs(os.path.join(root, sub), exist_ok=True)
    keys = ["Ses0%d_impro0%d" % (i, j) for i in range(1, 4) for j in range(1, 3)]
    L = {k: int(rng.integers(1, 9)) for k in keys}
    f = lambda d: {k: rng.standard_normal((L[k], d)).astype(np.float32) for k in keys}
    tup = ({k: ["%s_%03d" % (k, t) for t in range(L[k])] for k in keys},
           {k: ["M" if rng.random() < 0.5 else "F" for _ in range(L[k])] for k in keys},
           {k: [int(rng.integers(0, n_classes)) for _ in range(L[k])] for k in keys},
           f(100), f(100), f(512), {k: ["utt %d" % t for t in range(L[k])] for k in keys},
           keys[:4], keys[4:])
    name = "IEMOCAP_features.pkl" if n_classes == 6 else "IEMOCAP_features_4.pkl"
    with open(os.path.join(root, sub, name), "wb") as fh:
        pickle.dump(tup, fh)
    maps = {}
    if with_maps:
        maps = {"sbert_map.pkl": f(768), "tsn_vfeat.pkl": f(2048)}
        for fn, m in maps.items():
            with open(os.path.join(root, sub, fn), "wb") as fh:
                pickle.dump(m, fh)
    return tup, maps


def _write_meld(root, rng):
    os.makedirs(os.path.join(root, "MMGCN"), exist_ok=True)
    keys = list(range(7))
    L = {k: int(rng.integers(1, 6)) for k in keys}
    f = lambda d: {k: rng.standard_normal((L[k], d)) for k in keys}          # float64 on disk, cast by the reader
    tup = ({k: list(range(L[k])) for k in keys},
           {k: np.eye(9, dtype=np.int64)[rng.integers(0, 9, L[k])].tolist() for k in keys},
           {k: [int(rng.integers(0, 7)) for _ in range(L[k])] for k in keys},
           f(600), f(300), f(342), {k: ["u"] * L[k] for k in keys}, keys[:5], keys[5:], None)
    with open(os.path.join(root, "MMGCN", "MELD_features_raw.pkl"), "wb") as fh:
        pickle.dump(tup, fh)
    return tup


def test_name_grammar_covers_the_registry():
    from erc_amd.params import DATASETS
    for name in DATASETS:
        spec = datasets.parse_name(name)
        assert spec["n_classes"] == int(name.rsplit("-", 1)[1])
        assert (spec["text"] in name if spec["text"] else not any(t in name for t in ("sbert", "robert")))
        assert spec["concat_visual"] == ("v+" in name)
    with pytest.raises(ValueError):
        datasets.parse_name("iemocap-cogmen-9")


@pytest.mark.parametrize("name", ["iemocap-cogmen-6", "iemocap-cogmen-sbert-tsn-v+-6", "iemocap-cogmen-robert-tsnss-4",
                                  "iemocap-cogmen-tsn-4"])
def test_iemocap_reader(tmp_path, name):
    rng = np.random.default_rng(0)
    n = int(name[-1])
    tup, maps = _write_iemocap(str(tmp_path), n, rng)
    if "robert" in name:   # robert map: reuse the sbert one under the other file name
        sub = "cogmen/iemocap" if n == 6 else "cogmen/iemocap_4"
        os.link(os.path.join(tmp_path, sub, "sbert_map.pkl"), os.path.join(tmp_path, sub, "robert_map.pkl"))
    for split, ids in (("train", tup[7]), ("test", tup[8])):
        got = datasets.read_dialogues(name, split, roots={"iemocap": str(tmp_path)})
        assert len(got) == len(ids)
        for d, k in zip(got, ids):
            assert d["label"] == tup[2][k] and d["sentence"] == tup[6][k]
            assert d["speakers"] == [[1, 0] if s == "M" else [0, 1] for s in tup[1][k]]
            np.testing.assert_array_equal(d["audio"], tup[4][k])
            want_t = maps["sbert_map.pkl"][k] if ("sbert" in name or "robert" in name) else tup[3][k]
            np.testing.assert_array_equal(d["text"], want_t)
            if "tsn" in name:
                ex = maps["tsn_vfeat.pkl"][k]
                want_v = np.concatenate([tup[5][k], ex], 1) if "v+" in name else ex
            else:
                want_v = tup[5][k]
            np.testing.assert_array_equal(d["visual"], want_v)
            assert ("ids" in d) == (n == 6)
    # the derived hidden sizes of the run parameters match what the reader delivers
    p = ERCParams().from_args(["--dataset=" + name])
    d0 = datasets.read_dialogues(name, "train", roots=str(tmp_path))[0]
    assert (d0["audio"].shape[1], d0["text"].shape[1], d0["visual"].shape[1]) == \
        (p.hidden_audio, p.hidden_text, p.hidden_visual)


def test_meld_reader(tmp_path, monkeypatch):
    rng = np.random.default_rng(1)
    tup = _write_meld(str(tmp_path), rng)
    monkeypatch.setenv("ERC_MELD_ROOT", str(tmp_path))
    got = datasets.read_dialogues("meld-mmgcn-7", "test")
    assert [d["ids"] for d in got] == [tup[0][k] for k in tup[8]]
    for d, k in zip(got, tup[8]):
        assert d["speakers"] == tup[1][k] and d["text"].dtype == np.float32
        np.testing.assert_allclose(d["visual"], tup[5][k].astype(np.float32))
    monkeypatch.delenv("ERC_MELD_ROOT")
    with pytest.raises(FileNotFoundError):
        datasets.read_dialogues("meld-mmgcn-7", "test")


@pytest.mark.parametrize("flags", [[], ["--modality=tv"], ["--batch_first=False", "--speaker_onehot"], ["--speaker_onehot"]])
def test_device_store_equals_collate(tmp_path, flags):
    rng = np.random.default_rng(2)
    _write_iemocap(str(tmp_path), 6, rng, with_maps=False)
    dialogs = datasets.read_dialogues("iemocap-cogmen-6", "train", roots=str(tmp_path)) + \
        datasets.read_dialogues("iemocap-cogmen-6", "test", roots=str(tmp_path))
    p = ERCParams().from_args(["--dataset=iemocap-cogmen-6"] + flags)
    store = datasets.DeviceDialogueStore(dialogs, p, "cpu")
    idx = [4, 0, 5, 2]
    got = store.batch(idx)
    want = ERCCollate(p)([[dialogs[i]] for i in idx])
    for k in ("attention_mask", "text_length", "label", "input_tensor", "speaker_tensor", "text_feature", "audio_feature",
              "visual_feature"):
        if want[k] is None:
            assert got[k] is None
        else:
            assert got[k].dtype == want[k].dtype and got[k].shape == want[k].shape, k
            assert torch.equal(got[k], want[k]), k
