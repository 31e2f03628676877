"""Readers for the feature pickles the reference trains on: per-dialogue sample dicts with the keys
``ERCCollate`` consumes (``speakers, visual, audio, text, label, sentence[, ids]``).

File formats and name table follow the reference (behaviour, not code):
  * IEMOCAP (COGMEN release): a 9-tuple ``(ids, speakers, labels, text, audio, visual, sentence, train_ids, test_ids)`` of
    dicts keyed by dialogue id, ``cogmen/iemocap/IEMOCAP_features.pkl`` (6 classes) or
    ``cogmen/iemocap_4/IEMOCAP_features_4.pkl`` (4 classes); speakers are 'M' / 'F' strings, turned into 2-way one-hots
    (mmdatasets/datas/mm/iemocap_feature.py:360-458).
  * MELD (MMGCN release): the same tuple plus a tenth unused entry, ``MMGCN/MELD_features_raw.pkl``; speakers already
    9-way one-hot lists, features cast to float32 (mmdatasets/datas/mm/meld_feature.py:12-40).
  * optional per-dialogue override maps in the same directory: ``sbert_map.pkl`` / ``robert_map.pkl`` replace the text
    features, ``tsn_vfeat.pkl`` replaces the visual ones or, for the ``v+`` names, is concatenated to them.
  * dataset names ``{corpus}-{release}[-{text}][-{visual}[-v+]]-{n_classes}`` (mmdatasets/datas/__init__.py:33-69); the
    corpus prefix selects the data root (mmdatasets/const.py:9-10, config.py).

Reference quirk kept on purpose: the ``tsnss`` names resolve to ``tsn_vfeat.pkl`` there as well (the 'tsn' substring
test comes first, iemocap_feature.py:383-386), so they do here.

The pickles are the USER's data files (none ship with either repository); they are read with ``pickle`` as the reference
does.  A dialogue store can then be kept resident on the GPU (``DeviceDialogueStore``) so that batches are padded and
concatenated on the device instead of in DataLoader workers.
"""
import os
import pickle

import numpy as np
import torch

from .params import DATASETS

ROOT_ENV = {"iemocap": "ERC_IEMOCAP_ROOT", "meld": "ERC_MELD_ROOT"}


def data_root(dataset, roots=None):
    """Root directory of a corpus: ``roots[corpus]`` (the reference's config.py mapping) or $ERC_<CORPUS>_ROOT."""
    corpus = dataset.split("-")[0]
    if isinstance(roots, dict) and roots.get(corpus):
        return roots[corpus]
    if isinstance(roots, str) and roots:
        return roots
    env = os.environ.get(ROOT_ENV.get(corpus, ""), "")
    if not env:
        raise FileNotFoundError("no data root for %r: pass --data_root=... or set $%s" % (corpus, ROOT_ENV.get(corpus)))
    return env


def parse_name(dataset):
    """-> dict(corpus, n_classes, text override or '', visual override or '', concat_visual)."""
    if dataset not in DATASETS:
        raise ValueError("dataset %r not in %s" % (dataset, DATASETS))
    parts = dataset.split("-")
    mid = parts[2:-1]
    text = next((t for t in ("sbert", "robert") if t in mid), "")
    visual = next((v for v in ("tsnss", "tsn") if v in mid), "")
    return dict(corpus=parts[0], n_classes=int(parts[-1]), text=text, visual=visual, concat_visual="v+" in mid)


def _load(path):
    with open(path, "rb") as fh:
        return pickle.load(fh)


def read_dialogues(dataset, split="train", roots=None):
    """List of per-dialogue sample dicts of ``split`` ('train' | anything else = the test ids), reference order."""
    spec = parse_name(dataset)
    root = data_root(dataset, roots)
    if spec["corpus"] == "iemocap":
        sub = "cogmen/iemocap" if spec["n_classes"] == 6 else "cogmen/iemocap_4"
        main = "IEMOCAP_features.pkl" if spec["n_classes"] == 6 else "IEMOCAP_features_4.pkl"
    else:
        sub, main = "MMGCN", "MELD_features_raw.pkl"
    folder = os.path.join(root, sub)
    tup = _load(os.path.join(folder, main))
    ids, speakers, labels, text, audio, visual, sentence, train_ids, test_ids = tup[:9]
    if spec["text"]:
        text = _load(os.path.join(folder, spec["text"] + "_map.pkl"))
    if spec["visual"] and spec["corpus"] == "iemocap":
        extra = _load(os.path.join(folder, "tsn_vfeat.pkl"))      # also for the tsnss names (see module docstring)
        visual = {k: np.concatenate([visual[k], extra[k]], axis=1) for k in extra} if spec["concat_visual"] else extra
    out = []
    for k in (train_ids if split == "train" else test_ids):
        if spec["corpus"] == "iemocap":
            sample = {"speakers": [[1, 0] if s == "M" else [0, 1] for s in speakers[k]], "visual": visual[k],
                      "audio": audio[k], "text": text[k], "label": labels[k], "sentence": sentence[k]}
            if spec["n_classes"] == 6:
                sample["ids"] = ids[k]
        else:
            sample = {"ids": ids[k], "speakers": speakers[k], "visual": np.asarray(visual[k], dtype=np.float32),
                      "audio": np.asarray(audio[k], dtype=np.float32), "text": np.asarray(text[k], dtype=np.float32),
                      "label": labels[k], "sentence": sentence[k]}
        out.append(sample)
    return out


class DeviceDialogueStore:
    """All dialogues of a split resident in HBM as flat per-modality row stores; ``batch(indices)`` builds the same
    padded batch dict as ``ERCCollate`` with a handful of device gathers (no DataLoader worker, no host copy per step).
    Layout switches follow the params like ERCCollate's (mmbase.py:344-455)."""

    def __init__(self, dialogues, params, device, dtype=torch.float32):
        self.params, self.device = params, device
        lens = [len(d["label"]) for d in dialogues]
        self.lengths = torch.tensor(lens, dtype=torch.int64)
        self.offsets = torch.zeros(len(lens) + 1, dtype=torch.int64)
        self.offsets[1:] = torch.cumsum(self.lengths, 0)
        order = {"t": "text", "a": "audio", "v": "visual"}
        cat = lambda key: torch.from_numpy(np.concatenate([np.asarray(d[key], dtype=np.float32) for d in dialogues], 0))
        self.feats = {m: cat(order[m]).to(device=device, dtype=dtype) for m in params.modality}
        self.fused = torch.cat([self.feats[m] for m in params.modality], dim=1)           # column order = --modality
        spk = np.concatenate([np.asarray(d["speakers"], dtype=np.int64).argmax(-1) for d in dialogues], 0)
        self.speaker = torch.from_numpy(spk).to(device)
        self.label = torch.from_numpy(np.concatenate([np.asarray(d["label"], dtype=np.int64) for d in dialogues], 0)).to(device)
        self.lengths_dev, self.offsets_dev = self.lengths.to(device), self.offsets.to(device)

    def __len__(self):
        return int(self.lengths.numel())

    def batch(self, indices):
        p, dev = self.params, self.device
        idx = torch.as_tensor(indices, dtype=torch.int64)
        lens = self.lengths[idx]
        B, T, N = int(idx.numel()), int(lens.max()), int(lens.sum())
        idx_d = idx.to(dev)
        lens_d = self.lengths_dev[idx_d]
        t = torch.arange(T, device=dev)
        mask = t[None, :] < lens_d[:, None]                                           # [B, T]
        rows = (self.offsets_dev[idx_d][:, None] + t[None, :]).clamp_(max=self.offsets_dev[-1] - 1)
        pad = lambda src: torch.where(mask[..., None], src[rows], torch.zeros((), dtype=src.dtype, device=dev))
        out = {"attention_mask": mask.float(), "text_length": lens_d, "label": self.label[rows[mask]]}
        spk = torch.where(mask, self.speaker[rows], torch.zeros((), dtype=torch.int64, device=dev))
        if p.speaker_onehot:
            spk = torch.nn.functional.one_hot(spk, p.n_speakers).float()   # padded slots: speaker 0, as ERCCollate
        seq = {"input_tensor": pad(self.fused), "speaker_tensor": spk}
        for m, key in (("t", "text_feature"), ("a", "audio_feature"), ("v", "visual_feature")):
            seq[key] = pad(self.feats[m]) if m in p.modality else None
        if not p.batch_first:
            seq = {k: (v.transpose(0, 1).contiguous() if v is not None else None) for k, v in seq.items()}
        out.update(seq)
        assert out["label"].shape[0] == N
        return out
