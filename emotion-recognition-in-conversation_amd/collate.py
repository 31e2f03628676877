"""Batch layout of the ERC path (drop-in for ERCCollate, track_mm/mmbase.py:344-455).

Same keys, dtypes and shapes as the reference collate, so a model written
against the reference batch dict runs unchanged:

  attention_mask  float32 [B,T]          (mmbase.py:360-362; stays [B,T] even
                                          when batch_first=False)
  text_length     int64   [B]            (mmbase.py:356)
  {text,audio,visual}_feature  float32 [B,T,d_m] or None when the modality is
                                          off (mmbase.py:439-441)
  input_tensor    float32 [B,T,D], per-utterance concat in the order of the
                                          characters of ``modality`` (:408-415)
  speaker_tensor  int64 [B,T] argmax of the one-hot speakers, zero padded
                                          (:369,418) or float32 one-hot
                                          [B,T,S] when speaker_onehot (:432-433)
  label           int64   [N]            dialogue-major valid utterances (:435)
  sequence tensors are [T,B,...] when batch_first=False (:427-430,439-442).

Unlike the reference (per-utterance torch.from_numpy + stack), rows are
written straight into preallocated padded blocks.
"""
import numpy as np
import torch

_KEY = {"a": "audio", "t": "text", "v": "visual"}


class ERCCollate:
    def __init__(self, params):
        self.batch_first = bool(params.batch_first)
        self.speaker_onehot = bool(params.speaker_onehot)
        self.n_classes = params.n_classes
        self.n_speakers = params.n_speakers
        self.modalities = params.modality

    def __call__(self, samples):
        dialogs = [s[0] if isinstance(s, (list, tuple)) else s for s in samples]
        B = len(dialogs)
        lens = np.array([len(d["text"]) for d in dialogs], dtype=np.int64)
        T = int(lens.max())
        mask = (np.arange(T)[None, :] < lens[:, None]).astype(np.float32)

        blocks = {}
        for m, key in _KEY.items():
            dim = np.asarray(dialogs[0][key]).shape[1]
            blk = np.zeros((B, T, dim), dtype=np.asarray(dialogs[0][key]).dtype)
            for i, d in enumerate(dialogs):
                blk[i, :lens[i]] = d[key]
            blocks[m] = blk
        fused = np.concatenate([blocks[m] for m in self.modalities], axis=-1)

        spk = np.zeros((B, T), dtype=np.int64)
        labels = []
        for i, d in enumerate(dialogs):
            spk[i, :lens[i]] = np.asarray(d["speakers"]).argmax(-1)
            labels.extend(d["label"])

        def seq(a):
            t = torch.from_numpy(a)
            return t if self.batch_first else t.transpose(0, 1).contiguous()

        speaker = torch.from_numpy(spk)
        if not self.batch_first:
            speaker = speaker.transpose(0, 1)
        if self.speaker_onehot:
            speaker = torch.zeros(*speaker.shape, self.n_speakers).scatter_(-1, speaker.unsqueeze(-1), 1)

        data = {
            "attention_mask": torch.from_numpy(mask),
            "text_length": torch.from_numpy(lens),
            "text_feature": seq(blocks["t"]) if "t" in self.modalities else None,
            "audio_feature": seq(blocks["a"]) if "a" in self.modalities else None,
            "visual_feature": seq(blocks["v"]) if "v" in self.modalities else None,
            "input_tensor": seq(fused),
            "speaker_tensor": speaker,
            "label": torch.tensor(labels, dtype=torch.long),
        }
        sent = [d["sentence"] for d in dialogs if d.get("sentence") is not None]
        if sent:
            data["utterance_texts"] = sent
        return data


def batch_to(batch, device, non_blocking=True):
    """Move the tensor entries of a collated batch to ``device``."""
    return {k: (v.to(device, non_blocking=non_blocking) if torch.is_tensor(v) else v) for k, v in batch.items()}
