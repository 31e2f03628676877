"""Seeded synthetic IEMOCAP-/MELD-shaped dialogues.

There are no dataset pickles in the build or bench environment, so every test
and benchmark runs on dialogues drawn here.  A sample has the schema the
reference's readers produce (mmdatasets/datas/mm/iemocap_feature.py:401 and
meld_feature.py:12-40): per-utterance ``audio``/``text``/``visual`` float32
rows, ``speakers`` as one-hot lists, ``label`` as a list of ints.
"""
import numpy as np


def make_dialogues(n, dims, n_speakers=2, n_classes=6, min_len=20, max_len=110, seed=1,
                   force_max=False, p_switch=0.7):
    """``dims`` = dict(a=, t=, v=).  Lengths uniform in [min_len, max_len];
    speakers switch with probability ``p_switch`` (two-party) or are uniform
    (multi-party); labels uniform.  ``force_max`` pins dialogue 0 at max_len so
    that a bench batch always has T = max_len."""
    rng = np.random.RandomState(seed)
    out = []
    for i in range(n):
        L = int(rng.randint(min_len, max_len + 1))
        if force_max and i == 0:
            L = max_len
        if n_speakers == 2:
            flips = rng.rand(L) < p_switch
            spk = np.cumsum(flips) % 2
        else:
            spk = rng.randint(0, n_speakers, size=L)
        onehot = np.eye(n_speakers, dtype=np.int64)[spk]
        out.append({
            "speakers": onehot.tolist(),
            "audio": rng.standard_normal((L, dims["a"])).astype(np.float32),
            "text": rng.standard_normal((L, dims["t"])).astype(np.float32),
            "visual": rng.standard_normal((L, dims["v"])).astype(np.float32),
            "label": rng.randint(0, n_classes, size=L).tolist(),
            "sentence": ["utt %d.%d" % (i, k) for k in range(L)],
        })
    return out
