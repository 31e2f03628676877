"""Oracle: host-side dialogue-graph construction (integer work, bit-exact).

Restates, with the reference's own structure (python loops, one ``.item()``
per endpoint) so that its cost is representative of the reference CPU path:

  * window edges + relation ids  -- track_mm/cogmen_utils.py:109-172 (COGMEN),
    track_mm/dgcn_models.py:51-118 (DialogueGCN); relation-id table
    track_mm/cogmen.py:124-129, track_mm/dgcn.py:72-77
  * DAG predecessor adjacency    -- track_mm/dagerc.py:109-129
  * same-speaker mask            -- track_mm/dagerc.py:131-154

The reference's edge ORDER is CPython-set iteration order and is not a
contract; ``canonical_edges`` sorts (target-major, then source) which is the
order the HIP graph builder emits.
"""
import numpy as np
import torch


def relation_table(n_speakers):
    """'jk0'/'jk1' -> id, insertion order j, k, dir (cogmen.py:124-129)."""
    table = {}
    for a in range(n_speakers):
        for b in range(n_speakers):
            for d in "01":
                table["%d%d%s" % (a, b, d)] = len(table)
    return table


def window_pairs(length, wp, wf):
    """All (j, k) with k in [j-wp, j+wf] clipped to the dialogue
    (cogmen_utils.py:147-172).  -1 means unbounded on that side."""
    pairs = set()
    for j in range(length):
        lo = 0 if wp == -1 else max(0, j - wp)
        hi = length if wf == -1 else min(length, j + wf + 1)
        for k in range(lo, hi):
            pairs.add((j, k))
    return list(pairs)


def window_graph_loop(features, lengths, speaker_tensor, wp, wf, n_speakers):
    """Reference-shaped batch_graphify (cogmen_utils.py:109-144): returns
    node rows, edge_index [2,E] (row0 = j "source", row1 = k "target"),
    edge_type [E], per-dialogue edge counts.  Deliberately loop-y."""
    table = relation_table(n_speakers)
    rows, e_src, e_dst, e_typ, e_cnt = [], [], [], [], []
    base = 0
    for b in range(features.size(0)):
        L = lengths[b].item()
        rows.append(features[b, :L, :])
        pairs = window_pairs(L, wp, wf)
        e_cnt.append(len(pairs))
        for (j, k) in pairs:
            e_src.append(j + base)
            e_dst.append(k + base)
            sj = speaker_tensor[b, j].item()
            sk = speaker_tensor[b, k].item()
            e_typ.append(table["%d%d%s" % (sj, sk, "0" if j < k else "1")])
        base += L
    x = torch.cat(rows, dim=0)
    edge_index = torch.tensor([e_src, e_dst], dtype=torch.long)
    edge_type = torch.tensor(e_typ, dtype=torch.long)
    return x, edge_index, edge_type, torch.tensor(e_cnt, dtype=torch.long)


def window_graph_closed_form(lengths, speakers, wp, wf, n_speakers):
    """Vectorised numpy restatement of the same edge set in canonical
    (target-major, then source) order -- SURVEY.md Appendix C.  Used to test
    the loop version against itself at sizes where loops are slow."""
    lengths = np.asarray(lengths, dtype=np.int64)
    speakers = np.asarray(speakers, dtype=np.int64)
    src, dst, typ = [], [], []
    base = 0
    for b, L in enumerate(lengths):
        k = np.arange(L)
        # edge j->k exists iff k-wf <= j <= k+wp (inverse of j-wp <= k <= j+wf)
        lo = np.maximum(0, k - wf) if wf != -1 else np.zeros(L, np.int64)
        hi = np.minimum(L - 1, k + wp) if wp != -1 else np.full(L, L - 1)
        for kk in range(L):
            j = np.arange(lo[kk], hi[kk] + 1)
            src.append(j + base)
            dst.append(np.full(j.shape, kk + base))
            sj = speakers[b, j]
            sk = speakers[b, kk]
            typ.append(2 * (sj * n_speakers + sk) + (j >= kk).astype(np.int64))
        base += L
    if not src:
        z = np.zeros(0, np.int64)
        return np.stack([z, z]), z
    return np.stack([np.concatenate(src), np.concatenate(dst)]), np.concatenate(typ)


def canonical_edges(edge_index, edge_type=None, extra=None):
    """Sort edges by (target, source); returns sorted copies."""
    ei = np.asarray(edge_index)
    order = np.lexsort((ei[0], ei[1]))
    out = [ei[:, order]]
    if edge_type is not None:
        out.append(np.asarray(edge_type)[order])
    if extra is not None:
        out.append(np.asarray(extra)[order])
    return out if len(out) > 1 else out[0]


def dag_adjacency_loop(speakers, max_len, windowp=1):
    """dagerc.py:109-129.  ``speakers`` is a nested list [B][T] of hashable /
    comparable speaker descriptors (the reference passes one-hot lists)."""
    out = []
    for spk in speakers:
        a = torch.zeros(max_len, max_len)
        for i, s in enumerate(spk):
            seen = 0
            for j in range(i - 1, -1, -1):
                a[i, j] = 1
                if spk[j] == s:
                    seen += 1
                    if seen == windowp:
                        break
        out.append(a)
    return torch.stack(out)


def speaker_mask_loop(speakers, max_len):
    """dagerc.py:131-154 (only the int64 mask; the one-hot twin is discarded
    by the caller, dagerc.py:161)."""
    out = []
    for spk in speakers:
        m = torch.zeros(max_len, max_len, dtype=torch.long)
        for i in range(len(spk)):
            for j in range(len(spk)):
                if spk[i] == spk[j]:
                    m[i, j] = 1
        out.append(m)
    return torch.stack(out)


def dag_pred_closed_form(speaker_idx):
    """p[b,i] = largest j<i with the same speaker, else -1 (Appendix C).
    Row i of adj is ones on [max(p,0), i-1]."""
    s = np.asarray(speaker_idx)
    B, T = s.shape
    p = np.full((B, T), -1, dtype=np.int64)
    for b in range(B):
        last = {}
        for i in range(T):
            p[b, i] = last.get(int(s[b, i]), -1)
            last[int(s[b, i])] = i
    return p
