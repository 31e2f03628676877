"""Oracle: MMGCN forward / loss on CPU, fp32 (track_mm/mmgcn.py:56-157, track_mm/mmgcn_models.py:8-39,344-394,
493-646, track_mm/mmgcn_utils.py:5-21).

Structure-faithful: per-dialogue python loops for the (modalities*N)^2 dense adjacency, dense matmul of that
adjacency in each of the 64 GCNII layers, BiLSTM over the padded (unpacked) text block.  Parameters that the
reference constructs but never uses (att_model.*, gatedatt.*, graph_model.{a_fc,v_fc,l_fc,feature_fc,final_fc,
modal_embeddings,*_spk_embs}) are kept in the state dict (grad stays None).
PINNED: tests/golden/mmgcn_*.npz hold logits / loss / gradients of the reference's own MMGCNModule (loaded with
the torch-1.11 indexing statement of mmgcn_models.py:634 read as ``adj[tuple(idx)] = dia_sim``).
"""
import math

import numpy as np
import torch
from torch import nn
from torch.nn import functional as F


class GraphConvolution(nn.Module):
    """mmgcn_models.py:8-39, variant=True."""

    def __init__(self, nhidden):
        super().__init__()
        self.weight = nn.Parameter(torch.empty(2 * nhidden, nhidden))
        stdv = 1.0 / math.sqrt(nhidden)
        self.weight.data.uniform_(-stdv, stdv)

    def forward(self, x, adj, h0, lamda, alpha, l):
        theta = math.log(lamda / l + 1)
        hi = adj @ x                                         # dense (modalities*N)^2 product, as the reference
        support = torch.cat([hi, h0], 1)
        r = (1 - alpha) * hi + alpha * h0
        return theta * (support @ self.weight) + (1 - theta) * r


class GCNII(nn.Module):
    """GCNII_lyc, return_feature=True, use_residue=True (mmgcn_models.py:344-394)."""

    def __init__(self, nfeat, nlayers, nhidden, dropout, lamda, alpha):
        super().__init__()
        self.convs = nn.ModuleList([GraphConvolution(nhidden) for _ in range(nlayers)])
        self.fcs = nn.ModuleList([nn.Linear(nfeat, nhidden)])
        self.dropout, self.lamda, self.alpha = dropout, lamda, alpha

    def forward(self, x, adj):
        x = F.dropout(x, self.dropout, training=self.training)
        h = F.relu(self.fcs[0](x))
        h0 = h
        for i, conv in enumerate(self.convs):
            h = F.dropout(h, self.dropout, training=self.training)
            h = F.relu(conv(h, adj, h0, self.lamda, self.alpha, i + 1))
        h = F.dropout(h, self.dropout, training=self.training)
        return torch.cat([x, h], dim=-1)


def sim_block(x):
    """1 - acos(0.99999 cos)/pi between all rows of x (mmgcn_models.py:604-610)."""
    n = x / torch.sqrt((x * x).sum(1, keepdim=True))
    return 1 - torch.acos((n @ n.t()) * 0.99999) / np.pi


def big_adjacency(feats, dia_len):
    """create_big_adj (mmgcn_models.py:582-646); modality order is the order of ``feats``."""
    M, N = len(feats), feats[0].shape[0]
    adj = torch.zeros(M * N, M * N)
    start = 0
    for L in dia_len:
        L = int(L)
        for m in range(M):
            for n in range(M):
                ms, ns = start + N * m, start + N * n
                if m == n:
                    adj[ms:ms + L, ns:ns + L] = sim_block(feats[m][start:start + L])
                else:
                    a, b = feats[m][start:start + L], feats[n][start:start + L]
                    cos = ((a / torch.sqrt((a * a).sum(1, keepdim=True))) * (b / torch.sqrt((b * b).sum(1, keepdim=True)))).sum(1)
                    idx = torch.arange(L)
                    adj[ms + idx, ns + idx] = 1 - torch.acos(cos * 0.99999) / np.pi
        start += L
    d = adj.sum(1)
    D = torch.diag(torch.pow(d, -0.5))
    return D.mm(adj).mm(D)


class GraphModel(nn.Module):
    """MMGCN (mmgcn_models.py:493-580): use_speaker=True, use_modal=False, return_feature=True, use_residue=True."""

    def __init__(self, n_dim, nlayers, nhidden, nclass, dropout, lamda, alpha, n_speakers, modals):
        super().__init__()
        self.graph_net = GCNII(n_dim, nlayers, nhidden, dropout, lamda, alpha)
        self.a_fc, self.v_fc, self.l_fc = nn.Linear(n_dim, n_dim), nn.Linear(n_dim, n_dim), nn.Linear(n_dim, n_dim)
        self.feature_fc = nn.Linear(n_dim * 3 + nhidden * 3, nhidden)
        self.final_fc = nn.Linear(nhidden, nclass)
        self.modal_embeddings = nn.Embedding(3, n_dim)
        self.speaker_embeddings = nn.Embedding(n_speakers, n_dim)
        self.a_spk_embs, self.v_spk_embs, self.l_spk_embs = (nn.Embedding(n_speakers, n_dim) for _ in range(3))
        self.modals = modals

    def forward(self, a, v, l, dia_len, qmask):
        qm = torch.cat([qmask[:x, i, :] for i, x in enumerate(dia_len)], dim=0)
        spk = torch.argmax(qm, dim=-1)
        if "t" in self.modals:
            l = l + self.speaker_embeddings(spk)              # reference does it in place on the flattened copy (:545)
        feats = [f for f, m in ((a, "a"), (v, "v"), (l, "t")) if m in self.modals]
        adj = big_adjacency(feats, dia_len)
        self.last_adj = adj
        out = self.graph_net(torch.cat(feats, dim=0), adj)
        N = feats[0].shape[0]
        return torch.cat([out[N * i:N * (i + 1)] for i in range(len(feats))], dim=-1)


class _Holder(nn.Module):
    """Registers parameters by dotted name (constructed-but-unused sub-modules of the reference)."""

    def __init__(self, table):
        super().__init__()
        for name, shape in table:
            head, _, rest = name.partition(".")
            if rest:
                if not hasattr(self, head):
                    setattr(self, head, _Holder([]))
                getattr(self, head)._add(rest, shape)
            else:
                self._add(name, shape)

    def _add(self, name, shape):
        head, _, rest = name.partition(".")
        if rest:
            if not hasattr(self, head):
                setattr(self, head, _Holder([]))
            getattr(self, head)._add(rest, shape)
        else:
            self.register_parameter(name, nn.Parameter(torch.zeros(*shape).uniform_(-0.05, 0.05)))


def unused_tables(n_classes, n_speakers, n_modals):
    """Names / shapes of the never-used parameters (captured from the reference's MMGCNModule state dict)."""
    att = [("scalar.weight", (200, 200)), ("matchatt.transform.weight", (200, 200)), ("matchatt.transform.bias", (200,)),
           ("simpleatt.scalar.weight", (1, 200)), ("att.weight", (400,)), ("att.w_k.weight", (200, 200)),
           ("att.w_k.bias", (200,)), ("att.w_q.weight", (200, 200)), ("att.w_q.bias", (200,)),
           ("att.proj.weight", (200, 200)), ("att.proj.bias", (200,))]
    gated = []
    for n in ("l", "v", "a"):
        gated += [("transform_%s.weight" % n, (200, 400)), ("transform_%s.bias" % n, (200,))]
    for n in ("av", "al", "vl"):
        gated += [("transform_%s.weight" % n, (1, 1200)), ("transform_%s.bias" % n, (1,))]
    return att, gated


class MMGCNOracle(nn.Module):
    def __init__(self, hidden_text=100, hidden_visual=512, hidden_audio=100, n_speakers=2, n_classes=7, modals="atv"):
        super().__init__()
        self.modals = modals
        self.linear_l = nn.Linear(hidden_text, 200)
        self.lstm_l = nn.LSTM(200, 100, 2, bidirectional=True, dropout=0.4)
        self.linear_a = nn.Linear(hidden_audio, 200)
        self.linear_v = nn.Linear(hidden_visual, 200)
        att, gated = unused_tables(n_classes, n_speakers, len(modals))
        self.att_model = _Holder(att)
        self.graph_model = GraphModel(200, 64, 200, n_classes, 0.4, 0.5, 0.1, n_speakers, modals)
        self.gatedatt = _Holder(gated)
        self.dropout_ = nn.Dropout(0.4)
        self.smax_fc = nn.Linear(400 * len(modals), n_classes)

    @staticmethod
    def flatten(features, lengths):
        """simple_batch_graphify (mmgcn_utils.py:5-21): valid rows, dialogue-major."""
        return torch.cat([features[:lengths[j], j, :] for j in range(features.size(1))], dim=0)

    def forward(self, text_feature=None, audio_feature=None, visual_feature=None, speaker_tensor=None,
                text_length=None, **kwargs):
        fa = fv = fl = []
        if "a" in self.modals:
            fa = self.flatten(self.linear_a(audio_feature), text_length)
        if "v" in self.modals:
            fv = self.flatten(self.linear_v(visual_feature), text_length)
        if "t" in self.modals:
            out, _ = self.lstm_l(self.linear_l(text_feature))      # unpacked: runs over the padded tail too
            fl = self.flatten(out, text_length)
        feat = self.graph_model(fa, fv, fl, text_length, speaker_tensor)
        return self.smax_fc(F.relu(self.dropout_(feat))), None


def mmgcn_train_step(model, optim, batch):
    """mmgcn.py:141-157."""
    logits, _ = model(**batch)
    loss = F.cross_entropy(logits, batch["label"])
    optim.zero_grad()
    loss.backward()
    optim.step()
    return loss.detach(), torch.eq(logits.argmax(-1), batch["label"]).float().mean()
