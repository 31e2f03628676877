"""Oracle: DialogueGCN forward / loss on CPU, fp32 (track_mm/dgcn.py:53-134, track_mm/dgcn_models.py:10-170,
models/rgcn.py:264-355).

Structure-faithful: packed BiLSTM, per-node python softmax loop of EdgeAtt (dgcn_models.py:132-152), per-edge
python graph construction with one ``.item()`` per endpoint (dgcn_models.py:51-92), the vendored RGCNConv's
per-edge weight materialisation ``index_select(w, 0, edge_type)`` + ``bmm`` (models/rgcn.py:339-343).
PINNED by tests/golden/dgcn_*.npz: EdgeAtt + batch_graphify (edge_norm, edge order canonicalised) and the
vendored RGCNConv forward/backward run from the reference's own files.  ``GraphConv`` is torch_geometric
(absent) -> restated in oracle/pyg.py, parity unpinned for that operator.
"""
import torch
from torch import nn
from torch.nn import functional as F
from torch.nn.utils.rnn import pack_padded_sequence, pad_packed_sequence

from .graph import relation_table, window_pairs
from .pyg import GraphConvAdd, scatter_sum


class SeqContext(nn.Module):
    """dgcn_models.py:10-33 (lstm flavour)."""

    def __init__(self, u_dim, g_dim, dropout=0.4):
        super().__init__()
        self.rnn = nn.LSTM(u_dim, g_dim // 2, dropout=dropout, bidirectional=True, num_layers=2, batch_first=True)

    def forward(self, lengths, x):
        packed = pack_padded_sequence(x, lengths.cpu(), batch_first=True, enforce_sorted=False)
        out, _ = self.rnn(packed, None)
        return pad_packed_sequence(out, batch_first=True)[0]


class EdgeAtt(nn.Module):
    """dgcn_models.py:121-152: alpha[j, k] = softmax over the window of SOURCE j of (W x_k) . x_j."""

    def __init__(self, g_dim, wp, wf):
        super().__init__()
        self.wp, self.wf = wp, wf
        self.weight = nn.Parameter(torch.zeros(g_dim, g_dim))
        self.weight.data.normal_(0, 2.0 / (g_dim + g_dim))

    def forward(self, node_features, lengths):
        B, T = node_features.shape[:2]
        att = torch.matmul(self.weight[None, None], node_features.unsqueeze(-1)).squeeze(-1)
        out = []
        for b in range(B):
            L = lengths[b].item()
            alpha = torch.zeros(T, max(T, 110))            # the reference hard-codes 110 columns (:140)
            for j in range(L):
                s, e = max(j - self.wp, 0), min(j + self.wf, L - 1)
                alpha[j, s:e + 1] = F.softmax(att[b, s:e + 1] @ node_features[b, j], dim=-1)
            out.append(alpha)
        return out


def dgcn_graphify(features, lengths, speakers, wp, wf, n_speakers, att_model):
    """dgcn_models.py:51-92."""
    table = relation_table(n_speakers)
    weights = att_model(features, lengths)
    rows, src, dst, norm, typ = [], [], [], [], []
    base = 0
    for b in range(features.size(0)):
        L = lengths[b].item()
        rows.append(features[b, :L])
        for (j, k) in window_pairs(L, wp, wf):
            src.append(j + base)
            dst.append(k + base)
            norm.append(weights[b][j, k])
            sj, sk = speakers[b, j].item(), speakers[b, k].item()
            typ.append(table["%d%d%s" % (sj, sk, "0" if j < k else "1")])
        base += L
    return (torch.cat(rows, 0), torch.tensor([src, dst], dtype=torch.long), torch.stack(norm),
            torch.tensor(typ, dtype=torch.long))


class RGCNConvBasis(nn.Module):
    """Vendored torch_geometric-1.4.2 RGCNConv with basis decomposition and edge_norm (models/rgcn.py:264-355):
    message = norm_e * x_src W_{type(e)}, W_r = sum_b att[r,b] basis[b]; aggregate ADD at the target;
    + x root + bias."""

    def __init__(self, cin, cout, R, num_bases):
        super().__init__()
        self.cin, self.cout, self.R, self.nb = cin, cout, R, num_bases
        self.basis = nn.Parameter(torch.empty(num_bases, cin, cout))
        self.att = nn.Parameter(torch.empty(R, num_bases))
        self.root = nn.Parameter(torch.empty(cin, cout))
        self.bias = nn.Parameter(torch.empty(cout))
        bound = 1.0 / (num_bases * cin) ** 0.5                     # models/rgcn.py:317-322
        for p in (self.basis, self.att, self.root, self.bias):
            nn.init.uniform_(p, -bound, bound)

    def forward(self, x, edge_index, edge_type, edge_norm):
        w = (self.att @ self.basis.view(self.nb, -1)).view(self.R, self.cin, self.cout)
        w_e = torch.index_select(w, 0, edge_type)                  # [E, cin, cout] materialised, as the reference does
        msg = torch.bmm(x[edge_index[0]].unsqueeze(1), w_e).squeeze(-2) * edge_norm.view(-1, 1)
        return scatter_sum(msg, edge_index[1], x.size(0)) + x @ self.root + self.bias


class GCN(nn.Module):
    def __init__(self, g_dim, h1_dim, h2_dim, n_speakers):
        super().__init__()
        self.conv1 = RGCNConvBasis(g_dim, h1_dim, 2 * n_speakers ** 2, 30)
        self.conv2 = GraphConvAdd(h1_dim, h2_dim)

    def forward(self, x, edge_index, edge_norm, edge_type):
        return self.conv2(self.conv1(x, edge_index, edge_type, edge_norm), edge_index)


class _EmotionAtt(nn.Module):
    def __init__(self, d):
        super().__init__()
        self.lin = nn.Linear(d, d)                                # constructed, never used (dgcn_models.py:158,163-170)


class Classifier(nn.Module):
    def __init__(self, input_dim, hidden, n_classes, dropout):
        super().__init__()
        self.emotion_att = _EmotionAtt(input_dim)
        self.lin1 = nn.Linear(input_dim, hidden)
        self.drop = nn.Dropout(dropout)
        self.lin2 = nn.Linear(hidden, n_classes)

    def forward(self, h):
        return self.lin2(self.drop(F.relu(self.lin1(h))))


class DGCNOracle(nn.Module):
    def __init__(self, n_speakers, input_size=100, hidden_size=200, context=(10, 10), dropout=0.4, n_classes=4):
        super().__init__()
        self.wp, self.wf = context
        self.n_speakers = n_speakers
        self.rnn = SeqContext(input_size, hidden_size, dropout)
        self.edge_att = EdgeAtt(hidden_size, self.wp, self.wf)
        self.gcn = GCN(hidden_size, 100, 100, n_speakers)
        self.clf = Classifier(hidden_size + 100, 100, n_classes, dropout)

    def forward(self, input_tensor, speaker_tensor, text_length, **kwargs):
        node_features = self.rnn(text_length, input_tensor)
        x, ei, norm, typ = dgcn_graphify(node_features, text_length, speaker_tensor, self.wp, self.wf,
                                         self.n_speakers, self.edge_att)
        self.last_graph = (ei, norm, typ)
        graph_out = self.gcn(x, ei, norm, typ)
        return self.clf(torch.cat([x, graph_out], dim=-1)), graph_out


IEMOCAP6_WEIGHTS = [1 / 0.086747, 1 / 0.144406, 1 / 0.227883, 1 / 0.160585, 1 / 0.127711, 1 / 0.252668]  # dgcn.py:109-110


def dgcn_train_step(model, optim, batch, loss_weights=None):
    """dgcn.py:117-134."""
    logits, _ = model(**batch)
    loss = F.cross_entropy(logits, batch["label"], weight=loss_weights)
    optim.zero_grad()
    loss.backward()
    optim.step()
    return loss.detach(), torch.eq(logits.argmax(-1), batch["label"]).float().mean()
