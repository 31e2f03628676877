"""Oracle: DAG-ERC forward / loss on CPU, fp32 (track_mm/dagerc.py:73-237,
track_mm/dagerc_models.py:83-90,312-365,425-442).

Structure-faithful restatement (this is also the timed CPU baseline):
python adjacency / speaker-mask loops over ``.tolist()``-ed speakers, per-step
``torch.cat`` regrowth of the layer state, full-prefix recompute of Wr0/Wr1 at
every step.  ``state_dict`` keys mirror the reference,
including the parameters that never receive a gradient (``fcs.*``,
``attentive_node_features.transform.*``).
PINNED: tests/golden/dagerc_*.npz hold logits / loss / gradients produced by the
reference's own DAGERCModule on the same inputs and parameters.
"""
import torch
from torch import nn
from torch.nn import functional as F

from .graph import dag_adjacency_loop, speaker_mask_loop


class GatherV1(nn.Module):
    """GAT_dialoggcn_v1 (dagerc_models.py:312-365)."""

    def __init__(self, hidden):
        super().__init__()
        self.linear = nn.Linear(hidden * 2, 1)
        self.Wr0 = nn.Linear(hidden, hidden, bias=False)
        self.Wr1 = nn.Linear(hidden, hidden, bias=False)

    def forward(self, Q, K, V, adj, s_mask):
        n = K.size(1)
        X = torch.cat((Q.unsqueeze(1).expand(-1, n, -1), K), dim=2)
        alpha = self.linear(X).permute(0, 2, 1)                       # (B,1,n)
        alpha = alpha - (1 - adj.unsqueeze(1)) * 1e30                  # mask_logic, dagerc_models.py:83-90
        w = F.softmax(alpha, dim=2)
        m = s_mask.unsqueeze(2).float()
        Vr = self.Wr0(V) * m + self.Wr1(V) * (1 - m)                   # recomputed over the whole prefix each step
        return w, torch.bmm(w, Vr).squeeze(1)


class _Identity(nn.Module):
    """attentive_node_features with nodal_att_type None: identity, but owns an unused Linear (dagerc_models.py:425-442)."""

    def __init__(self, hidden):
        super().__init__()
        self.transform = nn.Linear(hidden, hidden)

    def forward(self, features, lengths, nodal_att_type):
        return features


class DAGERCOracle(nn.Module):
    def __init__(self, emb_dim=100, dropout=0.2, n_classes=7, gnn_layers=4):
        super().__init__()
        hidden = 300
        self.gnn_layers = gnn_layers
        self.dropout = nn.Dropout(dropout)
        self.gather = nn.ModuleList([GatherV1(hidden) for _ in range(gnn_layers)])
        self.grus_c = nn.ModuleList([nn.GRUCell(hidden, hidden) for _ in range(gnn_layers)])
        self.grus_p = nn.ModuleList([nn.GRUCell(hidden, hidden) for _ in range(gnn_layers)])
        self.fcs = nn.ModuleList([nn.Linear(hidden * 2, hidden) for _ in range(gnn_layers)])   # unused
        self.fc1 = nn.Linear(emb_dim, hidden)
        in_dim = hidden * (gnn_layers + 1) + emb_dim
        self.out_mlp = nn.Sequential(nn.Linear(in_dim, hidden), nn.ReLU(), nn.Linear(hidden, hidden), nn.ReLU(),
                                     nn.Dropout(dropout), nn.Linear(hidden, n_classes))
        self.attentive_node_features = _Identity(in_dim)

    def forward(self, input_tensor, text_length, speaker_tensor, **kwargs):
        T = input_tensor.size(1)
        speakers = speaker_tensor.tolist()                             # device->host sync in the reference
        mx = int(torch.max(text_length).item())
        adj = dag_adjacency_loop(speakers, mx).to(input_tensor.device)   # dagerc.py:109-129
        s_mask = speaker_mask_loop(speakers, mx).to(input_tensor.device)  # dagerc.py:131-154
        self.last_adj, self.last_s_mask = adj, s_mask
        H = [F.relu(self.fc1(input_tensor))]
        for l in range(self.gnn_layers):
            C = self.grus_c[l](H[l][:, 0, :]).unsqueeze(1)
            M = torch.zeros_like(C).squeeze(1)
            P = self.grus_p[l](M, H[l][:, 0, :]).unsqueeze(1)
            H1 = C + P
            for i in range(1, T):
                _, M = self.gather[l](H[l][:, i, :], H1, H1, adj[:, i, :i], s_mask[:, i, :i])
                C = self.grus_c[l](H[l][:, i, :], M).unsqueeze(1)
                P = self.grus_p[l](M, H[l][:, i, :]).unsqueeze(1)
                H1 = torch.cat((H1, C + P), dim=1)                     # O(T^2) regrowth, dagerc.py:186
            H.append(H1)
        H.append(input_tensor)
        H = torch.cat(H, dim=2)
        H = self.attentive_node_features(H, text_length, None)
        return self.out_mlp(H), None


def dagerc_loss(model, batch):
    """dagerc.py:221-226."""
    logits, _ = model(**batch)
    sel = logits[batch["attention_mask"].bool()]
    return F.cross_entropy(sel, batch["label"]), sel


def dagerc_train_step(model, optim, batch):
    """dagerc.py:217-237."""
    loss, sel = dagerc_loss(model, batch)
    optim.zero_grad()
    loss.backward()
    nn.utils.clip_grad_norm_(model.parameters(), 5)
    optim.step()
    return loss.detach(), torch.eq(sel.argmax(-1), batch["label"]).float().mean()
