"""Oracle: COGMEN forward / loss on CPU, fp32 (track_mm/cogmen.py:61-160).

Structure-faithful: per-edge python graph construction, the
computed-and-discarded 2-layer Transformer encoder (cogmen.py:145-147 applies
every module of ``self.rnn`` to the RAW input, so only ``rnn.1`` -- the
Linear(D,100) -- reaches the logits), per-relation RGCN loop.
``state_dict`` keys mirror the reference module.
"""
import torch
from torch import nn
from torch.nn import functional as F

from .graph import window_graph_loop
from .pyg import RGCNConvMean, TransformerConv1


def pick_heads(input_size, num_head=17):
    """First h in [6, num_head) dividing input_size (cogmen.py:86-92)."""
    for h in range(6, num_head):
        if input_size % h == 0:
            return h
    raise AssertionError(input_size)


class GNN(nn.Module):
    """cogmen.py:61-74.  num_relations is 2*2**2 = 8 whatever the dataset
    (COGMENModule never forwards its n_speakers, cogmen.py:114)."""

    def __init__(self, g_dim, h1_dim, h2_dim, n_speakers=2):
        super().__init__()
        self.conv1 = RGCNConvMean(g_dim, h1_dim, 2 * n_speakers ** 2)
        self.conv2 = TransformerConv1(h1_dim, h2_dim)
        self.bn = nn.BatchNorm1d(h2_dim)
        self.relu = nn.LeakyReLU()

    def forward(self, x, edge_index, edge_type):
        x = self.conv1(x, edge_index, edge_type)
        return self.relu(self.bn(self.conv2(x, edge_index)))


class COGMENOracle(nn.Module):
    def __init__(self, input_size, hidden_size=100, num_head=17, n_speakers=2,
                 n_classes=6, dead_encoder=True, chained=False, bf16_products=False):
        super().__init__()
        self.n_speakers = n_speakers
        self.dead_encoder = dead_encoder
        # chained: rnn.1(rnn.0(x, src_key_padding_mask)) -- the variant cogmen.py:94-109 builds the encoder for; NOT
        # what the reference computes ("parity unpinned": checked against torch autograd only)
        self.chained = chained
        self.enc_rnd = None      # chained mode: rounding hook of oracle/encoder.py (None: the torch module itself)
        layer = nn.TransformerEncoderLayer(d_model=input_size, nhead=pick_heads(input_size, num_head),
                                           dropout=0.5, batch_first=True)
        encoder = nn.TransformerEncoder(layer, num_layers=2, enable_nested_tensor=False)
        self.rnn = nn.ModuleList([encoder, nn.Linear(input_size, hidden_size)])
        self.gcn = GNN(hidden_size, hidden_size, hidden_size)
        self.cls = nn.Sequential(nn.Linear(100, 100), nn.ReLU(), nn.Dropout(p=0.5),
                                 nn.Linear(100, n_classes))
        # bf16_products: the graph part's dense products with the operand rounding of the bf16 compute mode
        # (oracle/pyg.py RoundedLinear / RGCNMeanRounded) -- the same algorithm, rounded where that mode rounds
        self.gcn.conv1.rounded = self.gcn.conv2.rounded = bool(bf16_products)
        self.bf16_products = bool(bf16_products)

    def _linear(self, mod, t):
        """rnn.1 / cls.0 / cls.3: in the bf16 mode only their WEIGHT gradient is a bf16 product (pyg.RoundedWgradLinear)"""
        if self.bf16_products:
            from .pyg import RoundedWgradLinear
            return RoundedWgradLinear.apply(t, mod.weight, mod.bias)
        return mod(t)

    def _cls(self, t):
        if not self.bf16_products:
            return self.cls(t)
        return self._linear(self.cls[3], self.cls[2](self.cls[1](self._linear(self.cls[0], t))))

    def forward(self, input_tensor, speaker_tensor, text_length, *args, **kwargs):
        if self.chained:
            T = input_tensor.shape[1]
            pad = torch.arange(T)[None, :] >= text_length[:, None]
            if self.enc_rnd is None:
                h0 = self.rnn[1](self.rnn[0](input_tensor, src_key_padding_mask=pad))
            else:
                from .encoder import encoder
                rnd = self.enc_rnd
                h0 = F.linear(rnd(encoder(input_tensor, self.rnn[0], pad, rnd=rnd)), rnd(self.rnn[1].weight), self.rnn[1].bias)
            x, edge_index, edge_type, _ = window_graph_loop(h0, text_length, speaker_tensor, 5, 5, self.n_speakers)
            self.last_graph = (edge_index, edge_type)
            return self.cls(self.gcn(x, edge_index, edge_type)), x
        node_features = input_tensor
        for mod in self.rnn:  # each module sees the raw input (cogmen.py:146-147)
            if mod is self.rnn[0] and not self.dead_encoder:
                continue
            node_features = self._linear(mod, input_tensor) if mod is self.rnn[1] else mod(input_tensor)
        x, edge_index, edge_type, _ = window_graph_loop(
            node_features, text_length, speaker_tensor, 5, 5, self.n_speakers)
        self.last_graph = (edge_index, edge_type)
        out = self.gcn(x, edge_index, edge_type)
        return self._cls(out), x


def cogmen_train_step(model, optim, batch):
    """track_mm/cogmen.py:179-195 without the lumo plumbing."""
    ys = batch["label"]
    logits, _ = model(**batch)
    loss = F.cross_entropy(logits, ys)
    optim.zero_grad()
    loss.backward()
    optim.step()
    acc = torch.eq(logits.argmax(dim=-1), ys).float().mean()
    return loss.detach(), acc
