"""DialogueGCN on the MI355X hot path (drop-in for track_mm/dgcn.py:53-134).

``DGCNModule`` keeps the reference's constructor signature, ``state_dict`` keys (including
the never-used ``clf.emotion_att.lin``) and ``forward(**batch) -> (logits [N,C], graph_out [N,100])``.

Chain: packed BiLSTM (rnn.py) -> gather of the valid rows -> EdgeAtt (GEMM + per-source softmax over the
window) -> K1 window graph (w = 10) -> basis-space aggregation + one GEMM (vendored RGCNConv, models/rgcn.py)
-> GraphConv (neighbour sum + two GEMMs) -> classifier on [features | graph_out] written side by side in one
[N,300] buffer (no torch.cat) -> class-weighted CE; hand-written backward of every stage.
"""
import torch
from torch import nn

from . import capi
from .engine import WorkspaceCache, FlatParams, FusedAdam, GemmPlanner, SideStream, all_reduce_grads, linear_fwd, linear_wgrad, \
    matmul_wgrad_io
from .rnn import BiLSTM2, lstm_groups

G_DIM, H1, NB = 200, 100, 30


class _SeqContext(nn.Module):
    def __init__(self, u_dim, g_dim, dropout):
        super().__init__()
        self.rnn = nn.LSTM(u_dim, g_dim // 2, dropout=dropout, bidirectional=True, num_layers=2, batch_first=True)


class _EdgeAtt(nn.Module):
    def __init__(self, g_dim):
        super().__init__()
        self.weight = nn.Parameter(torch.zeros(g_dim, g_dim))
        self.weight.data.normal_(0, 2.0 / (g_dim + g_dim))          # dgcn_models.py:128-130 (var used as std)


class _RGCNBasis(nn.Module):
    def __init__(self, cin, cout, R, nb):
        super().__init__()
        self.basis = nn.Parameter(torch.empty(nb, cin, cout))
        self.att = nn.Parameter(torch.empty(R, nb))
        self.root = nn.Parameter(torch.empty(cin, cout))
        self.bias = nn.Parameter(torch.empty(cout))
        bound = 1.0 / (nb * cin) ** 0.5                              # models/rgcn.py:317-322
        for p in (self.basis, self.att, self.root, self.bias):
            nn.init.uniform_(p, -bound, bound)


class _GraphConv(nn.Module):
    def __init__(self, cin, cout):
        super().__init__()
        self.lin_rel = nn.Linear(cin, cout, bias=True)
        self.lin_root = nn.Linear(cin, cout, bias=False)


class _GCN(nn.Module):
    def __init__(self, g_dim, h1, h2, n_speakers):
        super().__init__()
        self.conv1 = _RGCNBasis(g_dim, h1, 2 * n_speakers ** 2, NB)
        self.conv2 = _GraphConv(h1, h2)


class _EmotionAtt(nn.Module):
    def __init__(self, d):
        super().__init__()
        self.lin = nn.Linear(d, d)


class _Classifier(nn.Module):
    def __init__(self, input_dim, hidden, n_classes, dropout):
        super().__init__()
        self.emotion_att = _EmotionAtt(input_dim)                    # constructed, never used
        self.lin1 = nn.Linear(input_dim, hidden)
        self.drop = nn.Dropout(dropout)
        self.lin2 = nn.Linear(hidden, n_classes)


class DGCNModule(nn.Module):
    def __init__(self, n_speakers, input_size=100, hidden_size=200, context=(10, 10), dropout=0.4, n_classes=4,
                 compute="f32", seed=1):
        super().__init__()
        assert hidden_size == G_DIM
        self.wp, self.wf = context
        self.n_speakers, self.input_size, self.n_classes, self.compute = n_speakers, input_size, n_classes, compute
        self.R = 2 * n_speakers ** 2
        # few relations (two speakers: 8 < 30 bases): RGCNConv in relation space, W_r composed first as models/rgcn.py:300-304
        # does (csrc/dgcn_ops.hip); None = decide in finalize() from the library's limit, True / False = forced (tests)
        self.relation_space = None
        # the sequence encoder on compact rows (the sum(lengths) valid positions instead of B * T padded ones; rnn.py)
        self.compact_lstm = True
        # basis space: aggregate + Z @ basis + x @ root as one tile launch (erc_brgcn_fwd_tile), and the node side of the
        # backward likewise (erc_brgcn_bwd_source_tile); False = the separate kernels + GEMMs (tests compare the two)
        self.fused_rgcn_fwd = True
        # training step: the RGCN slab sum, GraphConv, the classifier, the loss and their backward down to dXc / dAGG / dHc as one
        # launch (erc_dgcn_tail) instead of nine; False = the separate kernels (tests compare the two)
        self.fused_tail = True
        # ... and the RGCN backward's slab sum into dXc + its relation sums (d att) inside EdgeAtt's backward launch
        self.fused_edge_bwd = True
        self.drop_p = float(dropout)
        self.rnn = _SeqContext(input_size, hidden_size, dropout)
        self.edge_att = _EdgeAtt(hidden_size)
        self.gcn = _GCN(hidden_size, H1, H1, n_speakers)
        self.clf = _Classifier(hidden_size + H1, 100, n_classes, dropout)
        self.flat, self._ws, self._seed = None, WorkspaceCache(), seed

    def live_groups(self):
        g, c = self.gcn, self.clf
        return lstm_groups("rnn.rnn.", self.rnn.rnn) + [
            [("edge_att.weight", self.edge_att.weight)],
            [("gcn.conv1.basis", g.conv1.basis)], [("gcn.conv1.att", g.conv1.att)],
            [("gcn.conv1.root", g.conv1.root)], [("gcn.conv1.bias", g.conv1.bias)],
            [("gcn.conv2.lin_rel.weight", g.conv2.lin_rel.weight)], [("gcn.conv2.lin_rel.bias", g.conv2.lin_rel.bias)],
            [("gcn.conv2.lin_root.weight", g.conv2.lin_root.weight)],
            [("clf.lin1.weight", c.lin1.weight)], [("clf.lin1.bias", c.lin1.bias)],
            [("clf.lin2.weight", c.lin2.weight)], [("clf.lin2.bias", c.lin2.bias)],
        ]

    def finalize(self, device):
        self.to(device)
        self.flat = FlatParams(self.live_groups(), device)
        self.lstm = BiLSTM2(self.flat, "rnn.rnn.", self.input_size, drop_p=self.drop_p)
        self.rng_state = torch.tensor([0, self._seed], dtype=torch.int64, device=device)
        self.side = SideStream()
        if self.relation_space is None:
            self.relation_space = self.R <= capi.rrgcn_max_relations()
        return self

    @property
    def _last_ws(self):
        """workspace of the most recent forward (tests / bench read results out of it)"""
        return self._ws.last

    def _workspace(self, B, T, N, device):
        return self._ws.get((B, T, N), lambda: self._make_workspace(B, T, N, device))

    def _make_workspace(self, B, T, N, device):
        f32 = lambda *s: torch.zeros(*s, dtype=torch.float32, device=device)
        i32 = lambda *s: torch.zeros(*s, dtype=torch.int32, device=device)
        w = (self.wp if self.wp >= 0 else T) + (self.wf if self.wf >= 0 else T) + 1
        E = max(1, N * min(w, T))
        C, BT = self.n_classes, B * T
        g = dict(node_off=i32(B + 1), node_row=i32(N), node_spk=i32(N), in_ptr=i32(N + 1), in_src=i32(E),
                 in_typ=i32(E), out_ptr=i32(N + 1), out_dst=i32(E), out_typ=i32(E), out_eid=i32(E), counts=i32(2))
        ws = dict(g=g, E=E, rnn_out=f32(BT, G_DIM), Xc=f32(N, G_DIM + H1), ATT=f32(N, G_DIM), norm=f32(E),
                  Z=f32(N, self._kb * G_DIM), Hc=f32(N, H1), AGG=f32(N, H1), Zc=f32(N, 100), logits=f32(N, C), stats=torch.zeros(max(256, capi.head_ce_stats_floats(N), capi.dgcn_tail_stats_floats(N)), dtype=torch.float32, device=device),
                  dlogits=f32(N, C), dZc=f32(N, 100), dXc=f32(N, G_DIM + H1), dAGG=f32(N, H1), dHc=f32(N, H1),
                  dZ=f32(N, self._kb * G_DIM), dnorm=f32(E), TT=f32(E, NB), U=f32(N, self._kb * H1),
                  basisT=f32(self._kb * H1, G_DIM), Wr=f32(self._kb * G_DIM, H1), dWr=f32(self._kb * G_DIM, H1),
                  DATT=f32(N, G_DIM), dscore=f32(E), drnn=f32(BT, G_DIM),
                  rgcn_slabs=f32(capi.brgcn_fwd_tile_slab_floats(N)), rgcn_dslabs=f32(capi.brgcn_fwd_tile_slabs() * N * G_DIM),
                  dn_slabs=f32(capi.brgcn_fwd_tile_slabs() * E))
        D = self.input_size
        slab = 12 * N * H1 + 4 * BT * 800 + 10 * (800 * D + 800 * 200 + 2 * 400 * 100 * 2) + 4 * NB * G_DIM * H1 + \
            8 * (G_DIM * G_DIM + 300 * 100) + (1 << 21)
        ws["planner"] = GemmPlanner(device, slab, grad=self.flat.grad)
        ws["jobs"] = None
        return ws

    @property
    def _kb(self):
        """blocks of the RGCN aggregate Z: relations (relation space) or bases"""
        return self.R if self.relation_space else NB

    def _shape(self, x, lens, label, n_nodes=None):
        B, T = x.shape[0], x.shape[1]
        N = int(label.shape[0]) if label is not None else (int(n_nodes) if n_nodes is not None else int(lens.sum().item()))
        return B, T, N

    def _tail_ok(self, N):
        max_rows, max_win = capi.dgcn_tail_limits()
        return self.fused_tail and self.n_classes <= 8 and N <= max_rows and 0 <= self.wp <= max_win and 0 <= self.wf <= max_win

    def _forward_impl(self, x, spk, lens, B, T, N, training, with_logits=True, tail=False):
        fp = self.flat
        ws = self._workspace(B, T, N, x.device)
        g, pl = ws["g"], ws["planner"]
        pl.reset()
        D, C, BT, XW = self.input_size, self.n_classes, B * T, G_DIM + H1
        x_bf16 = x.dtype == torch.bfloat16
        capi.window_graph_build(lens, spk, spk.stride(0), spk.stride(1), B, T, self.wp, self.wf, self.n_speakers, N,
                                ws["E"], g)
        Xc = ws["Xc"]
        if self.compact_lstm:
            # the BiLSTM on the N valid positions in node order (row = node_off[b] + t), written next to the graph output
            self.lstm.forward(pl, x, D, N, B, T, T, 1, lens, training, self.rng_state, Xc, XW, x_bf16=x_bf16,
                              node_off=g["node_off"], node_row=g["node_row"], store=ws)
        else:
            self.lstm.forward(pl, x, D, BT, B, T, T, 1, lens, training, self.rng_state, ws["rnn_out"], G_DIM, x_bf16=x_bf16,
                              store=ws)
            capi.gather_rows(ws["rnn_out"], G_DIM, g["node_row"], N, G_DIM, Xc, XW)
        # EdgeAtt: att = x W^T, softmax over each source's window
        capi.gemm_f32(Xc, XW, 0, None, fp.w("edge_att.weight"), G_DIM, 0, None, ws["ATT"], G_DIM, N, G_DIM, G_DIM)
        capi.edge_att_fwd(Xc, XW, ws["ATT"], G_DIM, G_DIM, N, g, ws["norm"])
        # RGCNConv(basis): Z (basis space) -> Z @ basis + x @ root + bias
        if self.relation_space:
            capi.basis_compose(fp.w("gcn.conv1.att"), fp.w("gcn.conv1.basis"), self.R, NB, G_DIM, H1, ws["Wr"], ws["basisT"])
            capi.rrgcn_agg_fwd(Xc, XW, G_DIM, N, self.R, g, ws["norm"], ws["Z"])
            Wz = ws["Wr"]
        else:
            Wz = fp.w("gcn.conv1.basis")
        if self.fused_rgcn_fwd and not self.relation_space:
            # aggregate + basis product + root product in one tile launch (csrc/dgcn_ops.hip), partial slabs
            capi.brgcn_fwd_tile(Xc, XW, G_DIM, H1, N, g, ws["norm"], fp.w("gcn.conv1.att"), NB, Wz, fp.w("gcn.conv1.root"),
                                ws["Z"], ws["rgcn_slabs"])
            ws["tail_src"] = (ws["rgcn_slabs"], capi.brgcn_fwd_tile_slabs(), N * H1)
            if tail:
                return ws      # the slab sum and everything behind it: erc_dgcn_tail (loss_and_grads)
            capi.slab_reduce(ws["rgcn_slabs"], capi.brgcn_fwd_tile_slabs(), N * H1, fp.w("gcn.conv1.bias"), H1, 0, ws["Hc"],
                             N * H1)
        else:
            if not self.relation_space:
                capi.brgcn_agg_fwd(Xc, XW, G_DIM, N, g, ws["norm"], fp.w("gcn.conv1.att"), NB, ws["Z"])
            K1 = self._kb * G_DIM
            S1 = pl.split_for(N, H1, K1)
            src = pl.take((S1 + 1) * N * H1)
            capi.gemm_f32(ws["Z"], K1, 0, None, Wz, H1, 1, None, pl.ws[src:], H1, N, H1, K1,
                          split_k=S1, c_slab=N * H1)
            capi.gemm_f32(Xc, XW, 0, None, fp.w("gcn.conv1.root"), H1, 1, None, pl.ws[src + S1 * N * H1:], H1, N, H1, G_DIM)
            ws["tail_src"] = (pl.ws[src:], S1 + 1, N * H1)
            if tail:
                return ws
            capi.slab_reduce(pl.ws[src:], S1 + 1, N * H1, fp.w("gcn.conv1.bias"), H1, 0, ws["Hc"], N * H1)
        # GraphConv: W_rel * sum_{j->i} h_j + b + W_root h_i, written next to the features
        capi.csr_sum(ws["Hc"], H1, H1, N, g["in_ptr"], g["in_src"], ws["AGG"], H1)
        gout = Xc[:, G_DIM:]
        capi.gemm_f32(ws["AGG"], H1, 0, None, fp.w("gcn.conv2.lin_rel.weight"), H1, 0, None, gout, XW, N, H1, H1,
                      bias=fp.w("gcn.conv2.lin_rel.bias"))
        capi.gemm_f32(ws["Hc"], H1, 0, None, fp.w("gcn.conv2.lin_root.weight"), H1, 0, None, gout, XW, N, H1, H1,
                      accumulate=1)
        # classifier on [features | graph_out]
        p = self.drop_p if training else 0.0
        linear_fwd(pl, Xc, XW, None, fp.w("clf.lin1.weight"), fp.w("clf.lin1.bias"), ws["Zc"], 100, N, 100, XW,
                   act=3 if p > 0 else 1, drop_p=p, rng=self.rng_state)
        if with_logits:      # the training step computes them together with the loss and its gradient (erc_head_ce)
            linear_fwd(pl, ws["Zc"], 100, None, fp.w("clf.lin2.weight"), fp.w("clf.lin2.bias"), ws["logits"], C, N, C, 100)
        return ws

    def forward(self, input_tensor, speaker_tensor, text_length, label=None, **kwargs):
        if self.flat is None:
            raise capi.ErcGraftError("call DGCNModule.finalize(device) before forward")
        B, T, N = self._shape(input_tensor, text_length, label, kwargs.get("n_nodes"))
        ws = self._forward_impl(input_tensor, speaker_tensor, text_length, B, T, N, self.training)
        return ws["logits"], ws["Xc"][:, G_DIM:]

    def loss_and_grads(self, batch, class_weight=None):
        x, spk, lens, ys = batch["input_tensor"], batch["speaker_tensor"], batch["text_length"], batch["label"]
        B, T, N = self._shape(x, lens, ys)
        training = self.training
        fused_tail = self.n_classes <= 8
        tail = self._tail_ok(N)
        ws = self._forward_impl(x, spk, lens, B, T, N, training, with_logits=not fused_tail, tail=tail)
        fp, g, pl, off = self.flat, ws["g"], ws["planner"], self.flat.offsets
        C, BT, XW = self.n_classes, B * T, G_DIM + H1
        Xc, dXc = ws["Xc"], ws["dXc"]
        p = self.drop_p if training else 0.0
        # classifier
        if tail:
            slabs, n_slabs, stride = ws["tail_src"]
            capi.dgcn_tail(slabs, n_slabs, stride, fp.w("gcn.conv1.bias"), g, max(self.wp, self.wf),
                           fp.w("gcn.conv2.lin_rel.weight"), fp.w("gcn.conv2.lin_rel.bias"), fp.w("gcn.conv2.lin_root.weight"),
                           fp.w("clf.lin1.weight"), fp.w("clf.lin1.bias"), fp.w("clf.lin2.weight"), fp.w("clf.lin2.bias"), ys,
                           class_weight, C, N, p, self.rng_state, Xc, XW, ws["Hc"], ws["AGG"], ws["Zc"], ws["logits"], ws["dlogits"],
                           ws["dZc"], dXc, XW, ws["dAGG"], ws["dHc"], ws["stats"])
        elif fused_tail:
            # lin2 + cross entropy + their backward through the ReLU / dropout mask in one launch (dgcn_models.py:163-170)
            capi.head_ce(ws["Zc"], 100, 100, C, N, fp.w("clf.lin2.weight"), fp.w("clf.lin2.bias"), ys, class_weight,
                         1.0 / (1.0 - p), ws["logits"], C, ws["dlogits"], C, ws["dZc"], 100, ws["stats"])
        else:
            capi.cross_entropy(ws["logits"], C, C, N, None, ys, class_weight, 1.0, ws["dlogits"], C, ws["stats"])
            capi.gemm_f32(ws["dlogits"], C, 0, None, fp.w("clf.lin2.weight"), 100, 1, None, ws["dZc"], 100, N, 100, C,
                          act=2, aux=ws["Zc"], ldaux=100, act_scale=1.0 / (1.0 - p))
        linear_wgrad(pl, ws["dlogits"], C, ws["Zc"], 100, None, C, 100, N, off["clf.lin2.weight"], off["clf.lin2.bias"], defer=True)
        if not tail:
            capi.gemm_f32(ws["dZc"], 100, 0, None, fp.w("clf.lin1.weight"), XW, 1, None, dXc, XW, N, XW, 100)
        linear_wgrad(pl, ws["dZc"], 100, Xc, XW, None, 100, XW, N, off["clf.lin1.weight"], off["clf.lin1.bias"], defer=True)
        # GraphConv
        dG = dXc[:, G_DIM:]
        if not tail:
            capi.gemm_f32(dG, XW, 0, None, fp.w("gcn.conv2.lin_rel.weight"), H1, 1, None, ws["dAGG"], H1, N, H1, H1)
        linear_wgrad(pl, dG, XW, ws["AGG"], H1, None, H1, H1, N, off["gcn.conv2.lin_rel.weight"],
                     off["gcn.conv2.lin_rel.bias"], defer=True)
        linear_wgrad(pl, dG, XW, ws["Hc"], H1, None, H1, H1, N, off["gcn.conv2.lin_root.weight"], None, defer=True)
        if not tail:
            capi.gemm_f32(dG, XW, 0, None, fp.w("gcn.conv2.lin_root.weight"), H1, 1, None, ws["dHc"], H1, N, H1, H1)
        capi.csr_sum(ws["dAGG"], H1, H1, N, g["out_ptr"], g["out_dst"], ws["dHc"], H1, accumulate=1)
        # RGCNConv(basis)
        KB = self._kb
        K1 = KB * G_DIM
        dn_src, dn_parts, dn_stride = ws["dnorm"], 1, 0
        if self.relation_space:
            capi.gemm_f32(ws["dHc"], H1, 0, None, ws["Wr"], H1, 0, None, ws["dZ"], K1, N, K1, H1)
            capi.rrgcn_bwd_edges(Xc, XW, G_DIM, N, self.R, g, ws["dZ"], ws["dnorm"])
            # dW_r = Z^T dHc (+ the bias strip); basis / comp gradients follow from it after the batched launch
            pl.defer(ws["Z"], K1, ws["dHc"], H1, ws["dWr"], H1, K1, H1, N, 2, pl.grad[off["gcn.conv1.bias"]:])
            capi.rrgcn_bwd_source(ws["dHc"], H1, H1, N, self.R, g, ws["norm"], ws["U"])
        else:
            if self.fused_rgcn_fwd:
                # dZ stays in LDS: blocks on the matrix cores, 6 dots per in-edge; d norm as partial vectors per basis group
                E_cap = ws["E"]
                capi.brgcn_bwd_edges_tile(Xc, XW, G_DIM, H1, N, self.R, g, ws["norm"], fp.w("gcn.conv1.att"), NB,
                                          fp.w("gcn.conv1.basis"), ws["dHc"], H1, ws["TT"], ws["dn_slabs"], E_cap,
                                          None if self.fused_edge_bwd else fp.g("gcn.conv1.att"))
                dn_src, dn_parts, dn_stride = ws["dn_slabs"], capi.brgcn_fwd_tile_slabs(), E_cap      # summed by edge_att_bwd
            else:
                capi.gemm_f32(ws["dHc"], H1, 0, None, fp.w("gcn.conv1.basis"), H1, 0, None, ws["dZ"], K1, N, K1, H1)
                capi.brgcn_bwd_edges(Xc, XW, G_DIM, N, self.R, g, ws["norm"], fp.w("gcn.conv1.att"), NB, ws["dZ"], ws["dnorm"],
                                     ws["TT"], fp.g("gcn.conv1.att"))
            matmul_wgrad_io(pl, ws["Z"], K1, ws["dHc"], H1, K1, H1, N, off["gcn.conv1.basis"], off["gcn.conv1.bias"], defer=True)
            if not self.fused_rgcn_fwd:
                capi.brgcn_bwd_source(ws["dHc"], H1, H1, N, g, ws["norm"], fp.w("gcn.conv1.att"), NB, ws["U"])
                capi.transpose_batched(fp.w("gcn.conv1.basis"), NB, G_DIM, H1, ws["basisT"])
        matmul_wgrad_io(pl, Xc, XW, ws["dHc"], H1, G_DIM, H1, N, off["gcn.conv1.root"], None, defer=True)
        if self.fused_rgcn_fwd and not self.relation_space:
            # dXc += sum_b U_b basis_b^T + dHc root^T: one tile launch + the slab sum added into dXc
            capi.brgcn_bwd_source_tile(ws["dHc"], H1, G_DIM, H1, N, g, ws["norm"], fp.w("gcn.conv1.att"), NB,
                                       fp.w("gcn.conv1.basis"), fp.w("gcn.conv1.root"), ws["rgcn_dslabs"])
            if not self.fused_edge_bwd:
                capi.slab_reduce(ws["rgcn_dslabs"], capi.brgcn_fwd_tile_slabs(), N * G_DIM, None, G_DIM, 4, dXc, N * G_DIM, ld_out=XW)
        else:
            capi.gemm_f32(ws["U"], KB * H1, 0, None, ws["basisT"], G_DIM, 1, None, dXc, XW, N, G_DIM, KB * H1, accumulate=1)
            capi.gemm_f32(ws["dHc"], H1, 0, None, fp.w("gcn.conv1.root"), H1, 0, None, dXc, XW, N, G_DIM, H1, accumulate=1)
        # EdgeAtt
        if self.fused_edge_bwd and self.fused_rgcn_fwd and not self.relation_space:
            # + the slab sum of the RGCN backward's feature gradients and its relation sums (d att), inside the source-side launch
            capi.edge_att_bwd_fused(Xc, XW, ws["ATT"], G_DIM, G_DIM, N, g, ws["norm"], dn_src, dXc, XW, 1, ws["DATT"], G_DIM,
                                    ws["dscore"], dn_parts=dn_parts, dn_stride=dn_stride, dx_slabs=ws["rgcn_dslabs"],
                                    n_dx_slabs=capi.brgcn_fwd_tile_slabs(), dx_slab_stride=N * G_DIM, rs_TT=ws["TT"],
                                    rs_datt=fp.g("gcn.conv1.att"), rs_R=self.R)
        else:
            capi.edge_att_bwd(Xc, XW, ws["ATT"], G_DIM, G_DIM, N, g, ws["norm"], dn_src, dXc, XW, 1, ws["DATT"], G_DIM,
                              ws["dscore"], dn_parts=dn_parts, dn_stride=dn_stride)
        linear_wgrad(pl, ws["DATT"], G_DIM, Xc, XW, None, G_DIM, G_DIM, N, off["edge_att.weight"], None, defer=True)
        capi.gemm_f32(ws["DATT"], G_DIM, 0, None, fp.w("edge_att.weight"), G_DIM, 1, None, dXc, XW, N, G_DIM, G_DIM,
                      accumulate=1)
        # the weight gradients of everything behind the BiLSTM (classifier, graph layers, EdgeAtt) do not wait for its backward
        # scans, which occupy 2 B of the 256 CUs: their launch goes to a second stream (ERC_SIDE_STREAM=1)
        if self.side.enabled:
            with self.side.fork():
                pl.flush_wgrads(ws, tag="_early")
        # back to the padded rows and through the BiLSTM
        if self.compact_lstm:
            self.lstm.backward(pl, dXc, XW)
        else:
            ws["drnn"].zero_()
            capi.gather_rows(dXc, XW, g["node_row"], N, G_DIM, ws["drnn"], G_DIM, scatter=1)
            self.lstm.backward(pl, ws["drnn"], G_DIM)
        pl.reduce_into(ws, fp.grad)
        self.side.join()
        if self.relation_space:
            capi.basis_decompose(fp.w("gcn.conv1.att"), fp.w("gcn.conv1.basis"), ws["dWr"], self.R, NB, G_DIM * H1,
                                 fp.g("gcn.conv1.basis"), fp.g("gcn.conv1.att"))
        return ws["stats"]


IEMOCAP6_WEIGHTS = [1 / 0.086747, 1 / 0.144406, 1 / 0.227883, 1 / 0.160585, 1 / 0.127711, 1 / 0.252668]  # dgcn.py:109-110


class DGCNTrainer:
    """train_step / to_logits of track_mm/dgcn.py:96-134 (class-weighted CE, Adam lr 3e-4)."""

    def __init__(self, params, device):
        self.params, self.device = params, torch.device(device)
        torch.manual_seed(params.seed)
        self.model = DGCNModule(input_size=params.hidden_all, hidden_size=200, n_speakers=params.n_speakers,
                                n_classes=params.n_classes, compute=params.get("compute", "f32"),
                                seed=params.seed).finalize(self.device)
        o = params.optim
        self.optim = FusedAdam(self.model.flat, lr=o.lr, weight_decay=o.get("weight_decay", 0.0),
                               decoupled=(o.name == "AdamW"), seed=params.seed)
        self.model.rng_state = self.optim.rng_state
        self.class_weight = None
        if params.get("loss_weights", True):
            if params.n_classes != 6:
                raise ValueError("--loss_weights uses the six hard-coded IEMOCAP-6 inverse frequencies "
                                 "(dgcn.py:109-110); run %d-class datasets with --loss_weights=False" % params.n_classes)
            self.class_weight = torch.tensor(IEMOCAP6_WEIGHTS, dtype=torch.float32, device=self.device)

    def to_logits(self, batch):
        return self.model(**batch)[0]

    def prepare_batch(self, batch):
        out = {k: (v.to(self.device) if torch.is_tensor(v) else v) for k, v in batch.items()}
        tl = batch.get("text_length")
        if "n_nodes" not in out and torch.is_tensor(tl) and not tl.is_cuda:
            out["n_nodes"] = int(tl.sum())      # host tensor: no device sync when a batch carries no labels
        if self.model.compute == "bf16":
            out["input_tensor"] = out["input_tensor"].to(torch.bfloat16)
        return out

    def train_step(self, batch):
        self.model.train()
        stats = self.model.loss_and_grads(batch, self.class_weight)
        scale = all_reduce_grads(self.model.flat)
        self.optim.step(grad_scale=scale)
        return stats
