"""COGMEN on the MI355X hot path (drop-in for track_mm/cogmen.py:61-195).

``COGMENModule`` keeps the reference's constructor signature, ``state_dict``
keys and ``forward(**batch) -> (logits [N,C],
features [N,100])`` contract; every numeric step runs in libercgraft.so
(capi.py) -- there is no PyTorch fallback.

The 2-layer Transformer encoder ``rnn.0`` is dead in the reference's forward
(cogmen.py:145-147 feeds the RAW input to each module of ``self.rnn`` in turn
and keeps only the last result): its parameters are carried in the state dict
and never touched, exactly as in the reference where their grad stays None.

Forward chain (N valid utterances, F = 100, R = 8 relations):
  K1 graph  -> H0 = X[node_row] W1^T + b1 -> M = relation means | self  [N,9F]
  -> H1 = M [W_0..W_7;W_root] + b -> QKVS = H1 [Wq;Wk;Wv;Ws]^T + b
  -> H2 = segmented-softmax attention + skip -> H3 = LeakyReLU(BN(H2))
  -> Z = dropout(relu(H3 Wc0^T + b)) -> logits = Z Wc3^T + b -> CE
and the hand-written backward of each stage in reverse, weight gradients
accumulated as split-K slabs that one batched kernel reduces into the flat
gradient buffer (deterministic; no float atomics).
"""
import math
import os

import torch
from torch import nn

from . import capi
from .engine import WorkspaceCache, FlatParams, FusedAdam, GemmPlanner, SideStream, all_reduce_grads, linear_fwd, linear_wgrad, \
    matmul_wgrad_io

F_HID = 100
PA = 128              # row pitch (bf16 elements) of the 100-wide weight-gradient operands: 256-byte rows = two full cache lines
PM = 960              # row pitch of the relation-mean operand M [N, 900]: 1920 bytes = 15 cache lines
WP = WF = 5           # cogmen.py:153-154
N_REL = 8             # GNN(n_speakers=2) always: cogmen.py:62-64,114


def pick_heads(input_size, num_head):
    for h in range(6, num_head):       # cogmen.py:86-92
        if input_size % h == 0:
            return h
    raise AssertionError(input_size)


class _RGCNParams(nn.Module):
    """Parameter holder with torch_geometric RGCNConv's names / shapes / init (glorot, zeros)."""

    def __init__(self, cin, cout, R):
        super().__init__()
        self.weight = nn.Parameter(torch.empty(R, cin, cout))
        self.root = nn.Parameter(torch.empty(cin, cout))
        self.bias = nn.Parameter(torch.zeros(cout))
        for w in (self.weight, self.root):
            a = math.sqrt(6.0 / (w.size(-2) + w.size(-1)))
            nn.init.uniform_(w, -a, a)


class _TConvParams(nn.Module):
    """torch_geometric TransformerConv(heads=1) names: lin_key / lin_query / lin_value / lin_skip."""

    def __init__(self, cin, cout):
        super().__init__()
        self.lin_key = nn.Linear(cin, cout)
        self.lin_query = nn.Linear(cin, cout)
        self.lin_value = nn.Linear(cin, cout)
        self.lin_skip = nn.Linear(cin, cout)


class _GNNParams(nn.Module):
    def __init__(self, g_dim, h1_dim, h2_dim, n_speakers=2):
        super().__init__()
        self.conv1 = _RGCNParams(g_dim, h1_dim, 2 * n_speakers ** 2)
        self.conv2 = _TConvParams(h1_dim, h2_dim)
        self.bn = nn.BatchNorm1d(h2_dim)


class COGMENModule(nn.Module):
    def __init__(self, input_size, hidden_size, num_head, n_speakers, n_classes, compute="f32", seed=1,
                 chained_encoder=False):
        super().__init__()
        # chained_encoder (opt-in): rnn.1(rnn.0(x, key-padding mask)) -- the variant the encoder is
        # built for -- instead of the reference's rnn.1(x) with rnn.0's output discarded (cogmen.py:145-147)
        self.chained_encoder = chained_encoder
        self.enc_train = None
        assert hidden_size == F_HID, "the reference hard-codes 100 (cogmen.py:116-122)"
        self.input_size, self.n_speakers, self.n_classes = input_size, n_speakers, n_classes
        if compute not in ("f32", "bf16", "f32x2", "f32x3", "f32x32"):
            raise capi.ErcGraftError("COGMEN compute mode %r (f32 | bf16 | f32x2 | f32x3 | f32x32)" % (compute, ))
        self.compute = compute
        # SPLIT COMPUTE MODES (csrc/split_dev.h): fp32 data everywhere, every dense product on the bf16 matrix cores from operands
        # expanded into `terms` bf16 terms -- the fused 5-launch step structure of the bf16 mode at fp32-class accuracy
        # (north_star's 1e-4: two terms give 8e-6 on the logits of config 2, three are indistinguishable from fp32 arithmetic)
        # f32x32: three terms in the FORWARD products (projection, RGCN, QKVS), two in the backward ones (dH1, dH0, every weight
        # gradient).  The forward feeds ReLU / LeakyReLU kinks and the argmax: a forward deviation of 3e-6 (two terms) puts a unit on
        # the other side of its kink in about every other config-2 batch, which moves a gradient entry by 3e-3 of its tensor's scale;
        # the backward is linear in its operands (1e-5 of a gradient's scale with two terms, against fp32's own 5e-6).
        self.terms = {"f32x2": 2, "f32x3": 3, "f32x32": 3}.get(compute, 1)
        self.terms_bwd = {"f32x32": 2}.get(compute, self.terms)
        layer = nn.TransformerEncoderLayer(d_model=input_size, nhead=pick_heads(input_size, num_head),
                                           dropout=0.5, batch_first=True)
        encoder = nn.TransformerEncoder(layer, num_layers=2, enable_nested_tensor=False)  # dead (see module doc)
        self.rnn = nn.ModuleList([encoder, nn.Linear(input_size, hidden_size)])
        self.gcn = _GNNParams(hidden_size, hidden_size, hidden_size)
        self.cls = nn.Sequential(nn.Linear(100, 100), nn.ReLU(), nn.Dropout(p=0.5), nn.Linear(100, n_classes))
        self.drop_p = 0.5
        self.use_fused_graph = False   # set by finalize() in bf16 mode
        self.wgrad_bf16 = True         # bf16 mode: the graph part's and the projection's weight gradients on bf16 matrix cores
        self.fuse_head = True   # training path: csrc/head.hip instead of separate BN / Linear / CE launches
        self.fuse_project_graph = True   # bf16 mode: graph build inside the projection launch (csrc/cogmen_project.hip)
        # CAPACITY MODE (trainer.StepGraphs buckets): the batch tensors are capacity-sized static buffers -- B dialogues of
        # which some may have length 0, label [N_cap] -- and the true node count is whatever the lengths add up to: the
        # projection launch writes it to g["counts"][0] and every later kernel of the step reads it there (n_dev / k_dev),
        # so ONE captured HIP graph serves every batch that fits the bucket.  bf16 fused path only (supports_capacity).
        self.dynamic_n = False
        # set by the trainer: a FusedAdam whose step the bf16 weight-gradient launch applies itself (single-rank steps)
        self.fused_optim = None
        self.flat = None
        self._ws = WorkspaceCache()
        self._seed = seed

    # ------------------------------------------------------------------ setup
    def live_groups(self):
        g, c = self.gcn, self.cls
        named = lambda mod, pre, names: [(pre + n, getattr(mod, n)) for n in names]
        enc = []
        if self.chained_encoder:
            from .encoder import encoder_live_groups
            enc = encoder_live_groups(self.rnn[0])
        return enc + [
            [("rnn.1.weight", self.rnn[1].weight)], [("rnn.1.bias", self.rnn[1].bias)],
            named(g.conv1, "gcn.conv1.", ["weight", "root"]), [("gcn.conv1.bias", g.conv1.bias)],
            [("gcn.conv2.lin_query.weight", g.conv2.lin_query.weight), ("gcn.conv2.lin_key.weight", g.conv2.lin_key.weight),
             ("gcn.conv2.lin_value.weight", g.conv2.lin_value.weight), ("gcn.conv2.lin_skip.weight", g.conv2.lin_skip.weight)],
            [("gcn.conv2.lin_query.bias", g.conv2.lin_query.bias), ("gcn.conv2.lin_key.bias", g.conv2.lin_key.bias),
             ("gcn.conv2.lin_value.bias", g.conv2.lin_value.bias), ("gcn.conv2.lin_skip.bias", g.conv2.lin_skip.bias)],
            [("gcn.bn.weight", g.bn.weight)], [("gcn.bn.bias", g.bn.bias)],
            [("cls.0.weight", c[0].weight)], [("cls.0.bias", c[0].bias)],
            [("cls.3.weight", c[3].weight)], [("cls.3.bias", c[3].bias)],
        ]

    def finalize(self, device):
        """Move to ``device`` and pack the live parameters (call once, after loading a state dict)."""
        self.to(device)
        self.flat = FlatParams(self.live_groups(), device)
        self.rng_state = torch.tensor([0, self._seed], dtype=torch.int64, device=device)
        self.side = SideStream()
        self.w1_shadow = None     # bf16 copy of rnn.1.weight, valid only while an optimizer keeps it in sync
        self.shadows = None       # bf16 mode: every bf16 weight copy (capi.ShadowTable)
        self._shadow_auto = True  # nobody maintains the shadows: rebuild them at every forward
        if self.chained_encoder:
            from .encoder import EncoderTrain
            self.enc_train = EncoderTrain(self.rnn[0], self.flat, device, drop_p=0.5)
        elif self.compute == "bf16" or self.terms > 1:
            self._build_shadows()
        return self

    @property
    def fused_graph(self):
        """bf16 mode: the graph part runs as the two row-tile kernels of csrc/cogmen_fused.hip (bf16 matrix cores); split modes:
        the same kernels on term planes -- two-speaker graphs only (other speaker counts keep the unfused fp32 graph kernels)."""
        return self.shadows is not None and self.use_fused_graph and (self.terms == 1 or self.n_speakers == 2)

    def _build_shadows(self):
        """bf16 copies of the weights in the layouts the bf16 products read them (ercgraft.h, ErcShadowTab): the input
        projection's W1 as is, and the four packed / transposed copies of the fused graph kernels."""
        fp, D, F = self.flat, self.input_size, F_HID
        t = capi.ShadowTable(fp.device)
        nt = self.terms          # split modes: every range as `terms` planes (the bf16 expansion of the weights)
        off_cat, off_q = fp.offsets["gcn.conv1.weight"], fp.offsets["gcn.conv2.lin_query.weight"]
        assert fp.offsets["gcn.conv1.root"] == off_cat + N_REL * F * F
        assert fp.offsets["gcn.conv2.lin_skip.weight"] == off_q + 3 * F * F
        i_w1 = t.add(fp.offsets["rnn.1.weight"], F * D, F * D, D, F, (0, 1, 0), (1, 0, 0), D, 0, terms=nt)            # row-major copy
        # logical [n][k] operands in MFMA B-fragment order (ercgraft.h, mode 1)
        i_catT = t.add(off_cat, 9 * F * F, 7 * 29 * 512, F, 9 * F, (1, 0, 0), (0, 1, 0), 29, 1, terms=nt)           # WcatT[o][r*100+c]
        i_wb = t.add(off_cat, 9 * F * F, 7 * 30 * 512, F, F, (0, 1, 0), (1, 0, 104), 30, 1, terms=self.terms_bwd)   # Wb[c][r*104+o]
        i_q = t.add(off_q, 4 * F * F, 25 * 4 * 512, F, 4 * F, (0, 1, 0), (1, 0, 0), 4, 1, terms=nt)                 # Wq[n][k]
        i_qT = t.add(off_q, 4 * F * F, 7 * 13 * 512, F, 4 * F, (1, 0, 0), (0, 1, 0), 13, 1, terms=self.terms_bwd)   # WqT[k][n]
        t.seal()
        self.shadows = t
        self._sh = dict(w1=t.view(i_w1)[:F * D].view(F, D), catT=t.view(i_catT), wb=t.view(i_wb), q=t.view(i_q), qT=t.view(i_qT))
        self._sh_plane = dict(w1=t.plane(i_w1), catT=t.plane(i_catT), wb=t.plane(i_wb), q=t.plane(i_q), qT=t.plane(i_qT))
        self.use_fused_graph = nt == 1 or os.environ.get("ERC_SPLIT_TILES", "1") != "0"
        if nt > 1:     # (the planes are exact expansions of the weights: valid whoever keeps them in sync -- refreshed per forward
            self.w1_shadow = self._sh["w1"]      #  until an optimizer takes over, attach_bf16_shadow)

    def supports_capacity(self, batch):
        """Can a training step on ``batch`` run in capacity mode (every launch of the step takes the node count from the
        device)?  Needs the fused bf16 path end to end."""
        x, C, D = batch["input_tensor"], self.n_classes, self.input_size
        return bool(self.fused_graph and self.enc_train is None and self.w1_shadow is not None and self.fuse_head and
                    self.fuse_project_graph and self.wgrad_bf16 and x.dtype == (torch.float32 if self.terms > 1 else torch.bfloat16) and
                    x.dim() == 3 and C <= 8 and
                    D % 4 == 0 and batch["speaker_tensor"].dim() == 2 and
                    capi.cogmen_project_graph_ok(D, F_HID, x.shape[0], D, D))

    def refresh_shadows(self):
        if self.shadows is not None:
            capi.shadow_refresh(self.flat.data, self.flat.numel, self.shadows)

    def attach_bf16_shadow(self, optim):
        """bf16 mode: let the fused optimizer maintain the bf16 weight copies (input projection, fused graph kernels)."""
        self.refresh_shadows()
        self.w1_shadow = self._sh["w1"]
        self._shadow_auto = False
        optim.shadow_table = self.shadows

    @property
    def _last_ws(self):
        """workspace of the most recent forward (tests / bench read results out of it)"""
        return self._ws.last

    def _workspace(self, B, T, N, device):
        return self._ws.get((B, T, N, self.fused_graph), lambda: self._make_workspace(B, T, N, device))

    def _make_workspace(self, B, T, N, device):
        f32 = lambda *s: torch.empty(*s, dtype=torch.float32, device=device)
        i32 = lambda *s: torch.zeros(*s, dtype=torch.int32, device=device)
        E = N * (WP + WF + 1)
        C, F, D = self.n_classes, F_HID, self.input_size
        if self.fused_graph:
            return self._make_workspace_fused(B, T, N, device, E)
        g = dict(node_off=i32(B + 1), node_row=i32(N), node_spk=i32(N), in_ptr=i32(N + 1), in_src=i32(E),
                 in_typ=i32(E), out_ptr=i32(N + 1), out_dst=i32(E), out_typ=i32(E), out_eid=i32(E), counts=i32(2))
        ws = dict(
            g=g, E=E,
            H0=f32(N, F), M=f32(N, 9 * F), inv_cnt=f32(N, N_REL), H1=f32(N, F), QKVS=f32(N, 4 * F),
            alpha=f32(E), H2=f32(N, F), H3=f32(N, F), Z=f32(N, F), logits=f32(N, C),
            bn_saved=f32(2 * F), bn_ws=f32(capi.bn_ws_floats(F)), stats=torch.zeros(1024, dtype=torch.float32, device=device),
            bn_stats_ws=torch.zeros(capi.bn_batch_stats_ws_floats(F), dtype=torch.float32, device=device),
            head_ws=torch.zeros(capi.head_fused_ws_floats(N), dtype=torch.float32, device=device), bn_bwd=f32(2 * F),
            # (split modes: dlogits with a pitch of 8 floats -- 16-byte rows for the weight-gradient launch; pad columns stay zero)
            dlogits=torch.zeros(N, self.LDDL, dtype=torch.float32, device=device) if self.terms > 1 else f32(N, C),
            dZ=f32(N, F), dH3=f32(N, F), dH2=f32(N, F), dQKVS=f32(N, 4 * F), dscore=f32(E),
            dH1=f32(N, F), dM=f32(N, 9 * F), dH0=f32(N, F),
        )
        # slab space: forward split-K of the input projection + every weight gradient, sized generously
        slab = 16 * N * F + 8 * (F * D + 9 * F * F + 4 * F * F + 2 * F * F) + (1 << 20)
        ws["planner"] = GemmPlanner(device, slab, grad=self.flat.grad)
        ws["jobs"] = None
        return ws

    def _make_workspace_fused(self, B, T, N, device, E):
        """bf16 mode: M / H1 only exist as bf16 weight-gradient operands, no dM / dscore buffers."""
        f32 = lambda *s: torch.empty(*s, dtype=torch.float32, device=device)
        i32 = lambda *s: torch.zeros(*s, dtype=torch.int32, device=device)
        bf = lambda *s: torch.zeros(*s, dtype=torch.bfloat16, device=device)
        C, F, D = self.n_classes, F_HID, self.input_size
        g = dict(node_off=i32(B + 1), node_row=i32(N), node_spk=i32(N), in_ptr=i32(N + 1), in_src=i32(E),
                 in_typ=i32(E), out_ptr=i32(N + 1), out_dst=i32(E), out_typ=i32(E), out_eid=i32(E), counts=i32(2))
        if self.terms > 1:
            # split modes: every buffer between the launches is fp32 (the weight-gradient launch expands its operands itself);
            # zero-filled once: capacity mode reads rows beyond the true count (masked) and dlogits' pad columns
            z32 = lambda *s: torch.zeros(*s, dtype=torch.float32, device=device)
            ws = dict(
                g=g, E=E, fused=True,
                H0=f32(N, F), M=z32(N, 9 * F), inv_cnt=f32(N, N_REL), H1=z32(N, F), QKVS=f32(N, 4 * F),
                alpha=f32(E), H2=f32(N, F), H3=z32(N, F), Z=z32(N, F), logits=f32(N, C),
                bn_saved=f32(2 * F), bn_ws=f32(capi.bn_ws_floats(F)), stats=torch.zeros(1024, dtype=torch.float32, device=device),
                bn_stats_ws=torch.zeros(capi.bn_batch_stats_ws_floats(F), dtype=torch.float32, device=device),
                bn_tile_ws=torch.zeros(capi.cogmen_fwd_tile_ws_doubles(N), dtype=torch.float64, device=device),
                head_ws=torch.zeros(capi.head_fused_ws_floats(N), dtype=torch.float32, device=device), bn_bwd=f32(2 * F),
                dlogits=z32(N, self.LDDL), dZ=z32(N, F), dH3=f32(N, F), dQKVS=z32(N, 4 * F), dH1=z32(N, F), dH0=z32(N, F),
            )
            slab = 16 * N * F + 8 * (F * D + 9 * F * F + 4 * F * F + 2 * F * F) + (1 << 20)
            ws["planner"] = GemmPlanner(device, slab, grad=self.flat.grad)
            ws["jobs"] = None
            return ws
        ws = dict(
            g=g, E=E, fused=True,
            H0=f32(N, F), Mb=bf(N, PM), inv_cnt=f32(N, N_REL), H1b=bf(N, PA), QKVS=f32(N, 4 * F),
            alpha=f32(E), H2=f32(N, F), H3=f32(N, F), Z=f32(N, F), logits=f32(N, C),
            bn_saved=f32(2 * F), bn_ws=f32(capi.bn_ws_floats(F)), stats=torch.zeros(1024, dtype=torch.float32, device=device),
            bn_stats_ws=torch.zeros(capi.bn_batch_stats_ws_floats(F), dtype=torch.float32, device=device),
            bn_tile_ws=torch.zeros(capi.cogmen_fwd_tile_ws_doubles(N), dtype=torch.float64, device=device),
            head_ws=torch.zeros(capi.head_fused_ws_floats(N), dtype=torch.float32, device=device), bn_bwd=f32(2 * F),
            dlogits=f32(N, C), dZ=f32(N, F), dH3=f32(N, F), dQKVS=f32(N, 4 * F), dH1=f32(N, F), dH0=f32(N, F),
            # bf16 operands of the weight-gradient launch (csrc/wgrad_bf16.hip), written by the head / backward tile kernels;
            # zero-filled once: the pad columns are read (into output rows nobody stores) and must stay finite
            H3b=bf(N, PA), Zb=bf(N, PA), dZb=bf(N, PA), dlb=bf(N, 8), dQKVSb=bf(N, 4 * F), dH1b=bf(N, PA), dH0b=bf(N, PA),
        )
        slab = 16 * N * F + 8 * (F * D + 9 * F * F + 4 * F * F + 2 * F * F) + (1 << 20)
        ws["planner"] = GemmPlanner(device, slab, grad=self.flat.grad)
        ws["jobs"] = None
        return ws

    LDDL = 8                # split modes: row pitch of dlogits
    BN_FUSED_MAX_N = 8192   # above: the tile partials are too many for one last arriver, BatchNorm statistics get their own launch

    # ---------------------------------------------------------------- forward
    def _shape(self, input_tensor, text_length, label, n_nodes=None):
        B, T = input_tensor.shape[0], input_tensor.shape[1]
        if label is not None:
            N = int(label.shape[0])          # label is [N]: no device sync needed
        elif n_nodes is not None:
            N = int(n_nodes)                 # host-side count stashed by prepare_batch: no device sync either
        else:
            N = int(text_length.sum().item())
        return B, T, N

    def _forward_impl(self, x, speaker_tensor, text_length, B, T, N, training, upto_h2=False, desc=None):
        """``desc`` (int32 [2 B]: lengths | first store rows): RESIDENT batch -- x [U, D] / speaker_tensor [U] are a feature
        store's arrays, B / T / N are capacities (capacity mode is implied), no padded block exists."""
        fp, dev = self.flat, x.device
        ws = self._workspace(B, T, N, dev)
        g, pl = ws["g"], ws["planner"]
        pl.reset()
        F, C, D = F_HID, self.n_classes, self.input_size
        x_bf16 = x.dtype == torch.bfloat16
        if self.shadows is not None and self._shadow_auto:
            self.refresh_shadows()
        # bf16 mode: the projection's workgroups build the window graph themselves (csrc/cogmen_project.hip): one launch
        # and one launch gap less than graph build + projection
        split = self.terms > 1
        project_graph = (self.fuse_project_graph and (x.dtype == torch.float32 if split else x_bf16) and self.w1_shadow is not None
                         and self.enc_train is None
                         and speaker_tensor.dim() == (1 if desc is not None else 2) and x.is_contiguous()
                         and (N <= self.BN_FUSED_MAX_N or split)      # beyond: many row groups per workgroup, the separate bf16 launches win (B = 512: 45 vs 54 us;
                                                                        #  split modes: 131 us fused vs 235 + 10 on the exact-fp32 kernels)
                         and capi.cogmen_project_graph_ok(D, F, B, D, D))
        if desc is not None and not (project_graph and self.dynamic_n):
            raise capi.ErcGraftError("COGMEN resident batches need the fused bf16 training path in capacity mode")
        if self.dynamic_n and not (project_graph and ws.get("fused") and upto_h2 and N <= self.BN_FUSED_MAX_N):
            raise capi.ErcGraftError("COGMEN capacity mode needs the fused bf16 training path (supports_capacity)")
        nd = g["counts"] if self.dynamic_n else None
        if project_graph:
            capi.cogmen_project_graph(x, D, self.w1_shadow, D, fp.w("rnn.1.bias"), ws["H0"], F, F, D, text_length,
                                      speaker_tensor, B, T, WP, WF, self.n_speakers, N, ws["E"], g, desc=desc,
                                      terms=self.terms, w_plane=self._sh_plane["w1"])
        else:
            capi.window_graph_build(text_length, speaker_tensor, speaker_tensor.stride(0), speaker_tensor.stride(1),
                                    B, T, WP, WF, self.n_speakers, N, ws["E"], g)
        if self.enc_train is not None:      # chained mode: the projection reads the encoder's bf16 output
            x = self.enc_train.forward(x, text_length, training, self.rng_state)
            x_bf16 = True
            ws["x_enc"] = x
        W1 = self.w1_shadow if (x_bf16 and self.w1_shadow is not None and not split) else fp.w("rnn.1.weight")
        if not project_graph:
            linear_fwd(pl, x, D, g["node_row"], W1, fp.w("rnn.1.bias"), ws["H0"], F, N, F, D, x_bf16=x_bf16)
        if ws.get("fused"):
            bn = self.gcn.bn
            # training with the fused head: BatchNorm's batch statistics come out of the same launch
            ws["bn_in_tile"] = bool(upto_h2 and N <= self.BN_FUSED_MAX_N)   # tile sums here, finalised by the head kernel
            tile_kw = dict(bn_fused=2 if ws["bn_in_tile"] else 0, running_mean=bn.running_mean, running_var=bn.running_var,
                           momentum=bn.momentum, eps=bn.eps, saved=ws["bn_saved"], bn_ws=ws["bn_tile_ws"], n_speakers=self.n_speakers,
                           n_dev=nd, health=fp.health if training else None, events=fp.events if training else None)
            if split:      # fp32 operands out, weights as term planes (erc_cogmen_fwd_tile_x)
                capi.cogmen_fwd_tile(ws["H0"], F, N, WP, WF, g, self._sh["catT"], fp.w("gcn.conv1.bias"), self._sh["q"],
                                     fp.w("gcn.conv2.lin_query.bias"), 1.0 / math.sqrt(F), ws["M"], 9 * F, ws["inv_cnt"],
                                     ws["H1"], F, ws["QKVS"], ws["H2"], F, ws["alpha"], terms=self.terms,
                                     catT_plane=self._sh_plane["catT"], q_plane=self._sh_plane["q"], **tile_kw)
            else:
                capi.cogmen_fwd_tile(ws["H0"], F, N, WP, WF, g, self._sh["catT"], fp.w("gcn.conv1.bias"), self._sh["q"],
                                     fp.w("gcn.conv2.lin_query.bias"), 1.0 / math.sqrt(F), ws["Mb"], PM, ws["inv_cnt"],
                                     ws["H1b"], PA, ws["QKVS"], ws["H2"], F, ws["alpha"], **tile_kw)
            if upto_h2:
                return ws
            return self._forward_tail(ws, N, training)
        capi.rgcn_mean_fwd(ws["H0"], F, F, N_REL, N, g, ws["M"], 9 * F, ws["inv_cnt"])
        # H1 = M @ [W_r ; W_root] + bias : B operand is the [9F, F] k-major stack conv1.weight|conv1.root
        Wcat = fp.w("gcn.conv1.weight")
        S = pl.split_for(N, F, 9 * F)
        if S == 1:
            capi.gemm_f32(ws["M"], 9 * F, 0, None, Wcat, F, 1, None, ws["H1"], F, N, F, 9 * F,
                          bias=fp.w("gcn.conv1.bias"))
        else:
            src = pl.take(S * N * F)
            capi.gemm_f32(ws["M"], 9 * F, 0, None, Wcat, F, 1, None, pl.ws[src:], F, N, F, 9 * F, split_k=S,
                          c_slab=N * F)
            capi.slab_reduce(pl.ws[src:], S, N * F, fp.w("gcn.conv1.bias"), F, 0, ws["H1"], N * F)
        linear_fwd(pl, ws["H1"], F, None, fp.w("gcn.conv2.lin_query.weight"), fp.w("gcn.conv2.lin_query.bias"),
                   ws["QKVS"], 4 * F, N, 4 * F, F)
        capi.tconv_attn_fwd(ws["QKVS"], 4 * F, F, N, 1.0 / math.sqrt(F), g, ws["H2"], F, ws["alpha"])
        if upto_h2:      # the training path runs everything behind H2 in the fused head kernel
            return ws
        return self._forward_tail(ws, N, training)

    def _forward_tail(self, ws, N, training):
        fp, pl = self.flat, ws["planner"]
        F, C = F_HID, self.n_classes
        bn = self.gcn.bn
        capi.bn_lrelu_fwd(ws["H2"], F, N, F, fp.w("gcn.bn.weight"), fp.w("gcn.bn.bias"), bn.running_mean,
                          bn.running_var, bn.momentum, bn.eps, 0.01, training, ws["bn_saved"], ws["H3"], F,
                          ws["bn_ws"])
        p = self.drop_p if training else 0.0
        linear_fwd(pl, ws["H3"], F, None, fp.w("cls.0.weight"), fp.w("cls.0.bias"), ws["Z"], F, N, F, F,
                   act=3 if p > 0 else 1, drop_p=p, rng=self.rng_state)
        linear_fwd(pl, ws["Z"], F, None, fp.w("cls.3.weight"), fp.w("cls.3.bias"), ws["logits"], C, N, C, F)
        return ws

    def check_cluster(self):
        """Raise if a bounded wait between cooperating workgroups timed out since the last call: the splits of a gradient tile
        in the weight-gradient launch with the optimizer inside (csrc/wgrad_bf16.hip), or the peer-to-peer gradient exchange
        (csrc/optim.hip).  These launches give up per gradient tile / chunk: the affected steps' updates are PARTIAL (and ranks may
        have diverged) -- the run is to be treated as failed; trainer.run calls this once per epoch."""
        self.flat.check_health("COGMEN weight-gradient / optimizer launch", partial=True)

    def forward(self, input_tensor, speaker_tensor, text_length, *args, label=None, **kwargs):
        if self.flat is None:
            raise capi.ErcGraftError("call COGMENModule.finalize(device) before forward")
        B, T, N = self._shape(input_tensor, text_length, label, kwargs.get("n_nodes"))
        ws = self._forward_impl(input_tensor, speaker_tensor, text_length, B, T, N, self.training)
        return ws["logits"], ws["H0"]

    # --------------------------------------------------------------- training
    def loss_and_grads(self, batch, class_weight=None):
        """Forward in the module's current mode, cross entropy, full backward into ``flat.grad``.
        Returns the stats tensor {loss, #correct, weight sum} (device, no sync)."""
        x, spk, lens, ys = batch["input_tensor"], batch["speaker_tensor"], batch["text_length"], batch["label"]
        desc = batch.get("desc")          # resident batch (trainer.ResidentEpochs): store arrays + 2 B int32 of batch description
        B, T, N = batch["caps"] if desc is not None else self._shape(x, lens, ys)
        training = self.training
        F, C, D = F_HID, self.n_classes, self.input_size
        # fused head (csrc/head.hip): BatchNorm apply .. cross entropy .. BatchNorm-backward sums in one launch
        fused_head = self.fuse_head and training and C <= 8 and F % 4 == 0 and F <= 100
        ws = self._forward_impl(x, spk, lens, B, T, N, training, upto_h2=fused_head, desc=desc)
        fp, g, pl = self.flat, ws["g"], ws["planner"]
        x_bf16 = x.dtype == torch.bfloat16
        p = self.drop_p if training else 0.0
        bn = self.gcn.bn
        fused = bool(ws.get("fused"))
        if fused and not fused_head:
            raise capi.ErcGraftError("COGMEN bf16 mode trains through the fused head (C <= 8)")
        # bf16 mode: every weight gradient of the step on the bf16 matrix cores from bf16 operands (csrc/wgrad_bf16.hip)
        # (N > 8 192: the planner takes the WIDE form of that kernel -- four column tiles per workgroup; the K-split form lost
        #  there against the 64 x 64-tile kernel, 181 vs 167 us at N = 33 k, because it streams the A operand once per tile)
        split = self.terms > 1
        w16 = bool(fused and not split and self.wgrad_bf16 and x_bf16 and C <= 8 and D % 4 == 0 and x.is_contiguous() and x.data_ptr() % 8 == 0
                   and (N <= self.BN_FUSED_MAX_N or os.environ.get("ERC_W2_WIDE", "1") != "0"))
        # split modes: the same launch from the fp32 operands, expanded into bf16 terms in registers (erc_wgrad_split)
        wsp = bool(split and fused_head and x.dtype == torch.float32 and D % 4 == 0 and x.is_contiguous() and x.data_ptr() % 16 == 0
                   and self.enc_train is None)
        if split and not wsp:
            raise capi.ErcGraftError("COGMEN %s mode: fp32 contiguous features, D %% 4 == 0, C <= 8, training through the fused head"
                                     % self.compute)
        ws["w16"], ws["wsp"] = w16, wsp
        pl.split_terms = self.terms_bwd if wsp else 1
        b16 = (ws["H3b"], ws["Zb"], ws["dZb"], ws["dlb"], PA) if w16 else None
        lddl = self.LDDL if split else 0
        nd = g["counts"] if self.dynamic_n else None
        if self.dynamic_n and not ((w16 or wsp) and fused_head):
            raise capi.ErcGraftError("COGMEN capacity mode needs the fused bf16 / split training path (supports_capacity)")
        if fused_head:
            # (bf16 mode: the weight-gradient launch reads the bf16 copies of H3 / Z / dZ / dlogits, nothing reads the fp32 ones: the
            #  head does not write them -- 1.2 KB of its 2.4 KB per row; ERC_HEAD_F32_OUT=1 keeps them for inspection)
            f32o = not w16 or os.environ.get("ERC_HEAD_F32_OUT", "0") == "1"
            head_args = (ws["H2"], F, N, F, C, fp.w("gcn.bn.weight"), fp.w("gcn.bn.bias"), ws["bn_saved"], 0.01,
                         fp.w("cls.0.weight"), fp.w("cls.0.bias"), fp.w("cls.3.weight"), fp.w("cls.3.bias"), ys,
                         class_weight, p, self.rng_state if p > 0 else None, ws["H3"] if f32o else None, ws["Z"] if f32o else None,
                         ws["logits"], ws["dlogits"] if f32o else None, ws["dZ"] if f32o else None, ws["dH3"], ws["bn_bwd"],
                         fp.g("gcn.bn.weight"), fp.g("gcn.bn.bias"), ws["stats"], ws["head_ws"])
            if fused and ws["bn_in_tile"]:
                # BatchNorm's batch statistics: per-tile sums from the forward tile kernel, added up by every head workgroup
                # ... and the head's own cross-workgroup sums (BatchNorm backward means, loss) are left to the backward tile kernel
                capi.head_fused_bn(*head_args, ws["bn_tile_ws"][2:].view(torch.float32), -(-N // 16), bn.running_mean,
                                   bn.running_var, bn.momentum, bn.eps, defer_reduce=True, bf16_out=b16, n_dev=nd,
                                   label_rows=g["node_row"] if desc is not None else None, lddl=lddl)
                ws["head_deferred"] = True
            else:
                capi.bn_batch_stats(ws["H2"], F, N, F, bn.running_mean, bn.running_var, bn.momentum, bn.eps, ws["bn_saved"],
                                    ws["bn_stats_ws"])
                capi.head_fused(*head_args, bf16_out=b16, lddl=lddl)
                ws["head_deferred"] = False
        else:
            capi.cross_entropy(ws["logits"], C, C, N, None, ys, class_weight, 1.0, ws["dlogits"], C, ws["stats"])
            capi.gemm_f32(ws["dlogits"], C, 0, None, fp.w("cls.3.weight"), F, 1, None, ws["dZ"], F, N, F, C,
                          act=2, aux=ws["Z"], ldaux=F, act_scale=1.0 / (1.0 - p))
        if w16:
            pl.defer16(ws["Zb"], PA, ws["dlb"], 8, fp.g("cls.3.weight"), F, F, C, N, ct=True, bias_b=fp.g("cls.3.bias"), k_dev=nd)
            pl.defer16(ws["dZb"], PA, ws["H3b"], PA, fp.g("cls.0.weight"), F, F, F, N, bias_a=fp.g("cls.0.bias"), k_dev=nd)
        elif wsp:
            pl.defer16(ws["Z"], F, ws["dlogits"], self.LDDL, fp.g("cls.3.weight"), F, F, C, N, ct=True, bias_b=fp.g("cls.3.bias"), k_dev=nd)
            pl.defer16(ws["dZ"], F, ws["H3"], F, fp.g("cls.0.weight"), F, F, F, N, bias_a=fp.g("cls.0.bias"), k_dev=nd)
        else:
            with self.side.fork():
                linear_wgrad(pl, ws["dlogits"], C, ws["Z"], F, None, C, F, N, fp.offsets["cls.3.weight"],
                             fp.offsets["cls.3.bias"], defer=True)
            with self.side.fork():
                linear_wgrad(pl, ws["dZ"], F, ws["H3"], F, None, F, F, N, fp.offsets["cls.0.weight"],
                             fp.offsets["cls.0.bias"], defer=True)
        if fused:
            self._backward_fused(ws, x, x_bf16, N)
            return ws["stats"]
        bn_prologue = None
        if fused_head:   # BatchNorm's elementwise backward runs inside the attention backward (one launch less)
            bn_prologue = (ws["H2"], F, fp.w("gcn.bn.weight"), ws["bn_saved"], ws["bn_bwd"], ws["dH2"])
        else:
            capi.gemm_f32(ws["dZ"], F, 0, None, fp.w("cls.0.weight"), F, 1, None, ws["dH3"], F, N, F, F)
            # BatchNorm + LeakyReLU
            capi.bn_lrelu_bwd(ws["H2"], F, N, F, fp.w("gcn.bn.weight"), fp.w("gcn.bn.bias"), ws["bn_saved"], 0.01,
                              ws["dH3"], F, ws["dH2"], F, fp.g("gcn.bn.weight"), fp.g("gcn.bn.bias"), ws["bn_ws"])
        # TransformerConv
        capi.tconv_attn_bwd(ws["QKVS"], 4 * F, F, N, 1.0 / math.sqrt(F), g, ws["alpha"],
                            ws["dH3"] if fused_head else ws["dH2"], F, ws["dQKVS"], ws["dscore"], bn=bn_prologue)
        capi.gemm_f32(ws["dQKVS"], 4 * F, 0, None, fp.w("gcn.conv2.lin_query.weight"), F, 1, None, ws["dH1"], F,
                      N, F, 4 * F)
        # RGCN: dM = dH1 @ Wcat^T ; dWcat = M^T dH1 ; dbias = colsum(dH1)
        capi.gemm_f32(ws["dH1"], F, 0, None, fp.w("gcn.conv1.weight"), F, 0, None, ws["dM"], 9 * F, N, 9 * F, F)
        capi.rgcn_mean_bwd(ws["dM"], 9 * F, F, N_REL, N, g, ws["inv_cnt"], ws["dH0"], F)
        if wsp:      # split modes with the graph part on the unfused fp32 kernels: weight gradients + optimizer as in the fused path
            self._split_wgrads(ws, x, N)
            return ws["stats"]
        with self.side.fork():
            linear_wgrad(pl, ws["dQKVS"], 4 * F, ws["H1"], F, None, 4 * F, F, N,
                         fp.offsets["gcn.conv2.lin_query.weight"], fp.offsets["gcn.conv2.lin_query.bias"], defer=True)
        with self.side.fork():
            matmul_wgrad_io(pl, ws["M"], 9 * F, ws["dH1"], F, 9 * F, F, N, fp.offsets["gcn.conv1.weight"],
                            fp.offsets["gcn.conv1.bias"], defer=True)
        # input projection (no gradient into the features)
        if self.enc_train is not None:
            x, x_bf16 = ws["x_enc"], True
        with self.side.fork():
            linear_wgrad(pl, ws["dH0"], F, x, D, g["node_row"], F, D, N, fp.offsets["rnn.1.weight"],
                         fp.offsets["rnn.1.bias"], x_bf16=x_bf16, defer=True)
        self.side.join()
        pl.reduce_into(ws, fp.grad)
        if self.enc_train is not None:
            # d(encoder output) at the valid rows = dH0 W1; the encoder backward reads it through the inverse row map
            ews = self.enc_train._last
            if ews.get("dXn") is None or ews["dXn"].shape[0] != N:
                ews["dXn"] = torch.zeros(N, D, dtype=torch.float32, device=x.device)
            capi.gemm_f32(ws["dH0"], F, 0, None, fp.w("rnn.1.weight"), D, 1, None, ews["dXn"], D, N, D, F)
            capi.enc_inverse_rows(g["node_row"], N, ews["inv"], B * T)
            self.enc_train.backward(ews["dXn"], ews["inv"])
        return ws["stats"]

    def _split_wgrads(self, ws, x, N):
        """split modes: the graph part's and the projection's weight gradients from fp32 operands (erc_wgrad_split), BatchNorm's
        scale / shift as finished ranges, the optimizer inside the launch when the trainer attached one"""
        fp, g, pl = self.flat, ws["g"], ws["planner"]
        F, D = F_HID, self.input_size
        nd = g["counts"] if self.dynamic_n else None
        pl.defer16(ws["H1"], F, ws["dQKVS"], 4 * F, fp.g("gcn.conv2.lin_query.weight"), F, F, 4 * F, N, ct=True,
                   bias_b=fp.g("gcn.conv2.lin_query.bias"), k_dev=nd)
        pl.defer16(ws["dH1"], F, ws["M"], 9 * F, fp.g("gcn.conv1.weight"), F, F, 9 * F, N, ct=True,
                   bias_a=fp.g("gcn.conv1.bias"), k_dev=nd)
        pl.defer16(ws["dH0"], F, x, D, fp.g("rnn.1.weight"), D, F, D, N, bias_a=fp.g("rnn.1.bias"), gather=g["node_row"], k_dev=nd)
        o_g, o_b = fp.offsets["gcn.bn.weight"], fp.offsets["gcn.bn.bias"]
        pl.defer16_range(fp.grad[o_g:o_g + F])
        pl.defer16_range(fp.grad[o_b:o_b + F])
        pl.fused_adam = self.fused_optim
        pl.reduce_into(ws, fp.grad)

    def _backward_fused(self, ws, x, x_bf16, N):
        """bf16 mode: BatchNorm backward .. dH0 in one launch (csrc/cogmen_fused.hip), then the batched weight gradients."""
        fp, g, pl = self.flat, ws["g"], ws["planner"]
        F, D = F_HID, self.input_size
        w16 = ws["w16"]
        nd = g["counts"] if self.dynamic_n else None
        head_kw = dict(head_part=ws["head_ws"], head_parts=-(-N // capi.head_fused_rows_per_workgroup(N)), dgamma=fp.g("gcn.bn.weight"), dbeta=fp.g("gcn.bn.bias"),
                       stats=ws["stats"]) if ws.get("head_deferred") else {}
        bwd_args = (ws["dH3"], ws["H2"], F, N, WP, WF, fp.w("gcn.bn.weight"), ws["bn_saved"], ws["bn_bwd"], ws["QKVS"],
                    ws["alpha"], g, ws["inv_cnt"], self._sh["qT"], self._sh["wb"], 1.0 / math.sqrt(F))
        if ws.get("wsp"):      # split modes: fp32 gradient tiles, weights as term planes; then the weight gradients + optimizer
            capi.cogmen_bwd_tile(*bwd_args, ws["dQKVS"], ws["dH1"], ws["dH0"], F, n_speakers=self.n_speakers, lddh1=F, n_dev=nd,
                                 terms=self.terms_bwd, qT_plane=self._sh_plane["qT"], wb_plane=self._sh_plane["wb"], **head_kw)
            self._split_wgrads(ws, x, N)
            return
        if w16:
            # the three gradients the backward hands to the weight-gradient launch are written as bf16 (nothing else reads them)
            capi.cogmen_bwd_tile(*bwd_args, ws["dQKVSb"], ws["dH1b"], ws["dH0b"], PA, n_speakers=self.n_speakers,
                                 grads_bf16=True, lddh1=PA, n_dev=nd, **head_kw)
            pl.defer16(ws["H1b"], PA, ws["dQKVSb"], 4 * F, fp.g("gcn.conv2.lin_query.weight"), F, F, 4 * F, N, ct=True,
                       bias_b=fp.g("gcn.conv2.lin_query.bias"), k_dev=nd)
            pl.defer16(ws["dH1b"], PA, ws["Mb"], PM, fp.g("gcn.conv1.weight"), F, F, 9 * F, N, ct=True,
                       bias_a=fp.g("gcn.conv1.bias"), k_dev=nd)
            pl.defer16(ws["dH0b"], PA, x, D, fp.g("rnn.1.weight"), D, F, D, N, bias_a=fp.g("rnn.1.bias"), gather=g["node_row"],
                       k_dev=nd)
            # BatchNorm's scale / shift gradients were completed by the backward tile launch: with the optimizer fused into
            # the weight-gradient launch, one of its work items updates them
            o_g, o_b = fp.offsets["gcn.bn.weight"], fp.offsets["gcn.bn.bias"]
            pl.defer16_range(fp.grad[o_g:o_g + F])
            pl.defer16_range(fp.grad[o_b:o_b + F])
            pl.fused_adam = self.fused_optim
            pl.reduce_into(ws, fp.grad)
            return
        capi.cogmen_bwd_tile(*bwd_args, ws["dQKVS"], ws["dH1"], ws["dH0"], F, n_speakers=self.n_speakers, **head_kw)
        pl.mma_bf16 = self.wgrad_bf16     # these three products on bf16 matrix cores (the head's stay fp32)
        linear_wgrad(pl, ws["dQKVS"], 4 * F, ws["H1b"], PA, None, 4 * F, F, N,
                     fp.offsets["gcn.conv2.lin_query.weight"], fp.offsets["gcn.conv2.lin_query.bias"], defer=True)
        matmul_wgrad_io(pl, ws["Mb"], PM, ws["dH1"], F, 9 * F, F, N, fp.offsets["gcn.conv1.weight"],
                        fp.offsets["gcn.conv1.bias"], defer=True)
        linear_wgrad(pl, ws["dH0"], F, x, D, g["node_row"], F, D, N, fp.offsets["rnn.1.weight"],
                     fp.offsets["rnn.1.bias"], x_bf16=x_bf16, defer=True)
        pl.mma_bf16 = False
        pl.reduce_into(ws, fp.grad)

    def sync_buffers(self, optimizer_steps):
        """BatchNorm1d.num_batches_tracked is bookkeeping only (momentum is fixed): it is set from the optimizer's
        device step counter when a checkpoint is taken instead of costing a launch per step."""
        self.gcn.bn.num_batches_tracked.fill_(int(optimizer_steps))

    def last_graph(self, B, T, N):
        return self._workspace(B, T, N, self.flat.device)["g"]

    def dominant_kernel_probe(self, batch, reps=200):
        """Time the HBM-dominant kernel of the step -- the input projection H0 = X[node_row] W1^T, the only
        kernel that touches the [B,T,D] feature block in the forward -- with HIP events on the stream it is
        launched on: ``reps`` back-to-back launches (one HIP-graph replay) between one event pair; inter-launch
        gaps are included, so the figure is conservative w.r.t. rocprofv3's per-dispatch duration.
        Algorithmic bytes per launch: N*D*sizeof(x) + F*D*4 + N*F*4 + N*4."""
        x, lens, ys = batch["input_tensor"], batch["text_length"], batch["label"]
        B, T, N = self._shape(x, lens, ys)
        ws = self._workspace(B, T, N, x.device)
        pl, g, fp = ws["planner"], ws["g"], self.flat
        F, D = F_HID, self.input_size
        x_bf16 = x.dtype == torch.bfloat16
        spk = batch["speaker_tensor"]
        split = self.terms > 1
        fusedpg = (self.fuse_project_graph and (x.dtype == torch.float32 if split else x_bf16) and self.w1_shadow is not None
                   and self.enc_train is None and spk.dim() == 2 and capi.cogmen_project_graph_ok(D, F, B, D, D))
        extra = 0
        if fusedpg:     # what the bf16 / split step launches: projection + window graph in one kernel (csrc/cogmen_project.hip)
            launch = lambda: capi.cogmen_project_graph(x, D, self.w1_shadow, D, fp.w("rnn.1.bias"), ws["H0"], F, F, D, lens, spk, B, T,
                                                       WP, WF, self.n_speakers, N, ws["E"], g, terms=self.terms,
                                                       w_plane=self._sh_plane["w1"])
            name = ("cogmen_project_graph_kernel<terms=%d> (input projection, fp32 features expanded into bf16 terms, weight planes "
                    "resident in registers, + the window graph)" % self.terms) if split else \
                "cogmen_project_graph_kernel (input projection, bf16 features, weights resident in registers, + the window graph)"
            extra = int(g["counts"][1]) * 17 + N * 12 + B * 16      # CSR arrays written, lengths / speakers read
        elif x_bf16:
            W1 = self.w1_shadow if self.w1_shadow is not None else fp.w("rnn.1.weight")
            launch = lambda: capi.gemm_bf16a_stream(x, D, g["node_row"], W1, D, ws["H0"], F, N, F, D, bias=fp.w("rnn.1.bias"))
            name = ("gemm_bf16a_persist_kernel (input projection, bf16 features, weights resident in registers)"
                    if (self.w1_shadow is not None and N >= 1024) else
                    "gemm_bf16a_stream_kernel<8,%s,6> (input projection, bf16 features)" % (
                        "true" if self.w1_shadow is not None else "false"))
        else:
            launch = lambda: capi.gemm_f32(x, D, 0, g["node_row"], fp.w("rnn.1.weight"), D, 0, None, ws["H0"], F, N, F, D,
                                           bias=fp.w("rnn.1.bias"))
            name = "gemm_f32_stream_kernel<0,0,8> (input projection, fp32 features)"
        S = 1
        wbytes = 2 * self.terms if ((x_bf16 or split) and self.w1_shadow is not None) else 4
        nbytes = N * D * x.element_size() + F * D * wbytes + N * F * 4 + N * 4 + extra
        for _ in range(10):
            launch()
        torch.cuda.synchronize()
        # the reps launches are captured into one HIP graph and replayed: eager launches from Python are host-bound
        # (~10 us per ctypes call), which would time the interpreter, not the kernel
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(graph):
            for _ in range(reps):
                launch()
        graph.replay()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        graph.replay()
        e1.record()
        torch.cuda.synchronize()
        us = e0.elapsed_time(e1) * 1e3 / reps
        return {"kernel": name, "us": us, "bytes": nbytes, "gbs": nbytes / us * 1e-3, "split_k": S}


def build_graph_tensors(text_length, speaker_tensor, wp, wf, n_speakers, n_nodes=None, explicit=True):
    """Device graph builder as a standalone op: returns the CSR dict plus, when ``explicit``, the
    reference-shaped ``edge_index [2,E]`` / ``edge_type [E]`` int64 tensors (cogmen_utils.py:139-142),
    in canonical (target, source) order."""
    dev = text_length.device
    if speaker_tensor.dim() != 2:
        raise capi.ErcGraftError("speaker_tensor must be [B,T] integer ids")
    B, T = speaker_tensor.shape
    N = int(n_nodes) if n_nodes is not None else int(text_length.sum().item())
    w = (wp if wp >= 0 else T) + (wf if wf >= 0 else T) + 1
    E = max(1, N * min(w, T))
    i32 = lambda *s: torch.zeros(*s, dtype=torch.int32, device=dev)
    g = dict(node_off=i32(B + 1), node_row=i32(max(N, 1)), node_spk=i32(max(N, 1)), in_ptr=i32(N + 1),
             in_src=i32(E), in_typ=i32(E), out_ptr=i32(N + 1), out_dst=i32(E), out_typ=i32(E), out_eid=i32(E),
             counts=i32(2))
    ei = torch.zeros(2, E, dtype=torch.int64, device=dev) if explicit else None
    et = torch.zeros(E, dtype=torch.int64, device=dev) if explicit else None
    capi.window_graph_build(text_length, speaker_tensor, speaker_tensor.stride(0), speaker_tensor.stride(1),
                            B, T, wp, wf, n_speakers, N, E, g, ei, et)
    g["e_cap"] = E
    return g, ei, et


class COGMENTrainer:
    """train_step / to_logits of track_mm/cogmen.py:163-195 without lumo."""

    def __init__(self, params, device):
        self.params, self.device = params, torch.device(device)
        torch.manual_seed(params.seed)
        self.model = COGMENModule(input_size=params.hidden_all, hidden_size=100,
                                  num_head=params.get("num_heads", 17), n_speakers=params.n_speakers,
                                  n_classes=params.n_classes, compute=params.get("compute", "f32"),
                                  seed=params.seed,
                                  chained_encoder=params.get("chained_encoder", False)).finalize(self.device)
        o = params.optim
        self.optim = FusedAdam(self.model.flat, lr=o.lr, weight_decay=o.weight_decay,
                               decoupled=(o.name == "AdamW"), seed=params.seed)
        self.model.rng_state = self.optim.rng_state   # dropout offset advances with the optimizer step
        self.optim.skip_flag = self.model.flat.health  # (nothing in this step raises it; StepGraphs.precapture's warm-ups do)
        self.optim.enable_p2p()                        # ERC_DP_P2P=1 under torch.distributed: exchange fused into the optimizer
        import os
        if os.environ.get("ERC_FUSE_ADAM", "1") != "0" and self.model.enc_train is None:
            self.model.fused_optim = self.optim        # single rank, bf16 mode: the optimizer rides in the weight-gradient launch
        if (self.model.compute == "bf16" or self.model.terms > 1) and self.model.enc_train is None:
            self.model.attach_bf16_shadow(self.optim)
        self.class_weight = None
        # faithful-cost mode: also run the reference's dead Transformer encoder on the padded
        # [B, T, D] block and discard the result, so that a step does the same arithmetic as the reference's
        self.encoder = None
        if params.get("faithful_dead_encoder", False):
            from .encoder import EncoderBlock
            self.encoder = EncoderBlock(self.model.rnn[0], self.device)

    def to_logits(self, batch):
        if self.encoder is not None:
            self.encoder.forward(batch["input_tensor"])
        return self.model(**batch)[0]

    def prepare_batch(self, batch):
        out = {k: (v.to(self.device) if torch.is_tensor(v) else v) for k, v in batch.items()}
        tl = batch.get("text_length")
        if "n_nodes" not in out and torch.is_tensor(tl) and not tl.is_cuda:
            out["n_nodes"] = int(tl.sum())      # host tensor: no device sync when a batch carries no labels
        if self.model.compute == "bf16":
            out["input_tensor"] = out["input_tensor"].to(torch.bfloat16)
        return out

    N_BUCKET = 256     # capacity buckets: node counts rounded up to a multiple of this

    def _bucket(self, like, B_cap, T_cap, N_cap):
        x, dev, D = like["input_tensor"], self.device, like["input_tensor"].shape[2]

        def make():
            return dict(input_tensor=torch.zeros(B_cap, T_cap, D, dtype=x.dtype, device=dev),
                        speaker_tensor=torch.zeros(B_cap, T_cap, dtype=like["speaker_tensor"].dtype, device=dev),
                        text_length=torch.zeros(B_cap, dtype=like["text_length"].dtype, device=dev),
                        label=torch.zeros(N_cap, dtype=like["label"].dtype, device=dev))

        def fill(static, b):
            Bb, Tb = b["input_tensor"].shape[:2]
            static["input_tensor"][:Bb, :Tb].copy_(b["input_tensor"], non_blocking=True)
            static["speaker_tensor"][:Bb, :Tb].copy_(b["speaker_tensor"], non_blocking=True)
            static["text_length"].zero_()                     # dialogues the batch does not have: length 0
            static["text_length"][:Bb].copy_(b["text_length"], non_blocking=True)
            static["label"][:b["label"].shape[0]].copy_(b["label"], non_blocking=True)

        return ("capacity", B_cap, T_cap, N_cap), make, fill

    def _caps(self, batch):
        B, T, D = batch["input_tensor"].shape
        B_cap = max(B, int(self.params.train.batch_size))
        return B_cap, max(T, int(getattr(self, "t_cap", 0))), D

    def capacity_bucket(self, batch):
        """trainer.StepGraphs: (key, make_static, fill) of the capacity bucket that holds ``batch`` (a prepared device batch),
        or None when the step cannot run in capacity mode (fp32 parity path, chained / faithful-cost encoder modes)."""
        if self.encoder is not None or not self.model.supports_capacity(batch):
            return None
        B_cap, T_cap, D = self._caps(batch)
        N = int(batch["label"].shape[0])
        N_cap = min(-(-N // self.N_BUCKET) * self.N_BUCKET, B_cap * T_cap)
        if N_cap > self.model.BN_FUSED_MAX_N or not capi.cogmen_project_graph_ok(D, F_HID, B_cap, D, D):
            return None
        return self._bucket(batch, B_cap, T_cap, N_cap)

    def resident_batch(self, store, cur_desc, B_cap, T_cap, N_cap):
        """trainer.ResidentEpochs: the "batch" of a step whose dialogues stay in the HBM-resident store -- the store's arrays,
        the 2 B_cap int32 the host rewrites per step (lengths | first store rows of the batch's dialogue slots) and the
        capacities the launches are sized for.  None when the step cannot run that way."""
        probe = dict(input_tensor=store.fused[None, :1], speaker_tensor=store.speaker[None, :1])
        D = int(store.fused.shape[1])
        if self.encoder is not None or store.fused.dtype != (torch.float32 if self.model.terms > 1 else torch.bfloat16) or \
                not self.model.supports_capacity(
                dict(probe, input_tensor=store.fused.view(1, -1, D))) or N_cap > self.model.BN_FUSED_MAX_N or \
                not capi.cogmen_project_graph_ok(D, F_HID, B_cap, D, D):
            return None
        return dict(input_tensor=store.fused, speaker_tensor=store.speaker, text_length=None, label=store.label, desc=cur_desc,
                    caps=(B_cap, T_cap, N_cap))

    def all_capacity_buckets(self, batch):
        """Every bucket a batch of this loader can fall into, smallest first, each with a synthetic filler (all B_cap
        dialogues present, lengths adding up to the capacity): data parallel runs capture all of them up front, in the
        same order on every rank (a captured step holds a collective: ranks must not capture at different times)."""
        if self.capacity_bucket(batch) is None:
            return []
        B_cap, T_cap, D = self._caps(batch)
        out = []
        for N_cap in range(self.N_BUCKET, min(B_cap * T_cap, self.model.BN_FUSED_MAX_N) + 1, self.N_BUCKET):
            key, make, fill = self._bucket(batch, B_cap, T_cap, N_cap)

            def synth(static, n=N_cap):
                lens = torch.full((B_cap, ), n // B_cap, dtype=torch.int64)
                lens[:n - int(lens.sum())] += 1                        # lengths add up to n, each <= T_cap
                static["text_length"].copy_(lens.clamp_(max=T_cap))
            out.append((key, make, fill, synth))
        return out

    def train_step(self, batch):
        """forward + CE + backward + (DP all-reduce) + Adam.  Returns the device stats tensor."""
        self.model.train()
        if self.encoder is not None:
            self.encoder.forward(batch["input_tensor"])      # dead work, result discarded (cogmen.py:146-147)
        stats = self.model.loss_and_grads(batch, self.class_weight)
        if self.model._last_ws["planner"].adam_fused:      # the weight-gradient launch applied the update (no optimizer launch)
            return stats
        scale = all_reduce_grads(self.model.flat)
        self.optim.step(grad_scale=scale)
        if self.model.enc_train is not None:
            self.model.enc_train.refresh_shadows()
        return stats
