"""MMGCN on the MI355X hot path (drop-in for track_mm/mmgcn.py:56-157).

``MMGCNModule`` keeps the reference's constructor signature, ``state_dict`` keys (the
constructed-but-unused ``att_model.*``, ``gatedatt.*`` and ``graph_model.{a_fc,...}`` parameters included) and
``forward(**batch) -> (logits [N,C], None)`` on the time-major batch layout (batch_first=False, one-hot speakers).

Chain: per-modality Linear(d_m,200) on the padded [T,B,.] blocks (+ unpacked BiLSTM on text, rnn.py) -> valid
rows, modality-major node order [a | v | l(+speaker embedding)] -> block-structured adjacency (cosine blocks by a
grouped MFMA GEMM, arccos similarity, cross-modal same-utterance entries, symmetric degree normalisation) ->
64 GCNII layers, each: grouped block product A*h + cross terms, [hi | h0] W_l as two accumulating GEMMs,
fused theta/alpha combine + ReLU + dropout -> regroup + dropout + ReLU -> Linear -> CE; hand-written backward
including the gradient THROUGH the adjacency into the features (the reference's adjacency is built with autograd
on, mmgcn_models.py:582-646).
"""
import math

import torch
from torch import nn

from . import capi
from .engine import WorkspaceCache, FlatParams, FusedAdam, GemmPlanner, SideStream, all_reduce_grads, linear_fwd, linear_wgrad, \
    matmul_wgrad_io
from .rnn import BiLSTM2, lstm_groups

FD, NLAYERS, LAMDA, ALPHA, DROP = 200, 64, 0.5, 0.1, 0.4
KSPLIT = 16      # parts of the layer-plane sums in the backward (adjacency / h0 gradients); 8 .. 32 measured alike
_KEY = {"a": "audio_feature", "v": "visual_feature", "t": "text_feature"}
_LIN = {"a": "linear_a", "v": "linear_v", "t": "linear_l"}


class _Holder(nn.Module):
    """Registers parameters by dotted name: sub-modules the reference constructs but never uses."""

    def __init__(self, table=()):
        super().__init__()
        for name, shape in table:
            self._add(name, shape)

    def _add(self, name, shape):
        head, _, rest = name.partition(".")
        if rest:
            if not hasattr(self, head):
                setattr(self, head, _Holder())
            getattr(self, head)._add(rest, shape)
        else:
            self.register_parameter(name, nn.Parameter(torch.zeros(*shape).uniform_(-0.05, 0.05)))


class _Conv(nn.Module):
    def __init__(self):
        super().__init__()
        self.weight = nn.Parameter(torch.empty(2 * FD, FD).uniform_(-1.0 / math.sqrt(FD), 1.0 / math.sqrt(FD)))


class _GraphNet(nn.Module):
    def __init__(self):
        super().__init__()
        self.convs = nn.ModuleList([_Conv() for _ in range(NLAYERS)])
        self.fcs = nn.ModuleList([nn.Linear(FD, FD)])


class _GraphModel(_Holder):
    def __init__(self, n_classes, n_speakers):
        super().__init__([("a_fc.weight", (FD, FD)), ("a_fc.bias", (FD,)), ("v_fc.weight", (FD, FD)), ("v_fc.bias", (FD,)),
                          ("l_fc.weight", (FD, FD)), ("l_fc.bias", (FD,)), ("feature_fc.weight", (FD, 6 * FD)),
                          ("feature_fc.bias", (FD,)), ("final_fc.weight", (n_classes, FD)), ("final_fc.bias", (n_classes,)),
                          ("modal_embeddings.weight", (3, FD)), ("a_spk_embs.weight", (n_speakers, FD)),
                          ("v_spk_embs.weight", (n_speakers, FD)), ("l_spk_embs.weight", (n_speakers, FD))])
        self.graph_net = _GraphNet()
        self.speaker_embeddings = nn.Embedding(n_speakers, FD)


def _unused_tables():
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


class MMGCNModule(nn.Module):
    X3_SPLIT = 10      # K splits of dh0 = DG U^T on erc_gemm_x3 (K = 12 800: 50 output tiles x 10 = 500 workgroups)

    def __init__(self, hidden_text=100, D_e=100, graph_hidden_size=200, n_speakers=2, max_seq_len=200, window_past=10,
                 window_future=10, n_classes=7, nodal_attention=True, hidden_visual=512, hidden_audio=100, modals="atv",
                 seed=1):
        super().__init__()
        if len(modals) < 2:
            raise NotImplementedError("MMGCN needs at least two modalities (mmgcn_models.py:594-595)")
        self.modals, self.n_speakers, self.n_classes = modals, n_speakers, n_classes
        self.dims = {"a": hidden_audio, "v": hidden_visual, "t": hidden_text}
        self.order = [m for m in "avt" if m in modals]            # [a, v, l] order of create_big_adj
        self.linear_l = nn.Linear(hidden_text, FD)
        self.lstm_l = nn.LSTM(FD, 100, 2, bidirectional=True, dropout=DROP)
        self.linear_a = nn.Linear(hidden_audio, FD)
        self.linear_v = nn.Linear(hidden_visual, FD)
        att, gated = _unused_tables()
        self.att_model = _Holder(att)
        self.graph_model = _GraphModel(n_classes, n_speakers)
        self.gatedatt = _Holder(gated)
        self.smax_fc = nn.Linear(2 * FD * len(modals), n_classes)
        self.drop_p = DROP
        self.flat, self._ws, self._seed = None, WorkspaceCache(), seed

    def live_groups(self):
        groups = []
        for m in self.order:
            lin = getattr(self, _LIN[m])
            groups += [[(_LIN[m] + ".weight", lin.weight)], [(_LIN[m] + ".bias", lin.bias)]]
        if "t" in self.order:
            groups += lstm_groups("lstm_l.", self.lstm_l)
            groups += [[("graph_model.speaker_embeddings.weight", self.graph_model.speaker_embeddings.weight)]]
        gn = self.graph_model.graph_net
        groups += [[("graph_model.graph_net.fcs.0.weight", gn.fcs[0].weight)],
                   [("graph_model.graph_net.fcs.0.bias", gn.fcs[0].bias)]]
        groups += [[("graph_model.graph_net.convs.%d.weight" % i, gn.convs[i].weight)] for i in range(NLAYERS)]
        groups += [[("smax_fc.weight", self.smax_fc.weight)], [("smax_fc.bias", self.smax_fc.bias)]]
        return groups

    def finalize(self, device):
        self.to(device)
        self.flat = FlatParams(self.live_groups(), device)
        self.lstm = BiLSTM2(self.flat, "lstm_l.", FD, drop_p=DROP) if "t" in self.order else None
        self.rng_state = torch.tensor([0, self._seed], dtype=torch.int64, device=device)
        self.side = SideStream()
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
        Mo, C = len(self.order), self.n_classes
        R3, TB, P = Mo * N, T * B, (T + 3) // 4 * 4
        ws = dict(P=P, node_off=i32(B + 1), node_row=i32(N), node_dlg=i32(N), node_spk=i32(N),
                  LIN={m: f32(TB, FD) for m in self.order}, LO=f32(TB, FD), X=f32(R3, FD), XD=f32(R3, FD),
                  XH=f32(R3, FD), INV=f32(R3), COS=f32(B * Mo, P, P), ADJ=f32(B * Mo, P, P), CR=f32(B, Mo * Mo, P),
                  CCOS=f32(B, Mo * Mo, P), DEG=f32(R3), DDEG=f32(R3), H0=f32(R3, FD), Gt=f32(R3, FD),
                  HI=f32(NLAYERS + 1, R3, 2 * FD), HD=f32(NLAYERS + 2, R3, FD), FE=f32(N, Mo * 2 * FD), logits=f32(N, C),
                  stats=torch.zeros(256, dtype=torch.float32, device=device), dlogits=f32(N, C), dFE=f32(N, Mo * 2 * FD), dXD=f32(R3, FD), DH=f32(R3, FD),
                  dG=f32(NLAYERS + 1, R3, FD), dHIa=f32(NLAYERS + 1, R3, FD), dH0=f32(R3, FD),
                  dADJs=f32(KSPLIT, B * Mo, P, P), dH0s=f32(KSPLIT + 1, R3, FD),
                  emb_ws=f32(capi.mm_emb_grad_ws_floats(self.n_speakers)), dADJ=f32(B * Mo, P, P), dCR=f32(B, Mo * Mo, P),
                  Gb=f32(B * Mo, P, P), GC=f32(B, Mo * Mo, P), dXH=f32(R3, FD), dX=f32(R3, FD),
                  dLIN={m: f32(TB, FD) for m in self.order}, dLL=f32(TB, FD))
        import os
        ws["chain"] = os.environ.get("ERC_MM_CHAIN", "1") != "0" and T <= 128 and NLAYERS == 64 and FD == 200
        if ws["chain"]:
            # K8 (csrc/gcnii_chain.hip): the 64 layers in one persistent launch per direction
            LDS = NLAYERS * FD
            ws["chain_cfg"] = capi.gcnii_chain_config(B, T, Mo, P)
            ws["VT"], ws["V"] = f32(NLAYERS + 1, FD, 208), f32(NLAYERS + 1, FD, 208)
            ws["U"], ws["Call"] = f32(FD, LDS), f32(R3, LDS)
            # the two 16 GFLOP products around the chain (Call = h0 U, dh0 = DG U^T) as three-term bf16 splits on the bf16 matrix
            # cores (csrc/gemm_x3.hip: fp32-class, 2.4 x the exact-fp32 tiles); ERC_MM_GEMM_X3=0 keeps erc_gemm_f32
            ws["gemm_x3"] = os.environ.get("ERC_MM_GEMM_X3", "1") != "0"
            if ws["gemm_x3"]:
                ws["UT"], ws["dH0s3"] = f32(LDS, FD), f32(self.X3_SPLIT * R3 * FD)
            ws["ZS"], ws["DGl"], ws["DZl"] = f32(R3, LDS), f32(R3, LDS), f32(R3, LDS)
            ws["ZX"], ws["DH1"] = f32(2, R3, FD), f32(R3, FD)
            ws["chain_state"] = i32(1 + B + B * Mo * ws["chain_cfg"][0])
        dmax = max(self.dims[m] for m in self.order)
        slab = 4 * TB * 800 + 8 * (800 * FD + 2 * 400 * 100 * 2) + 8 * FD * dmax * 3 + NLAYERS * 2 * 8 * FD * FD + \
            8 * FD * FD + 8 * self.n_classes * Mo * 2 * FD + 8 * R3 * FD + (1 << 21)
        ws["planner"] = GemmPlanner(device, slab, grad=self.flat.grad)
        ws["planner"].MAX_SPLIT = 8      # 128 weight-gradient GEMMs per step: keep their slab sets small
        # ERC_MM_X3=1: weight gradients as three-term bf16 splits (csrc/wgrad.hip MB == 2: fp32-class products on the bf16
        # matrix cores).  Measured: 3.456 -> 3.421 ms per step -- the operand split on the VALU (two truncations, two
        # subtractions, three packs per value) eats most of what the matrix cores give back.  On by default with the other
        # fp32-class products (erc_gemm_x3); ERC_MM_X3=0 ERC_MM_GEMM_X3=0 is the exact-fp32 step.
        ws["planner"].mma_bf16 = 2 if os.environ.get("ERC_MM_X3", "1") == "1" else 0
        ws["jobs"] = None
        return ws

    def check_cluster(self):
        """Raise if a GCNII chain kernel (csrc/gcnii_chain.hip) flagged a per-layer exchange wait that ran into its bound.
        The affected optimizer steps were skipped ON THE DEVICE, on every rank (FusedAdam.skip_flag = the health word the
        chain kernels raise, summed over the ranks by the gradient all-reduce); the trainer calls this once per epoch
        after the training loop and after the evaluation loop, as for DAG-ERC."""
        self.flat.check_health("MMGCN GCNII chain")

    def _shape(self, batch_feat, lens, label, n_nodes=None):
        T, B = batch_feat.shape[0], batch_feat.shape[1]
        N = int(label.shape[0]) if label is not None else (int(n_nodes) if n_nodes is not None else int(lens.sum().item()))
        return B, T, N

    @staticmethod
    def theta(l):
        return math.log(LAMDA / l + 1)

    # ---------------------------------------------------------------- forward
    def _forward_impl(self, feats, qmask, lens, B, T, N, training):
        fp = self.flat
        dev = lens.device
        ws = self._workspace(B, T, N, dev)
        pl = ws["planner"]
        pl.reset()
        Mo, C, TB, P = len(self.order), self.n_classes, T * B, ws["P"]
        R3 = Mo * N
        p = self.drop_p if training else 0.0
        rng = self.rng_state
        if qmask.stride(2) != 1:
            qmask = qmask.contiguous()
        capi.mm_meta(lens, qmask, qmask.stride(0), qmask.stride(1), qmask.shape[2], B, ws["node_off"], ws["node_row"],
                     ws["node_dlg"], ws["node_spk"])
        X = ws["X"]
        for mi, m in enumerate(self.order):
            x = feats[m].reshape(TB, self.dims[m])
            if m != "t":
                # audio / visual: Linear only -- computed for the N valid utterances straight into node order (the padded rows
                # and the gather behind them are not needed; the text branch keeps them for its unpacked LSTM)
                linear_fwd(pl, x, self.dims[m], ws["node_row"], fp.w(_LIN[m] + ".weight"), fp.w(_LIN[m] + ".bias"),
                           X[mi * N:], FD, N, FD, self.dims[m])
                continue
            linear_fwd(pl, x, self.dims[m], None, fp.w(_LIN[m] + ".weight"), fp.w(_LIN[m] + ".bias"), ws["LIN"][m], FD, TB,
                       FD, self.dims[m])
            src = ws["LIN"][m]
            emb = spk = None
            if m == "t":
                # unpacked BiLSTM over the padded [T,B,200] block (row(b,t) = t*B + b): mmgcn.py:113-114
                self.lstm.forward(pl, ws["LIN"][m], FD, TB, B, T, 1, B, None, training, rng, ws["LO"], FD, store=ws)
                src, emb, spk = ws["LO"], fp.w("graph_model.speaker_embeddings.weight"), ws["node_spk"]
            capi.mm_flatten(src, FD, ws["node_row"], emb, spk, N, X[mi * N:], FD)
        # adjacency (mmgcn_models.py:582-646)
        capi.mm_row_normalize(X, R3, ws["XH"], ws["INV"])
        capi.gemm_grouped(1, ws["XH"], FD, ws["XH"], FD, ws["COS"], P, FD, ws["node_off"], B, Mo, N, T, P)
        capi.mm_adj_finish(ws["COS"], ws["XH"], ws["node_off"], B, Mo, N, P, ws["ADJ"], ws["CR"], ws["CCOS"], ws["DEG"])
        # GCNII input layer (mmgcn_models.py:382-384)
        n_el = R3 * FD
        XD = ws["XD"] if p > 0 else X
        if p > 0:
            capi.dropout_fwd(X, n_el, p, rng, 1000, XD)
        gn = "graph_model.graph_net."
        HD, HI = ws["HD"], ws["HI"]
        # h0 = relu(fc0 x); the chain's first plane is dropout(h0): without dropout the product writes it in place
        H0 = ws["H0"] if p > 0 else HD[1]
        linear_fwd(pl, XD, FD, None, fp.w(gn + "fcs.0.weight"), fp.w(gn + "fcs.0.bias"), H0, FD, R3, FD, FD, act=1)
        if p > 0:
            capi.dropout_fwd(H0, n_el, p, rng, 1001, HD[1])
        ws["_H0"] = H0
        if ws["chain"]:
            # K8: V_l / U_l from the layer weights, c_l = h0 U_l for all 64 layers as ONE product, then the whole chain in one
            # persistent launch (adjacency rows resident in LDS; csrc/gcnii_chain.hip)
            Wn0 = gn + "convs.0.weight"
            w_stride = fp.offsets[gn + "convs.1.weight"] - fp.offsets[Wn0]
            LDS = NLAYERS * FD
            capi.gcnii_chain_prep(fp.w(Wn0), w_stride, LAMDA, ALPHA, ws["VT"], ws["V"], ws["U"], ws.get("UT"))
            if ws["gemm_x3"]:
                capi.gemm_x3(H0, FD, ws["UT"], FD, ws["Call"], LDS, R3, LDS, FD)
            else:
                capi.gemm_f32(H0, FD, 0, None, ws["U"], LDS, 1, None, ws["Call"], LDS, R3, LDS, FD)
            capi.gcnii_chain_fwd(ws["ADJ"], P, ws["CR"], ws["node_off"], N, Mo, B, T, ws["chain_cfg"], ws["VT"], ws["Call"], LDS,
                                 HD, R3 * FD, ws["ZS"], LDS, ws["ZX"], ws["chain_state"], p, rng, 2000, health=fp.health)
        else:
            # every layer's input rows are [hi_l | h0] (pitch 2 FD): h0 is copied next to the 64 hi slots once per step, so that
            # [hi | h0] W is ONE product per layer, with the GCNII tail (residual mix, relu, dropout) in its epilogue
            HI[1:, :, FD:] = H0
            for l in range(1, NLAYERS + 1):
                W = fp.w(gn + "convs.%d.weight" % (l - 1))
                capi.gemm_grouped(0, ws["ADJ"], P, HD[l], FD, HI[l], 2 * FD, FD, ws["node_off"], B, Mo, N, T, P, cross=ws["CR"])
                capi.gcnii_layer_fwd(HI[l], 2 * FD, W, FD, self.theta(l), ALPHA, p, rng, 2000 + l, HD[l + 1], FD, R3, FD)
        capi.mm_regroup_fwd(XD, HD[NLAYERS + 1], Mo, N, p, rng, 3000, ws["FE"])
        linear_fwd(pl, ws["FE"], Mo * 2 * FD, None, fp.w("smax_fc.weight"), fp.w("smax_fc.bias"), ws["logits"], C, N, C,
                   Mo * 2 * FD)
        ws["_p"], ws["_XD"] = p, XD
        return ws

    def _feats(self, kw):
        return {m: kw[_KEY[m]] for m in self.order}

    def forward(self, text_feature=None, audio_feature=None, visual_feature=None, speaker_tensor=None,
                text_length=None, label=None, **kwargs):
        if self.flat is None:
            raise capi.ErcGraftError("call MMGCNModule.finalize(device) before forward")
        feats = self._feats(dict(text_feature=text_feature, audio_feature=audio_feature, visual_feature=visual_feature))
        B, T, N = self._shape(feats[self.order[0]], text_length, label, kwargs.get("n_nodes"))
        ws = self._forward_impl(feats, speaker_tensor, text_length, B, T, N, self.training)
        return ws["logits"], None

    def _legacy_chain_backward(self, ws, pl, DH, B, T, N, p, ks):
        """Round-1 form of the chain's backward (ERC_MM_CHAIN=0): launches per layer."""
        fp, off = self.flat, self.flat.offsets
        Mo, P = len(self.order), ws["P"]
        R3, n_el = Mo * N, Mo * N * FD
        HD, HI = ws["HD"], ws["HI"]
        gn = "graph_model.graph_net."
        # Per layer only what the NEXT layer's gradient needs stays on the dependency chain: dG_l, dhi_l = its residual
        # part + dG_l W_l[:FD]^T, and DH = A^T dhi_l.  Everything that only meets in a sum over the layers -- the
        # adjacency gradient sum_l dhi_l h_l^T, its cross-modal entries, the h0 gradient sum_l dG_l W_l[FD:]^T and the
        # weight gradients -- is computed after the loop, one launch each over all 64 layer planes.
        dHIa, dH0 = ws["dHIa"], ws["dH0"]
        dH0e = ws["dH0s"][KSPLIT]          # elementwise residual contributions to dh0: the last slab of the dh0 sum
        dH0e.zero_(), ws["dCR"].zero_()
        for l in range(NLAYERS, 0, -1):
            Wn = gn + "convs.%d.weight" % (l - 1)
            W = fp.w(Wn)
            dG, dhi = ws["dG"][l], dHIa[l]
            capi.gcnii_combine_bwd(DH, HD[l + 1], n_el, self.theta(l), ALPHA, ks, 0, dG, dhi, dH0e, F=FD, ld_d=FD)
            capi.gemm_f32(dG, FD, 0, None, W, FD, 0, None, dhi, FD, R3, FD, FD, accumulate=1)          # dhi += dG W[:FD]^T
            matmul_wgrad_io(pl, HI[l], 2 * FD, dG, FD, 2 * FD, FD, R3, off[Wn], None, defer=True)        # dW = [hi|h0]^T dG
            capi.gemm_grouped(0, ws["ADJ"], P, dhi, FD, DH, FD, FD, ws["node_off"], B, Mo, N, T, P, cross=ws["CR"])
        # the sums over the layer planes, cut into KSPLIT parts each (slabs, reduced in order)
        plane, n_adj = R3 * FD, B * Mo * P * P
        capi.gemm_grouped(1, dHIa[1], FD, HD[1], FD, ws["dADJs"], P, FD, ws["node_off"], B, Mo, N, T, P, planes=NLAYERS,
                          a_plane=plane, b_plane=plane, split=KSPLIT, c_slab=n_adj)
        capi.slab_reduce(ws["dADJs"], KSPLIT, n_adj, None, P, 0, ws["dADJ"], n_adj)
        capi.mm_cross_grad(dHIa[1], FD, HD[1], FD, ws["node_dlg"], ws["node_off"], Mo, N, P, ws["dCR"], planes=NLAYERS,
                           d_plane=plane, h_plane=plane)
        w_stride = off[gn + "convs.1.weight"] - off[gn + "convs.0.weight"]
        assert all(off[gn + "convs.%d.weight" % i] == off[gn + "convs.0.weight"] + i * w_stride for i in range(NLAYERS))
        W_bot = fp.data[off[gn + "convs.0.weight"] + FD * FD:]                                        # rows FD.. of layer 1's weight
        capi.gemm_f32_planes(ws["dG"][1], FD, plane, W_bot, FD, w_stride, ws["dH0s"], FD, R3, FD, FD, NLAYERS,
                             split_k=KSPLIT, c_slab=n_el)
        capi.slab_reduce(ws["dH0s"], KSPLIT + 1, n_el, None, FD, 0, dH0, n_el)
        return DH, dH0

    # --------------------------------------------------------------- training
    def loss_and_grads(self, batch):
        feats = self._feats(batch)
        qmask, lens, ys = batch["speaker_tensor"], batch["text_length"], batch["label"]
        B, T, N = self._shape(feats[self.order[0]], lens, ys)
        self.flat.roll_health()      # a timeout of the previous step: counted, cleared -- this step runs normally
        ws = self._forward_impl(feats, qmask, lens, B, T, N, self.training)
        fp, pl, off = self.flat, ws["planner"], self.flat.offsets
        Mo, C, TB, P = len(self.order), self.n_classes, T * B, ws["P"]
        R3, n_el = Mo * N, Mo * N * FD
        p, XD = ws["_p"], ws["_XD"]
        ks = 1.0 / (1.0 - p)
        HD, HI = ws["HD"], ws["HI"]
        gn = "graph_model.graph_net."
        capi.cross_entropy(ws["logits"], C, C, N, None, ys, None, 1.0, ws["dlogits"], C, ws["stats"])
        FW = Mo * 2 * FD
        capi.gemm_f32(ws["dlogits"], C, 0, None, fp.w("smax_fc.weight"), FW, 1, None, ws["dFE"], FW, N, FW, C)
        linear_wgrad(pl, ws["dlogits"], C, ws["FE"], FW, None, C, FW, N, off["smax_fc.weight"], off["smax_fc.bias"],
                     defer=True)
        DH = ws["DH"]
        capi.mm_regroup_bwd(ws["dFE"], ws["FE"], Mo, N, ks, ws["dXD"], DH)
        if ws["chain"]:
            # K8 backward: one persistent launch leaves dg_l (= d out_l) and dz_l (= A dg_l) of every layer and the gradient wrt
            # the chain's input; what only meets in sums over the layers follows as batched products
            LDS = NLAYERS * FD
            plane, n_adj = R3 * FD, B * Mo * P * P
            dH0 = ws["dH0"]
            ws["dCR"].zero_()
            capi.gcnii_chain_bwd(ws["ADJ"], P, ws["CR"], ws["node_off"], N, Mo, B, T, ws["chain_cfg"], ws["V"], HD, plane, DH,
                                 ws["DH1"], ws["DGl"], ws["DZl"], LDS, ws["ZX"], ws["chain_state"], p, health=fp.health)
            for l in range(1, NLAYERS + 1):
                Wn = gn + "convs.%d.weight" % (l - 1)
                th = self.theta(l)
                # dW_l[:200] = theta_l h_l^T dz_l ; dW_l[200:] = theta_l h0^T dg_l  (V_l, U_l are theta_l W + multiples of I)
                matmul_wgrad_io(pl, HD[l], FD, ws["DZl"][:, (l - 1) * FD:], LDS, FD, FD, R3, off[Wn], None, defer=True, scale=th)
                matmul_wgrad_io(pl, ws["_H0"], FD, ws["DGl"][:, (l - 1) * FD:], LDS, FD, FD, R3, off[Wn] + FD * FD, None,
                                defer=True, scale=th)
            if self.side.enabled:
                # nothing but the optimizer waits for the 128 weight gradients of the chain: second stream (ERC_SIDE_STREAM=1),
                # next to the rest of the backward, whose BiLSTM scans occupy 2 B of the 256 CUs
                with self.side.fork():
                    pl.flush_wgrads(ws, tag="_early")
            # dA = sum_l dg_l z_l^T on the block structure (and its cross-modal entries)
            if ws["gemm_x3"]:     # the planes of a row are contiguous: one K = 64 * 200 contraction per block
                capi.gemm_x3_grouped(ws["DGl"], LDS, ws["ZS"], LDS, ws["dADJs"], P, ws["node_off"], B, Mo, N, T, LDS,
                                     split_k=KSPLIT, c_slab=n_adj)
            else:
                capi.gemm_grouped(1, ws["DGl"], LDS, ws["ZS"], LDS, ws["dADJs"], P, FD, ws["node_off"], B, Mo, N, T, P, planes=NLAYERS,
                                  a_plane=FD, b_plane=FD, split=KSPLIT, c_slab=n_adj)
            capi.slab_reduce(ws["dADJs"], KSPLIT, n_adj, None, P, 0, ws["dADJ"], n_adj)
            capi.mm_cross_grad(ws["DGl"], LDS, ws["ZS"], LDS, ws["node_dlg"], ws["node_off"], Mo, N, P, ws["dCR"], planes=NLAYERS,
                               d_plane=FD, h_plane=FD)
            # dh0 = sum_l dg_l U_l^T = DG U^T: one product with K = 64 * 200
            if ws["gemm_x3"]:
                capi.gemm_x3(ws["DGl"], LDS, ws["U"], LDS, ws["dH0s3"], FD, R3, FD, LDS, split_k=self.X3_SPLIT, c_slab=R3 * FD)
                capi.slab_reduce(ws["dH0s3"], self.X3_SPLIT, R3 * FD, None, FD, 0, dH0, R3 * FD)
            else:
                linear_fwd(pl, ws["DGl"], LDS, None, ws["U"], None, dH0, FD, R3, FD, LDS)
            DH = ws["DH1"]
        else:
            DH, dH0 = self._legacy_chain_backward(ws, pl, DH, B, T, N, p, ks)
        # input layer: HD[1] = dropout(H0), H0 = relu(fc0(XD))
        capi.axpy_mask(DH, HD[1] if p > 0 else None, n_el, ks, 1, dH0)
        dG0 = ws["dG"][0]
        capi.gcnii_combine_bwd(dH0, ws["_H0"], n_el, 0.0, 0.0, 1.0, 1, dG0, None, None)
        capi.gemm_f32(dG0, FD, 0, None, fp.w(gn + "fcs.0.weight"), FD, 1, None, ws["dXD"], FD, R3, FD, FD, accumulate=1)
        linear_wgrad(pl, dG0, FD, XD, FD, None, FD, FD, R3, off[gn + "fcs.0.weight"], off[gn + "fcs.0.bias"], defer=True)
        dX = ws["dX"]
        capi.axpy_mask(ws["dXD"], XD if p > 0 else None, n_el, ks, 0, dX)
        # through the adjacency into the features
        capi.mm_adj_finish_bwd(ws["COS"], ws["CCOS"], ws["DEG"], ws["dADJ"], ws["dCR"], ws["node_off"], B, Mo, N, P,
                               ws["Gb"], ws["GC"], ws["DDEG"])
        capi.gemm_grouped(0, ws["Gb"], P, ws["XH"], FD, ws["dXH"], FD, FD, ws["node_off"], B, Mo, N, T, P)
        capi.mm_cross_apply(ws["GC"], ws["XH"], FD, ws["node_dlg"], ws["node_off"], Mo, N, P, ws["dXH"], FD)
        capi.mm_row_normalize_bwd(ws["XH"], ws["INV"], ws["dXH"], R3, dX)
        # per modality: back to the padded [T,B] rows, (speaker embedding, BiLSTM,) Linear
        for mi, m in enumerate(self.order):
            dm = dX[mi * N:]
            x = feats[m].reshape(TB, self.dims[m])
            if m != "t":
                linear_wgrad(pl, dm, FD, x, self.dims[m], ws["node_row"], FD, self.dims[m], N, off[_LIN[m] + ".weight"],
                             off[_LIN[m] + ".bias"])
                continue
            dpad = ws["dLIN"][m]
            dpad.zero_()
            capi.gather_rows(dm, FD, ws["node_row"], N, FD, dpad, FD, scatter=1)
            dlin = dpad
            if m == "t":
                capi.mm_emb_grad(dm, FD, ws["node_spk"], N, self.n_speakers, fp.g("graph_model.speaker_embeddings.weight"),
                                 ws["emb_ws"])
                self.lstm.backward(pl, dpad, FD, dx=ws["dLL"], lddx=FD)
                dlin = ws["dLL"]
            x = feats[m].reshape(TB, self.dims[m])
            linear_wgrad(pl, dlin, FD, x, self.dims[m], None, FD, self.dims[m], TB, off[_LIN[m] + ".weight"],
                         off[_LIN[m] + ".bias"])
        pl.reduce_into(ws, fp.grad)
        self.side.join()
        return ws["stats"]


class MMGCNTrainer:
    """train_step / to_logits of track_mm/mmgcn.py:126-157 (CE, Adam lr 3e-4 wd 3e-5)."""

    def __init__(self, params, device):
        self.params, self.device = params, torch.device(device)
        torch.manual_seed(params.seed)
        self.model = MMGCNModule(hidden_text=params.hidden_text, hidden_visual=params.hidden_visual,
                                 hidden_audio=params.hidden_audio, n_speakers=params.n_speakers,
                                 n_classes=params.n_classes, modals=params.modality, seed=params.seed).finalize(self.device)
        o = params.optim
        self.optim = FusedAdam(self.model.flat, lr=o.lr, weight_decay=o.get("weight_decay", 0.0),
                               decoupled=(o.name == "AdamW"), seed=params.seed)
        self.model.rng_state = self.optim.rng_state
        self.optim.skip_flag = self.model.flat.health    # a chain exchange timed out (on any rank) -> the update is skipped

    def to_logits(self, batch):
        return self.model(**batch)[0]

    def prepare_batch(self, batch):
        out = {k: (v.to(self.device) if torch.is_tensor(v) else v) for k, v in batch.items()}
        tl = batch.get("text_length")
        if "n_nodes" not in out and torch.is_tensor(tl) and not tl.is_cuda:
            out["n_nodes"] = int(tl.sum())      # host tensor: no device sync when a batch carries no labels
        return out

    def train_step(self, batch):
        self.model.train()
        stats = self.model.loss_and_grads(batch)
        scale = all_reduce_grads(self.model.flat)
        self.optim.step(grad_scale=scale)
        return stats
