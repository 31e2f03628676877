"""DAG-ERC on the MI355X hot path (drop-in for track_mm/dagerc.py:73-237).

``DAGERCModule`` keeps the reference's constructor signature, ``state_dict``
keys (including the never-trained ``fcs.*`` and
``attentive_node_features.transform.*``) and the
``forward(**batch) -> (logits [B,T,C] padded, None)`` contract; the harness masks
with ``attention_mask`` exactly as for the reference (dagerc.py:225).

Per layer: one hoisted GEMM for the input-side gates of ``grus_c``, the
hidden-side gates of ``grus_p`` and the attention's query score over all B*T
rows (stacked weight [W_ih_c ; W_hh_p ; w_q]), then the weight-stationary
recurrence kernel (csrc/dag_rec.hip: groups of dialogues as the MFMA M
dimension, each workgroup keeps its slice of the recurrent weights in
registers for all T steps).  The five hidden states
H0..H4 are written straight into one [B*T, 1500] buffer, so the
``torch.cat`` of dagerc.py:190-192 never happens: the head's first Linear is
two GEMMs (hidden block, raw-feature block) summed by the slab reducer.
"""
import os

import torch
from torch import nn

from . import capi
from .engine import WorkspaceCache, FlatParams, FusedAdam, GemmPlanner, all_reduce_grads, linear_fwd, linear_wgrad

HID = 300
LDG = 6 * HID + 4      # row pitch of the hoisted gate block: 1800 gate columns + the query-score column (+ pad to 16 bytes)


class _Gather(nn.Module):
    """Parameter holder named like GAT_dialoggcn_v1 (dagerc_models.py:319-324)."""

    def __init__(self, hidden):
        super().__init__()
        self.linear = nn.Linear(hidden * 2, 1)
        self.Wr0 = nn.Linear(hidden, hidden, bias=False)
        self.Wr1 = nn.Linear(hidden, hidden, bias=False)


class _Attentive(nn.Module):
    def __init__(self, hidden):
        super().__init__()
        self.transform = nn.Linear(hidden, hidden)   # unused (nodal_att_type None), dagerc_models.py:441-442


class DAGERCModule(nn.Module):
    def __init__(self, emb_dim=100, dropout=0.2, n_classes=7, gnn_layers=4, compute="f32", seed=1):
        super().__init__()
        self.emb_dim, self.n_classes, self.gnn_layers, self.compute = emb_dim, n_classes, gnn_layers, compute
        self.drop_p = float(dropout)
        self.dropout = nn.Dropout(dropout)
        self.gather = nn.ModuleList([_Gather(HID) for _ in range(gnn_layers)])
        self.grus_c = nn.ModuleList([nn.GRUCell(HID, HID) for _ in range(gnn_layers)])
        self.grus_p = nn.ModuleList([nn.GRUCell(HID, HID) for _ in range(gnn_layers)])
        self.fcs = nn.ModuleList([nn.Linear(HID * 2, HID) for _ in range(gnn_layers)])   # never used
        self.fc1 = nn.Linear(emb_dim, HID)
        in_dim = HID * (gnn_layers + 1) + emb_dim
        self.in_dim = in_dim
        self.out_mlp = nn.Sequential(nn.Linear(in_dim, HID), nn.ReLU(), nn.Linear(HID, HID), nn.ReLU(),
                                     nn.Dropout(dropout), nn.Linear(HID, n_classes))
        self.attentive_node_features = _Attentive(in_dim)
        self.flat = None
        self._ws = WorkspaceCache()
        self._seed = seed

    # ------------------------------------------------------------------ setup
    def live_groups(self):
        groups = [[("fc1.weight", self.fc1.weight)], [("fc1.bias", self.fc1.bias)]]
        for l in range(self.gnn_layers):
            c, p, g = self.grus_c[l], self.grus_p[l], self.gather[l]
            groups += [
                # hoisted: ONE [1802, 300] operand -- rows 0..1799 the two gate matrices, row 1800 w_q, row 1801 w_k (the
                # [1, 600] gather.linear.weight is two 300-rows back to back); the bias group lines up with rows 0..1800
                [("grus_c.%d.weight_ih" % l, c.weight_ih), ("grus_p.%d.weight_hh" % l, p.weight_hh),
                 ("gather.%d.linear.weight" % l, g.linear.weight)],
                [("grus_c.%d.bias_ih" % l, c.bias_ih), ("grus_p.%d.bias_hh" % l, p.bias_hh),
                 ("gather.%d.linear.bias" % l, g.linear.bias)],
                [("grus_c.%d.weight_hh" % l, c.weight_hh), ("grus_p.%d.weight_ih" % l, p.weight_ih)],   # sequential
                [("grus_c.%d.bias_hh" % l, c.bias_hh), ("grus_p.%d.bias_ih" % l, p.bias_ih)],
                [("gather.%d.Wr0.weight" % l, g.Wr0.weight), ("gather.%d.Wr1.weight" % l, g.Wr1.weight)],
            ]
        m = self.out_mlp
        groups += [[("out_mlp.0.weight", m[0].weight)], [("out_mlp.0.bias", m[0].bias)],
                   [("out_mlp.2.weight", m[2].weight)], [("out_mlp.2.bias", m[2].bias)],
                   [("out_mlp.5.weight", m[5].weight)], [("out_mlp.5.bias", m[5].bias)]]
        return groups

    def finalize(self, device):
        self.to(device)
        self.flat = FlatParams(self.live_groups(), device)
        self.rng_state = torch.tensor([0, self._seed], dtype=torch.int64, device=device)
        # [0] error flag of the recurrence kernels (an exchange timed out: the optimizer skips the step on the device),
        # then one launch epoch per dialogue group; shared by every workspace so that the flag has ONE address
        self.rec_state = torch.zeros(1 + 4096, dtype=torch.int32, device=device)
        return self

    @property
    def _last_ws(self):
        """workspace of the most recent forward (tests / bench read results out of it)"""
        return self._ws.last

    def _workspace(self, B, T, N, device):
        return self._ws.get((B, T, N), lambda: self._make_workspace(B, T, N, device))

    def _make_workspace(self, B, T, N, device):
        BT, L, C, D = B * T, self.gnn_layers, self.n_classes, self.emb_dim
        f32 = lambda *s: torch.empty(*s, dtype=torch.float32, device=device)
        i32 = lambda *s: torch.zeros(*s, dtype=torch.int32, device=device)
        ws = dict(
            spk=i32(B, T), pred=i32(B, T), node_off=i32(B + 1), node_row=i32(max(N, 1)),
            Hall=f32(BT, HID * (L + 1)), dHall=f32(BT, HID * (L + 1)),
            GI=[f32(BT, LDG) for _ in range(L)], GH=[f32(BT, 6 * HID) for _ in range(L)],
            Mseq=[f32(BT, HID) for _ in range(L)], R=[f32(BT, 2 * HID) for _ in range(L)],
            ks=[f32(BT) for _ in range(L)], alpha=[torch.zeros(B, T, T, dtype=torch.float32, device=device) for _ in range(L)],
            Y1=f32(BT, HID), Y2=f32(BT, HID), logits=f32(BT, C), dlogits=f32(BT, C), dY2=f32(BT, HID),
            dY1=f32(BT, HID), DGI=[f32(BT, LDG) for _ in range(L)], DGH=[f32(BT, 6 * HID) for _ in range(L)],
            # dM | dks | attention-weighted sums per layer (kept until the batched weight-gradient launch at the end of the step)
            dM=[f32(BT, HID) for _ in range(L)], dks=[f32(BT) for _ in range(L)], A=[f32(BT, 2 * HID) for _ in range(L)],
            stats=torch.zeros(256, dtype=torch.float32, device=device),
        )
        # cfg = (elements per workgroup, dialogues per group, groups per launch, layers per launch) of the recurrence
        # kernels per direction, from the device's CU count and the occupancy query (csrc/dag_rec.hip);
        # ERC_DAG_EPC / ERC_DAG_DG / ERC_DAG_LPL force a forward configuration, ERC_DAG_BEPC / ERC_DAG_BDG a backward one
        import os
        if B > 4096:
            raise capi.ErcGraftError("DAG-ERC: more than 4096 dialogues per batch")
        env = lambda k: int(os.environ.get(k, 0))
        ws["cfg_f"] = capi.dag_rec_config(0, B, T, L, env("ERC_DAG_EPC"), env("ERC_DAG_DG"), env("ERC_DAG_LPL"))
        ws["cfg_b"] = capi.dag_rec_config(1, B, T, L, env("ERC_DAG_BEPC"), env("ERC_DAG_BDG"), env("ERC_DAG_BLPL"))
        ws["cfg"] = (tuple(ws["cfg_f"]), tuple(ws["cfg_b"]))
        i64 = lambda n: torch.zeros(n // 8 + 1, dtype=torch.int64, device=device)
        ws["scratch_f"] = i64(capi.dag_rec_scratch_bytes(0, B, T, ws["cfg_f"]))
        ws["scratch_b"] = i64(capi.dag_rec_scratch_bytes(1, B, T, ws["cfg_b"]))
        W5 = HID * (L + 1)
        lw = [self._layer_w(l) for l in range(L)]
        ws["tables"] = {k: capi.ptr_table([w[k] for w in lw]) for k in ("Wh", "bh", "W_hh_c", "b_hh_c", "W_ih_p", "b_ih_p", "Wr", "w_k")}
        ws["tables"].update(H1=capi.ptr_table([ws["Hall"][:, HID * (l + 1):] for l in range(L)]),
                            Hl=capi.ptr_table([ws["Hall"][:, HID * l:] for l in range(L)]),
                            **{k: capi.ptr_table(ws[k]) for k in ("GI", "Mseq", "GH", "R", "ks", "alpha", "DGI", "DGH", "dM", "dks")})
        slab = 8 * BT * HID + 10 * (HID * self.in_dim) + 6 * L * (6 * HID * HID + 2 * HID * HID) + (1 << 20)
        ws["planner"] = GemmPlanner(device, slab, grad=self.flat.grad)
        # the fp32 weight gradients as three-term bf16 splits (csrc/wgrad.hip MB == 2, fp32-class: 3.54 -> 3.50 ms per step);
        # ERC_DAG_X3=0: exact fp32 products
        ws["planner"].mma_bf16 = 2 if os.environ.get("ERC_DAG_X3", "1") == "1" else 0
        ws["jobs"] = None
        return ws

    def check_cluster(self):
        """Raise if a recurrence kernel flagged an exchange wait that ran into its bound (its workgroups were not all
        resident at once, e.g. another process holds CUs).  The affected optimizer steps were already skipped ON THE
        DEVICE, on every rank (FusedAdam.skip_flag = the health word in the tail of the flat gradient buffer, which the
        gradient all-reduce sums over the ranks); this host-side check (one device->host copy) reports how many -- the
        trainer calls it after the training loop and after the evaluation loop of every epoch."""
        self.flat.check_health("DAG-ERC recurrence")

    def _shape(self, input_tensor, text_length, label, n_nodes=None):
        B, T = input_tensor.shape[0], input_tensor.shape[1]
        N = int(label.shape[0]) if label is not None else (int(n_nodes) if n_nodes is not None else int(text_length.sum().item()))
        return B, T, N

    def _layer_w(self, l):
        fp = self.flat
        return dict(Wh=fp.w("grus_c.%d.weight_ih" % l), bh=fp.w("grus_c.%d.bias_ih" % l),
                    W_hh_c=fp.w("grus_c.%d.weight_hh" % l), b_hh_c=fp.w("grus_c.%d.bias_hh" % l),
                    W_ih_p=fp.w("grus_p.%d.weight_ih" % l), b_ih_p=fp.w("grus_p.%d.bias_ih" % l),
                    Wr=fp.w("gather.%d.Wr0.weight" % l), w_k=fp.w("gather.%d.linear.weight" % l).view(-1)[HID:])

    # ---------------------------------------------------------------- forward
    def _forward_impl(self, x, speaker_tensor, text_length, B, T, N, training):
        fp, dev = self.flat, x.device
        ws = self._workspace(B, T, N, dev)
        pl = ws["planner"]
        pl.reset()
        BT, L, C, D, W5 = B * T, self.gnn_layers, self.n_classes, self.emb_dim, HID * (self.gnn_layers + 1)
        x_bf16 = x.dtype == torch.bfloat16
        if speaker_tensor.dim() == 3:     # one-hot [B,T,S] (speaker_onehot=True, dagerc.py:41)
            if speaker_tensor.stride(2) != 1:
                speaker_tensor = speaker_tensor.contiguous()
            capi.dag_meta(speaker_tensor.float() if speaker_tensor.dtype != torch.float32 else speaker_tensor, None,
                          speaker_tensor.stride(0), speaker_tensor.stride(1), speaker_tensor.shape[2], text_length,
                          B, T, ws["spk"], ws["pred"], ws["node_off"], ws["node_row"])
        else:
            capi.dag_meta(None, speaker_tensor, speaker_tensor.stride(0), speaker_tensor.stride(1),
                          1 << 30, text_length, B, T, ws["spk"], ws["pred"], ws["node_off"], ws["node_row"])
        Hall = ws["Hall"]
        # H0 = relu(fc1(x)) over ALL B*T rows, padded ones included (dagerc.py:164)
        linear_fwd(pl, x, D, None, fp.w("fc1.weight"), fp.w("fc1.bias"), Hall, W5, BT, HID, D, act=1, x_bf16=x_bf16)
        # all layers in one pipelined launch (csrc/dag_rec.hip): the hoisted products (gates of both cells' hoisted sides,
        # query score) are computed -- and saved to GI -- by the recurrence's own workgroups
        capi.dag_rec_fwd(Hall, W5, L, ws["tables"], ws["pred"], ws["spk"], B, T, W5, LDG, ws["cfg_f"], self.rec_state,
                         ws["scratch_f"], health=fp.health)
        # head: Y1 = relu([Hall | x] W0^T + b0) as two GEMMs into one slab set
        W0 = fp.w("out_mlp.0.weight")
        Sa = pl.split_for(BT, HID, W5)
        Sx = pl.split_for(BT, HID, D, bk=64 if x_bf16 else None, min_chunks=4 if x_bf16 else None)
        src = pl.take((Sa + Sx) * BT * HID)
        capi.gemm_f32(Hall, W5, 0, None, W0, self.in_dim, 0, None, pl.ws[src:], HID, BT, HID, W5, split_k=Sa,
                      c_slab=BT * HID)
        xs = pl.ws[src + Sa * BT * HID:]
        if x_bf16:
            capi.gemm_bf16x(x, D, 0, None, W0[:, W5:], self.in_dim, 0, None, 1, xs, HID, BT, HID, D, split_k=Sx,
                            c_slab=BT * HID)
        else:
            capi.gemm_f32(x, D, 0, None, W0[:, W5:], self.in_dim, 0, None, xs, HID, BT, HID, D, split_k=Sx,
                          c_slab=BT * HID)
        capi.slab_reduce(pl.ws[src:], Sa + Sx, BT * HID, fp.w("out_mlp.0.bias"), HID, 1, ws["Y1"], BT * HID)
        p = self.drop_p if training else 0.0
        linear_fwd(pl, ws["Y1"], HID, None, fp.w("out_mlp.2.weight"), fp.w("out_mlp.2.bias"), ws["Y2"], HID, BT, HID,
                   HID, act=3 if p > 0 else 1, drop_p=p, rng=self.rng_state)
        linear_fwd(pl, ws["Y2"], HID, None, fp.w("out_mlp.5.weight"), fp.w("out_mlp.5.bias"), ws["logits"], C, BT, C,
                   HID)
        return ws

    def forward(self, input_tensor, text_length, speaker_tensor, label=None, **kwargs):
        if self.flat is None:
            raise capi.ErcGraftError("call DAGERCModule.finalize(device) before forward")
        B, T, N = self._shape(input_tensor, text_length, label, kwargs.get("n_nodes"))
        ws = self._forward_impl(input_tensor, speaker_tensor, text_length, B, T, N, self.training)
        return ws["logits"].view(B, T, self.n_classes), None

    # --------------------------------------------------------------- training
    def loss_and_grads(self, batch):
        x, spk, lens, ys = batch["input_tensor"], batch["speaker_tensor"], batch["text_length"], batch["label"]
        B, T, N = self._shape(x, lens, ys)
        training = self.training
        self.flat.roll_health()      # a timeout of the previous step: counted, cleared -- this step runs normally
        ws = self._forward_impl(x, spk, lens, B, T, N, training)
        fp, pl = self.flat, ws["planner"]
        BT, L, C, D, W5 = B * T, self.gnn_layers, self.n_classes, self.emb_dim, HID * (self.gnn_layers + 1)
        x_bf16 = x.dtype == torch.bfloat16
        off = fp.offsets
        # masked CE (dagerc.py:225-226): the mask is the valid-row map; padded rows get zero gradient
        ws["dlogits"].zero_()
        capi.cross_entropy(ws["logits"], C, C, N, ws["node_row"], ys, None, 1.0, ws["dlogits"], C, ws["stats"])
        p = self.drop_p if training else 0.0
        capi.gemm_f32(ws["dlogits"], C, 0, None, fp.w("out_mlp.5.weight"), HID, 1, None, ws["dY2"], HID, BT, HID, C,
                      act=2, aux=ws["Y2"], ldaux=HID, act_scale=1.0 / (1.0 - p))
        linear_wgrad(pl, ws["dlogits"], C, ws["Y2"], HID, None, C, HID, BT, off["out_mlp.5.weight"],
                     off["out_mlp.5.bias"], defer=True)
        capi.gemm_f32(ws["dY2"], HID, 0, None, fp.w("out_mlp.2.weight"), HID, 1, None, ws["dY1"], HID, BT, HID, HID,
                      act=2, aux=ws["Y1"], ldaux=HID, act_scale=1.0)
        linear_wgrad(pl, ws["dY2"], HID, ws["Y1"], HID, None, HID, HID, BT, off["out_mlp.2.weight"],
                     off["out_mlp.2.bias"], defer=True)
        # dHall = dY1 W0[:, :1500]; dW0 = dY1^T [Hall | x] as two column slices of one slab set
        W0 = fp.w("out_mlp.0.weight")
        capi.gemm_f32(ws["dY1"], HID, 0, None, W0, self.in_dim, 1, None, ws["dHall"], W5, BT, W5, HID)
        slab = linear_wgrad(pl, ws["dY1"], HID, ws["Hall"], W5, None, HID, W5, BT, off["out_mlp.0.weight"], None,
                            ld_w=self.in_dim, force_slab=x_bf16)
        linear_wgrad(pl, ws["dY1"], HID, x, D, None, HID, D, BT, None, off["out_mlp.0.bias"], x_bf16=x_bf16,
                     slab=slab, col_off=W5)
        # all layers in one pipelined launch, top layer first (csrc/dag_rec.hip): leaves the gate gradients, dM and dks of
        # every layer and the complete gradient wrt H_0 (through fc1's relu mask) in block 0 of dHall
        capi.dag_rec_bwd(L, ws["tables"], W5, LDG, ws["pred"], ws["spk"], B, T, ws["dHall"], W5, LDG, ws["cfg_b"],
                         self.rec_state, ws["scratch_b"], health=fp.health)
        for l in range(L):
            Hl, H1 = ws["Hall"][:, HID * l:], ws["Hall"][:, HID * (l + 1):]
            # d[W_ih_c ; W_hh_p] and their biases (1800 rows: 16-byte operand loads); the two halves of gather.linear:
            # dw_q = DGI[:, 1800]^T H_l (+ its bias), dw_k = dks^T H1 -- one-row products of the same batched launch
            linear_wgrad(pl, ws["DGI"][l], LDG, Hl, W5, None, 6 * HID, HID, BT, off["grus_c.%d.weight_ih" % l],
                         off["grus_c.%d.bias_ih" % l], defer=True)
            linear_wgrad(pl, ws["DGI"][l][:, 6 * HID:], LDG, Hl, W5, None, 1, HID, BT, off["gather.%d.linear.weight" % l],
                         off["gather.%d.linear.bias" % l], defer=True)
            linear_wgrad(pl, ws["dks"][l], 1, H1, W5, None, 1, HID, BT, off["gather.%d.linear.weight" % l] + HID, None,
                         defer=True)
            linear_wgrad(pl, ws["DGH"][l], 6 * HID, ws["Mseq"][l], HID, None, 6 * HID, HID, BT,
                         off["grus_c.%d.weight_hh" % l], off["grus_c.%d.bias_hh" % l], defer=True)
            # d[Wr0 ; Wr1] = sum_j dR_j h_j^T = sum_i dM_i A_i^T with the attention-weighted sums A (a forward quantity)
            capi.dag_attn_sums(ws["alpha"][l], H1, W5, ws["pred"], ws["spk"], B, T, ws["A"][l])
            linear_wgrad(pl, ws["dM"][l], HID, ws["A"][l], 2 * HID, None, HID, HID, BT, off["gather.%d.Wr0.weight" % l], None,
                         defer=True)
            linear_wgrad(pl, ws["dM"][l], HID, ws["A"][l][:, HID:], 2 * HID, None, HID, HID, BT,
                         off["gather.%d.Wr1.weight" % l], None, defer=True)
        linear_wgrad(pl, ws["dHall"], W5, x, D, None, HID, D, BT, off["fc1.weight"], off["fc1.bias"], x_bf16=x_bf16)
        pl.reduce_into(ws, fp.grad)
        return ws["stats"]


class DAGERCTrainer:
    """train_step / to_logits of track_mm/dagerc.py:201-237 (masked CE, clip_grad_norm_ 5, AdamW)."""

    def __init__(self, params, device):
        self.params, self.device = params, torch.device(device)
        torch.manual_seed(params.seed)
        self.model = DAGERCModule(emb_dim=params.hidden_all, dropout=params.get("dropout", 0),
                                  n_classes=params.n_classes, gnn_layers=params.get("gnn_layers", 4),
                                  compute=params.get("compute", "f32"), seed=params.seed).finalize(self.device)
        o = params.optim
        self.optim = FusedAdam(self.model.flat, lr=o.lr, weight_decay=o.get("weight_decay", 1e-2),
                               decoupled=(o.name == "AdamW"), clip_norm=5.0, seed=params.seed)
        self.model.rng_state = self.optim.rng_state
        self.optim.skip_flag = self.model.flat.health    # a recurrence exchange timed out (on any rank) -> the update is skipped

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
        stats = self.model.loss_and_grads(batch)
        scale = all_reduce_grads(self.model.flat)
        self.optim.step(grad_scale=scale)
        return stats
