"""2-layer bidirectional LSTM (hidden 100 per direction) on libercgraft: the sequence-context encoder of
DialogueGCN (``SeqContext``, packed; track_mm/dgcn_models.py:10-33) and of MMGCN's text branch (unpacked over
the padded length; track_mm/mmgcn.py:69,113-114).  ``torch.nn.LSTM`` parameter names / shapes are kept (the
module that owns the parameters IS an ``nn.LSTM``, used as a parameter holder only); the math runs in
``erc_gemm_*`` (hoisted input projections, all weight gradients) and ``erc_lstm_scan_*`` (the recurrence).
"""
import torch

from . import capi
from .engine import linear_fwd, linear_wgrad

H = 100


def lstm_groups(prefix, m):
    """FlatParams groups for an nn.LSTM(num_layers=2, bidirectional=True): forward|reverse members adjacent."""
    out = []
    for k in (0, 1):
        for base in ("weight_ih_l%d", "bias_ih_l%d", "weight_hh_l%d", "bias_hh_l%d"):
            n = base % k
            out.append([(prefix + n, getattr(m, n)), (prefix + n + "_reverse", getattr(m, n + "_reverse"))])
    return out


class BiLSTM2:
    def __init__(self, flat, prefix, d_in, drop_p=0.4):
        self.flat, self.prefix, self.d_in, self.drop_p = flat, prefix, d_in, drop_p
        self._own = {}     # buffers of callers that pass no store (tests): one set, replaced when the row count changes

    def _w(self, name):
        return self.flat.w(self.prefix + name)

    def _off(self, name):
        return self.flat.offsets[self.prefix + name]

    def _buf(self, rows, device, store=None):
        """The recurrence's saved state for ``rows`` positions.  ``store`` is the calling module's per-shape workspace
        dict: the buffers live (and are evicted, and are kept alive by a captured HIP graph) with it -- with compact rows
        the row count changes almost every step of a shuffled epoch, and a cache of its own here would grow by ~25 KB per
        row and distinct count."""
        if store is None:
            store = self._own
            if store.get("lstm_rows") != rows:
                store.clear()
        key = "lstm:" + self.prefix
        ws = store.get(key)
        if ws is None or store.get("lstm_rows", rows) != rows:
            # zeros, not empty: rows of padded positions are never written but ARE read by the weight-gradient
            # GEMMs (times a zero gate gradient), so they must stay finite
            f32 = lambda *s: torch.zeros(*s, dtype=torch.float32, device=device)
            ws = dict(GX=[f32(rows, 8 * H), f32(rows, 8 * H)], gates=[f32(rows, 8 * H), f32(rows, 8 * H)],
                      Cst=[f32(rows, 2 * H), f32(rows, 2 * H)], Hprev=[f32(rows, 2 * H), f32(rows, 2 * H)],
                      H0=f32(rows, 2 * H), H0d=f32(rows, 2 * H), dGX=[f32(rows, 8 * H), f32(rows, 8 * H)], dH0d=f32(rows, 2 * H))
            store[key], store["lstm_rows"] = ws, rows
        return ws

    def forward(self, pl, x, ldx, rows, B, T, sb, st, lengths, training, rng, out, ldo, x_bf16=False, node_off=None,
                node_row=None, store=None):
        """x [*, d_in] -> out [rows, 200] (row pitch ldo).  Padded rows: row(b,t) = b*sb + t*st of x and of every buffer.
        Compact rows (``node_off`` [B+1] and ``node_row`` [rows] given, packed sequences only): every buffer holds the
        ``rows`` = sum(lengths) valid positions in dialogue order (row = node_off[b] + t), x is read through node_row --
        the input projections, the saved state and the weight gradients shrink from B*T to sum(lengths) rows and the
        caller needs no gather / scatter between the padded and the node layout (DialogueGCN)."""
        ws = self._buf(rows, out.device, store)
        p = self.drop_p if training else 0.0
        linear_fwd(pl, x, ldx, node_row, self._w("weight_ih_l0"), self._w("bias_ih_l0"), ws["GX"][0], 8 * H, rows, 8 * H,
                   self.d_in, x_bf16=x_bf16)
        capi.lstm_scan_fwd(ws["GX"][0], 8 * H, self._w("weight_hh_l0"), self._w("bias_hh_l0"), lengths, node_off, sb, st,
                           B, T, ws["H0"], 2 * H, ws["H0d"], 2 * H, p, rng, 0x5EED0, ws["gates"][0], ws["Cst"][0],
                           ws["Hprev"][0])
        linear_fwd(pl, ws["H0d"], 2 * H, None, self._w("weight_ih_l1"), self._w("bias_ih_l1"), ws["GX"][1], 8 * H, rows,
                   8 * H, 2 * H)
        capi.lstm_scan_fwd(ws["GX"][1], 8 * H, self._w("weight_hh_l1"), self._w("bias_hh_l1"), lengths, node_off, sb, st,
                           B, T, out, ldo, None, 0, 0.0, None, 0, ws["gates"][1], ws["Cst"][1], ws["Hprev"][1])
        self._last = (x, ldx, rows, B, T, sb, st, lengths, p, rng, x_bf16, node_off, node_row, store)

    def backward(self, pl, dout, lddo, dx=None, lddx=0):
        """dout = gradient wrt the layer-1 output.  Registers all weight-gradient jobs; optionally writes
        dx [rows, d_in] (needed when the LSTM input is itself trainable, MMGCN)."""
        x, ldx, rows, B, T, sb, st, lengths, p, rng, x_bf16, node_off, node_row, store = self._last
        ws = self._buf(rows, dout.device, store)
        # one gate-gradient buffer per layer: every weight gradient of the LSTM then joins the step's ONE batched
        # weight-gradient launch (erc_wgrad_table) instead of 6 split-K GEMMs + a slab reduce per layer pair
        for k in (1, 0):
            dGX = ws["dGX"][k]
            if k == 1:
                capi.lstm_scan_bwd(self._w("weight_hh_l1"), lengths, node_off, sb, st, B, T, ws["gates"][1], ws["Cst"][1],
                                   dout, lddo, 0.0, None, 0, dGX)
                xin, ldin, d_in, bf, gat = ws["H0d"], 2 * H, 2 * H, False, None
            else:
                capi.lstm_scan_bwd(self._w("weight_hh_l0"), lengths, node_off, sb, st, B, T, ws["gates"][0], ws["Cst"][0],
                                   ws["dH0d"], 2 * H, p, rng, 0x5EED0, dGX)
                xin, ldin, d_in, bf, gat = x, ldx, self.d_in, x_bf16, node_row
            # W_ih (both directions stacked [800, d_in]) and b_ih
            linear_wgrad(pl, dGX, 8 * H, xin, ldin, gat, 8 * H, d_in, rows, self._off("weight_ih_l%d" % k),
                         self._off("bias_ih_l%d" % k), x_bf16=bf, defer=True)
            # W_hh per direction: dGX[:, 400d:]^T Hprev[:, 100d:]; b_hh receives the same gradient as b_ih = the column
            # sums of the direction's gate gradients (the bias strip of this product)
            for d in (0, 1):
                linear_wgrad(pl, dGX[:, 4 * H * d:], 8 * H, ws["Hprev"][k][:, H * d:], 2 * H, None, 4 * H, H, rows,
                             self._off("weight_hh_l%d" % k) + d * 4 * H * H, self._off("bias_hh_l%d" % k) + d * 4 * H,
                             defer=True)
            if k == 1:   # gradient wrt the (dropped) layer-0 output
                capi.gemm_f32(dGX, 8 * H, 0, None, self._w("weight_ih_l1"), 2 * H, 1, None, ws["dH0d"], 2 * H,
                              rows, 2 * H, 8 * H)
            elif dx is not None:
                if node_row is not None:
                    raise capi.ErcGraftError("BiLSTM2.backward: dx with compact rows is not built")
                capi.gemm_f32(dGX, 8 * H, 0, None, self._w("weight_ih_l0"), self.d_in, 1, None, dx, lddx,
                              rows, self.d_in, 8 * H)
