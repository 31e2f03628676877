"""Oracle: restatement of the three torch_geometric operators on the path.

torch_geometric is a third-party dependency of the reference, NOT vendored,
NOT version-pinned (requirements.txt:12 is a bare ``torch_geometric``) and not
installed here.  These are restated from the published operator definitions
(PyG 2.x docs) and structurally cross-checked against the vendored 1.4.2
``MessagePassing.propagate`` in models/rgcn.py:188-221 (gather
x[edge_index[0]] -> message -> scatter at edge_index[1] -> update).
PARITY UNPINNED for these three operators: the reference holds no test or
golden vector at this boundary.

Call sites: track_mm/cogmen.py:65-66,71-72 (RGCNConv, TransformerConv);
track_mm/dgcn_models.py:42,46 (GraphConv).
"""
import math

import torch
from torch import nn


def rb(t):
    """round to bf16 and back: the operand rounding of the bf16 matrix-core products (COGMEN bf16 compute mode)"""
    return t.to(torch.bfloat16).to(torch.float32)


class RoundedLinear(torch.autograd.Function):
    """nn.Linear as the bf16 compute mode evaluates it (csrc/cogmen_fused.hip): y = rb(x) rb(W)^T + b with fp32
    accumulation; backward dx = rb(dy) rb(W) (bf16 product), dW = rb(dy)^T rb(x) (the batched weight-gradient launch
    runs on bf16 matrix cores in this mode from operands the backward kernels STORE as bf16), db = colsum(rb(dy)) in fp32
    (the bias strip of that launch sums the stored values).  Not part of the reference: it restates the SAME algorithm
    with the operand rounding of the mode, so that a parity test isolates implementation error from quantisation."""

    @staticmethod
    def forward(ctx, x, W, b):
        xr, Wr = rb(x), rb(W)
        ctx.save_for_backward(xr, Wr)
        return xr @ Wr.t() + b

    @staticmethod
    def backward(ctx, dy):
        xr, Wr = ctx.saved_tensors
        dyr = rb(dy)
        return dyr @ Wr, dyr.t() @ xr, dyr.sum(0)


class RoundedWgradLinear(torch.autograd.Function):
    """nn.Linear whose forward and input gradient stay fp32 (the input projection's product is exact on its already
    rounded operands; the classifier head runs on the fp32 matrix cores) but whose WEIGHT gradient is the bf16 product of
    the mode's batched weight-gradient launch (csrc/wgrad_bf16.hip): dW = rb(dy)^T rb(x), db = colsum(rb(dy))."""

    @staticmethod
    def forward(ctx, x, W, b):
        ctx.save_for_backward(x, W)
        return x @ W.t() + b

    @staticmethod
    def backward(ctx, dy):
        x, W = ctx.saved_tensors
        dyr = rb(dy).reshape(-1, dy.shape[-1])          # [B, T, .] inputs (the padded block in front of rnn.1): rows flattened
        return dy @ W, dyr.t() @ rb(x).reshape(-1, x.shape[-1]), dyr.sum(0)


class RGCNMeanRounded(torch.autograd.Function):
    """RGCNConvMean.forward with the bf16 mode's operand rounding: H1 = rb(M) rb([W_r; root]) + b, M = [mean_r x | x];
    backward dx = sum_r rb(dP_r) rb(W_r)^T with dP = the transposed means of dH1 (aggregate first, then one product per
    relation -- the order the fused backward kernel uses), d[W_r; root] = rb(M)^T rb(dH1)."""

    @staticmethod
    def forward(ctx, x, weight, root, bias, src, dst, typ):
        n, R, F = x.size(0), weight.size(0), x.size(1)
        M = torch.zeros(n, (R + 1) * F, dtype=x.dtype)
        inv = torch.zeros(n, R, dtype=x.dtype)
        for r in range(R):
            sel = typ == r
            if not bool(sel.any()):
                continue
            s, d = src[sel], dst[sel]
            cnt = scatter_sum(torch.ones(s.numel(), dtype=x.dtype), d, n)
            M[:, r * F:(r + 1) * F] = scatter_sum(x[s], d, n) / cnt.clamp(min=1)[:, None]
            inv[:, r] = torch.where(cnt > 0, 1.0 / cnt.clamp(min=1), torch.zeros_like(cnt))
        M[:, R * F:] = x
        Wcat = torch.cat([weight.reshape(R * F, -1), root], 0)
        Mr, Wr = rb(M), rb(Wcat)
        ctx.save_for_backward(Mr, Wr, inv, src, dst, typ)
        ctx.dims = (n, R, F)
        return Mr @ Wr + bias

    @staticmethod
    def backward(ctx, dH1):
        Mr, Wr, inv, src, dst, typ = ctx.saved_tensors
        n, R, F = ctx.dims
        dx = torch.zeros(n, F, dtype=dH1.dtype)
        for r in range(R):
            sel = typ == r
            if not bool(sel.any()):
                continue
            s, d = src[sel], dst[sel]
            dP = scatter_sum(dH1[d] * inv[d, r][:, None], s, n)
            dx = dx + rb(dP) @ Wr[r * F:(r + 1) * F].t()
        dx = dx + rb(dH1) @ Wr[R * F:].t()
        dW = Mr.t() @ rb(dH1)
        return dx, dW[:R * F].reshape(R, F, -1), dW[R * F:], rb(dH1).sum(0), None, None, None


def scatter_sum(src, index, n):
    out = torch.zeros((n,) + tuple(src.shape[1:]), dtype=src.dtype)
    return out.index_add(0, index, src)


class RGCNConvMean(nn.Module):
    """x'_i = sum_r mean_{j in N_r(i)} x_j @ W_r + x_i @ root + bias.

    PyG RGCNConv(in, out, num_relations) defaults: aggr='mean', no bases /
    blocks, root_weight, bias.  Parameters ``weight [R,in,out]``, ``root
    [in,out]``, ``bias [out]`` (Appendix A of SURVEY.md).  Edge types >=
    num_relations are ignored (PyG loops ``for i in range(num_relations)``).
    Init: glorot(weight), glorot(root), zeros(bias).
    """

    def __init__(self, in_channels, out_channels, num_relations):
        super().__init__()
        self.in_channels, self.out_channels = in_channels, out_channels
        self.num_relations = num_relations
        self.rounded = False      # bf16 compute mode's operand rounding (RGCNMeanRounded)
        self.weight = nn.Parameter(torch.empty(num_relations, in_channels, out_channels))
        self.root = nn.Parameter(torch.empty(in_channels, out_channels))
        self.bias = nn.Parameter(torch.zeros(out_channels))
        for w in (self.weight, self.root):
            a = math.sqrt(6.0 / (w.size(-2) + w.size(-1)))
            nn.init.uniform_(w, -a, a)

    def forward(self, x, edge_index, edge_type):
        n = x.size(0)
        if self.rounded:
            return RGCNMeanRounded.apply(x, self.weight, self.root, self.bias, edge_index[0], edge_index[1], edge_type)
        out = torch.zeros(n, self.out_channels, dtype=x.dtype)
        src, dst = edge_index[0], edge_index[1]
        for r in range(self.num_relations):  # per-relation loop, as PyG does
            sel = edge_type == r
            if not bool(sel.any()):
                continue
            s, d = src[sel], dst[sel]
            summed = scatter_sum(x[s], d, n)
            cnt = scatter_sum(torch.ones(s.numel(), dtype=x.dtype), d, n).clamp(min=1)
            out = out + (summed / cnt[:, None]) @ self.weight[r]
        return out + x @ self.root + self.bias


class TransformerConv1(nn.Module):
    """PyG TransformerConv(in, out, heads=1, concat=True, beta=False,
    dropout=0, edge_dim=None, root_weight=True):
    alpha_ij = softmax_{j in N(i)} (W_q x_i + b_q).(W_k x_j + b_k) / sqrt(out),
    out_i = sum_j alpha_ij (W_v x_j + b_v) + W_skip x_i + b_skip; the softmax
    groups by TARGET node edge_index[1]."""

    def __init__(self, in_channels, out_channels):
        super().__init__()
        self.out_channels = out_channels
        self.rounded = False      # bf16 compute mode's operand rounding (RoundedLinear)
        self.lin_key = nn.Linear(in_channels, out_channels)
        self.lin_query = nn.Linear(in_channels, out_channels)
        self.lin_value = nn.Linear(in_channels, out_channels)
        self.lin_skip = nn.Linear(in_channels, out_channels)

    def forward(self, x, edge_index):
        n = x.size(0)
        src, dst = edge_index[0], edge_index[1]
        lin = (lambda m, t: RoundedLinear.apply(t, m.weight, m.bias)) if self.rounded else (lambda m, t: m(t))
        q, k, v = lin(self.lin_query, x), lin(self.lin_key, x), lin(self.lin_value, x)
        score = (q[dst] * k[src]).sum(-1) / math.sqrt(self.out_channels)
        mx = torch.full((n,), -float("inf"), dtype=x.dtype).scatter_reduce(
            0, dst, score.detach(), reduce="amax", include_self=True)
        ex = torch.exp(score - mx[dst])
        den = scatter_sum(ex, dst, n)
        alpha = ex / (den[dst] + 1e-16)
        out = scatter_sum(alpha[:, None] * v[src], dst, n)
        return out + lin(self.lin_skip, x)


class GraphConvAdd(nn.Module):
    """PyG GraphConv(in, out, aggr='add'): x'_i = W_rel sum_{j->i} x_j + b +
    W_root x_i (lin_rel has the bias, lin_root has none)."""

    def __init__(self, in_channels, out_channels):
        super().__init__()
        self.lin_rel = nn.Linear(in_channels, out_channels, bias=True)
        self.lin_root = nn.Linear(in_channels, out_channels, bias=False)

    def forward(self, x, edge_index):
        agg = scatter_sum(x[edge_index[0]], edge_index[1], x.size(0))
        return self.lin_rel(agg) + self.lin_root(x)
