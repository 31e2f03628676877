"""Numerics study behind the split compute modes (DESIGN.md finding 51; test infrastructure: uses the oracle).

The COGMEN oracle with every dense product (forward, input gradient, weight gradient) replaced by an emulation of a split
scheme -- bf16 / fp16 terms, 2 or 3 of them, the term products of weight >= 2^(-8 (terms - 1)) accumulated in fp32 -- compared
with the float64 oracle and the fp32 oracle at a small shape and at the config-2 shape:

    python tests/study_split_numerics.py [big]
"""
import sys, math, torch
sys.path.insert(0, __import__('os').path.dirname(__import__('os').path.dirname(__import__('os').path.abspath(__file__))))
from torch.nn import functional as F
from oracle import pyg
from oracle.cogmen import COGMENOracle
from tests.util_cases import cogmen_case, rel_err
import oracle.cogmen as oc

SCHEME = ['f32']
GSCALE = [1.0]

def terms(x, dt, n):
    out = []
    r = x.clone()
    for i in range(n):
        t = r.to(dt).to(torch.float32)
        out.append(t)
        r = r - t
    return out

def mm(a, b, grad_a=False, grad_b=False):
    """a @ b emulating a split scheme; grad_* says which operand is a gradient (scaled for fp16)"""
    s = SCHEME[0]
    if s == 'f32':
        return a @ b
    if s == 'bf16':
        return a.bfloat16().float() @ b.bfloat16().float()
    dt = torch.bfloat16 if s.startswith('bf16') else torch.float16
    n = int(s[-1])
    sa = GSCALE[0] if (grad_a and dt == torch.float16) else 1.0
    sb = GSCALE[0] if (grad_b and dt == torch.float16) else 1.0
    A, B = terms(a * sa, dt, n), terms(b * sb, dt, n)
    acc = torch.zeros(a.shape[0], b.shape[1])
    # small terms first
    pairs = [(i, j) for i in range(n) for j in range(n) if i + j < n]
    pairs.sort(key=lambda p: -(p[0] + p[1]))
    for i, j in pairs:
        acc = acc + A[i] @ B[j]
    return acc / (sa * sb)

class Lin(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, W, b):
        ctx.save_for_backward(x, W)
        return mm(x.reshape(-1, x.shape[-1]), W.t()).reshape(*x.shape[:-1], -1) + b
    @staticmethod
    def backward(ctx, dy):
        x, W = ctx.saved_tensors
        dy2 = dy.reshape(-1, dy.shape[-1])
        return mm(dy2, W, grad_a=True).reshape(x.shape), mm(dy2.t(), x.reshape(-1, x.shape[-1]), grad_a=True), dy2.sum(0)

class RGCN(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, weight, root, bias, src, dst, typ):
        n, R, Fh = x.size(0), weight.size(0), x.size(1)
        M = torch.zeros(n, (R + 1) * Fh); inv = torch.zeros(n, R)
        for r in range(R):
            sel = typ == r
            if not bool(sel.any()): continue
            s, d = src[sel], dst[sel]
            cnt = pyg.scatter_sum(torch.ones(s.numel()), d, n)
            M[:, r*Fh:(r+1)*Fh] = pyg.scatter_sum(x[s], d, n) / cnt.clamp(min=1)[:, None]
            inv[:, r] = torch.where(cnt > 0, 1.0 / cnt.clamp(min=1), torch.zeros_like(cnt))
        M[:, R*Fh:] = x
        Wcat = torch.cat([weight.reshape(R*Fh, -1), root], 0)
        ctx.save_for_backward(M, Wcat, inv, src, dst, typ); ctx.dims = (n, R, Fh)
        return mm(M, Wcat) + bias
    @staticmethod
    def backward(ctx, dH1):
        M, Wcat, inv, src, dst, typ = ctx.saved_tensors
        n, R, Fh = ctx.dims
        dx = torch.zeros(n, Fh)
        for r in range(R):
            sel = typ == r
            if not bool(sel.any()): continue
            s, d = src[sel], dst[sel]
            dP = pyg.scatter_sum(dH1[d] * inv[d, r][:, None], s, n)
            dx = dx + mm(dP, Wcat[r*Fh:(r+1)*Fh].t(), grad_a=True)
        dx = dx + mm(dH1, Wcat[R*Fh:].t(), grad_a=True)
        dW = mm(M.t(), dH1, grad_b=True)
        return dx, dW[:R*Fh].reshape(R, Fh, -1), dW[R*Fh:], dH1.sum(0), None, None, None

def patch(model):
    conv1, conv2 = model.gcn.conv1, model.gcn.conv2
    conv1.forward = lambda x, ei, et: RGCN.apply(x, conv1.weight, conv1.root, conv1.bias, ei[0], ei[1], et)
    def c2(x, edge_index):
        n = x.size(0); src, dst = edge_index[0], edge_index[1]
        lin = lambda m, t: Lin.apply(t, m.weight, m.bias)
        q, k, v = lin(conv2.lin_query, x), lin(conv2.lin_key, x), lin(conv2.lin_value, x)
        score = (q[dst] * k[src]).sum(-1) / math.sqrt(conv2.out_channels)
        mx = torch.full((n,), -float("inf")).scatter_reduce(0, dst, score.detach(), reduce="amax", include_self=True)
        ex = torch.exp(score - mx[dst]); den = pyg.scatter_sum(ex, dst, n)
        alpha = ex / (den[dst] + 1e-16)
        return pyg.scatter_sum(alpha[:, None] * v[src], dst, n) + lin(conv2.lin_skip, x)
    conv2.forward = c2
    model._linear = lambda mod, t: Lin.apply(t, mod.weight, mod.bias)
    model.bf16_products = True   # routes rnn.1 / cls through _linear

def run(case, scheme, gscale=1.0):
    SCHEME[0] = scheme; GSCALE[0] = gscale
    torch.manual_seed(case["seed"])
    D, C, S = case["D"], case["n_classes"], case["n_speakers"]
    ref = COGMENOracle(D, 100, 17, S, C, dead_encoder=False)
    with torch.no_grad():
        ref.gcn.bn.weight.uniform_(0.5, 1.5); ref.gcn.bn.bias.uniform_(-0.3, 0.3)
        ref.gcn.conv1.bias.uniform_(-0.1, 0.1)
    mine = COGMENOracle(D, 100, 17, S, C, dead_encoder=False)
    mine.load_state_dict(ref.state_dict()); patch(mine)
    ref64 = COGMENOracle(D, 100, 17, S, C, dead_encoder=False).double(); ref64.load_state_dict(ref.state_dict())
    batch = case["batch"]
    b64 = dict(batch, input_tensor=batch["input_tensor"].double())
    res = {}
    for m in (ref, mine, ref64):
        m.train()
        for mm_ in m.modules():
            if isinstance(mm_, torch.nn.Dropout): mm_.p = 0.0
    outs = {}
    for name, m, b in (("ref", ref, batch), ("mine", mine, batch), ("r64", ref64, b64)):
        logits, _ = m(**b)
        loss = F.cross_entropy(logits, b["label"]); m.zero_grad(); loss.backward()
        outs[name] = (logits.detach(), {n: p.grad for n, p in m.named_parameters() if p.grad is not None})
    def cmp(a, b):
        le = float((outs[a][0].double() - outs[b][0].double()).abs().max())
        ge = 0; gn = 0
        for n, g in outs[b][1].items():
            if n in ("gcn.conv2.lin_key.bias","gcn.conv2.lin_value.bias","gcn.conv2.lin_skip.bias"): continue
            ga = outs[a][1][n]
            ge = max(ge, rel_err(ga, g)); gn = max(gn, float((ga.double()-g.double()).norm()/(g.double().norm()+1e-12)))
        return le, ge, gn
    return cmp("mine", "r64"), cmp("ref", "r64"), cmp("mine", "ref")

if __name__ == "__main__":
    big = len(sys.argv) > 1
    if big:
        case = cogmen_case(B=32, min_len=20, max_len=110, dims=dict(a=100, t=768, v=512), seed=5)
    else:
        case = cogmen_case(B=8, min_len=3, max_len=40, dims=dict(a=12, t=20, v=16), seed=3)
    N = case["batch"]["label"].shape[0]
    print("N", N)
    S = 2.0 ** math.ceil(math.log2(N))
    for sch, gs in (("f32",1),("bf16",1),("bf16x2",1),("bf16x3",1),("f16x2",1.0),("f16x2",S)):
        a, b, c = run(case, sch, gs)
        print("%-7s gs=%-6g vs fp64: logit %.2e grad %.2e norm %.2e | fp32ref vs fp64 %.2e %.2e | mine vs fp32ref: %.2e %.2e %.2e" % (sch, gs, *a, b[0], b[1], *c))
