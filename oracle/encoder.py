"""Oracle (test infrastructure only): functional restatement of the Transformer encoder layer the reference instantiates
for COGMEN (track_mm/cogmen.py:94-102; layer math contrib/nn.py:283-305, a vendored copy of torch.nn's: post-norm, ReLU,
batch_first, dropout on the attention probabilities / after the attention block / inside and after the feed-forward
block) with two hooks the module form does not offer:

  * ``keeps``  explicit dropout decisions per (layer, site) -- site 0 attention probabilities [B, h, T, T], 1 after the
               attention block [B*T, D], 2 inside the FFN [B*T, ffn], 3 after the FFN [B*T, D]; values 0 or 1 / (1 - p) --
               so that a checker can apply the masks of the library's counter-based generator;
  * ``rnd``    a rounding hook applied where the HIP path stores bf16 (GEMM operands: activations and weights, the
               dropped probabilities, the attention output), straight-through for autograd.  With it the forward of the
               checker and of the HIP path agree to accumulation order, and a gradient comparison isolates the backward.

Pinned by tests/test_oracle_encoder.py against torch.nn.TransformerEncoder (eval mode, with and without the key-padding
mask).  The chained COGMEN variant that uses it is NOT what the reference computes: "parity unpinned".
"""
import math

import torch
import torch.nn.functional as F


def identity(t):
    return t


def round_bf16(t):
    """bf16 rounding with a straight-through gradient."""
    return t + (t.detach().to(torch.bfloat16).float() - t.detach())


def encoder_layer(x, lyr, pad_mask=None, keeps=None, layer=0, rnd=identity):
    """x [B, T, D] fp32; lyr: a torch.nn.TransformerEncoderLayer (only its parameters are read); pad_mask [B, T] bool,
    True = padded key.  Returns [B, T, D]."""
    B, T, D = x.shape
    sa = lyr.self_attn
    h = sa.num_heads
    hd = D // h
    keep = lambda site, shape: keeps[(layer, site)].view(shape) if keeps is not None and (layer, site) in keeps else 1.0
    qkv = rnd(F.linear(rnd(x), rnd(sa.in_proj_weight), sa.in_proj_bias))
    q, k, v = qkv.view(B, T, 3, h, hd).permute(2, 0, 3, 1, 4)
    s = q @ k.transpose(-1, -2) / math.sqrt(hd)
    if pad_mask is not None:
        s = s.masked_fill(pad_mask[:, None, None, :], float("-inf"))
    p = rnd(torch.softmax(s, -1) * keep(0, (B, h, T, T)))
    att = rnd((p @ v).permute(0, 2, 1, 3).reshape(B, T, D))
    y = F.linear(att, rnd(sa.out_proj.weight), sa.out_proj.bias)
    x1 = F.layer_norm(x + y * keep(1, (B, T, D)), (D,), lyr.norm1.weight, lyr.norm1.bias, lyr.norm1.eps)
    hid = rnd(F.relu(F.linear(rnd(x1), rnd(lyr.linear1.weight), lyr.linear1.bias)) * keep(2, (B, T, -1)))
    z = F.linear(hid, rnd(lyr.linear2.weight), lyr.linear2.bias)
    return F.layer_norm(x1 + z * keep(3, (B, T, D)), (D,), lyr.norm2.weight, lyr.norm2.bias, lyr.norm2.eps)


def encoder(x, enc, pad_mask=None, keeps=None, rnd=identity):
    """enc: torch.nn.TransformerEncoder (cogmen.py:99-101: 2 layers, no final norm)."""
    for i, lyr in enumerate(enc.layers):
        x = encoder_layer(x, lyr, pad_mask, keeps, i, rnd)
    return x
