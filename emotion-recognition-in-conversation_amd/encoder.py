"""K9: COGMEN's (dead) Transformer encoder ``rnn.0`` on the matrix cores -- the faithful-cost mode of SURVEY.md 8a C2.

The reference applies ``rnn.0`` (2 x TransformerEncoderLayer: d_model = D, nhead = first divisor of D in [6, 17),
ffn 2048, post-norm, ReLU, batch_first, no padding mask; track_mm/cogmen.py:86-102) to the raw ``[B, T, D]`` block and
throws the result away (cogmen.py:146-147).  ``EncoderBlock.forward`` does the same arithmetic (inference-mode: the
dropout layers are identities -- the output is discarded either way) with bf16 operands and fp32 accumulation:
four dense products per layer through ``erc_enc_gemm_bf16``, attention over the T padded positions of every
(dialogue, head), residual + LayerNorm fused.  The 26.6 M encoder parameters never train, so their bf16 copies are made
once.
"""
import torch

from . import capi


class EncoderBlock:
    def __init__(self, encoder, device):
        """``encoder``: the module's ``rnn[0]`` (torch.nn.TransformerEncoder); only its parameters are read."""
        self.device = torch.device(device)
        self.layers = []
        for lyr in encoder.layers:
            sa = lyr.self_attn
            bf = lambda t: t.detach().to(self.device, torch.bfloat16).contiguous()
            f32 = lambda t: t.detach().to(self.device, torch.float32).contiguous()
            self.layers.append(dict(
                heads=sa.num_heads, Win=bf(sa.in_proj_weight), b_in=f32(sa.in_proj_bias),
                Wo=bf(sa.out_proj.weight), bo=f32(sa.out_proj.bias),
                W1=bf(lyr.linear1.weight), b1=f32(lyr.linear1.bias), W2=bf(lyr.linear2.weight), b2=f32(lyr.linear2.bias),
                g1=f32(lyr.norm1.weight), be1=f32(lyr.norm1.bias), eps1=lyr.norm1.eps,
                g2=f32(lyr.norm2.weight), be2=f32(lyr.norm2.bias), eps2=lyr.norm2.eps))
        self.D = self.layers[0]["Wo"].shape[0]
        self.ffn = self.layers[0]["W1"].shape[0]
        self._ws = {}

    def flops(self, B, T):
        M, D, F = B * T, self.D, self.ffn
        per_layer = 2 * M * D * (3 * D + D + 2 * F) + 4 * B * T * T * D
        return per_layer * len(self.layers)

    def _workspace(self, M):
        ws = self._ws.get(M)
        if ws is None:
            D, F, dev = self.D, self.ffn, self.device
            h = lambda *s: torch.empty(*s, dtype=torch.bfloat16, device=dev)
            f = lambda *s: torch.empty(*s, dtype=torch.float32, device=dev)
            ws = dict(xh=h(M, D), xf=f(M, D), qkv=h(M, 3 * D), att=h(M, D), y=f(M, D), x1f=f(M, D), x1h=h(M, D),
                      hid=h(M, F), z=f(M, D), x2f=f(M, D), x2h=h(M, D))
            self._ws[M] = ws
        return ws

    def forward(self, x):
        """x: [B, T, D] fp32 or bf16 on the device.  Returns the encoder output [B, T, D] fp32 (a workspace view)."""
        B, T, D = x.shape
        M = B * T
        ws = self._workspace(M)
        if x.dtype == torch.bfloat16:
            xh = x.contiguous().view(M, D)
            ws["xf"].copy_(xh)
        else:
            ws["xf"].copy_(x.reshape(M, D))
            capi.enc_to_bf16(ws["xf"], M * D, ws["xh"])
            xh = ws["xh"]
        xf = ws["xf"]
        for L in self.layers:
            capi.enc_gemm_bf16(xh, D, L["Win"], D, L["b_in"], None, ws["qkv"], 3 * D, M, 3 * D, D)
            capi.enc_attention(ws["qkv"], B, T, D, L["heads"], ws["att"])
            capi.enc_gemm_bf16(ws["att"], D, L["Wo"], D, L["bo"], ws["y"], None, D, M, D, D)
            capi.enc_add_layernorm(xf, ws["y"], D, M, L["g1"], L["be1"], L["eps1"], ws["x1f"], ws["x1h"])
            capi.enc_gemm_bf16(ws["x1h"], D, L["W1"], D, L["b1"], None, ws["hid"], self.ffn, M, self.ffn, D, relu=1)
            capi.enc_gemm_bf16(ws["hid"], self.ffn, L["W2"], self.ffn, L["b2"], ws["z"], None, D, M, D, self.ffn)
            capi.enc_add_layernorm(ws["x1f"], ws["z"], D, M, L["g2"], L["be2"], L["eps2"], ws["x2f"], ws["x2h"])
            # the next layer reads this layer's output; ping-pong through the two output buffers
            xf, xh = ws["x2f"], ws["x2h"]
            ws["x2f"], ws["xf"] = ws["xf"], ws["x2f"]
            ws["x2h"], ws["xh"] = ws["xh"], ws["x2h"]
        return xf.view(B, T, D)
