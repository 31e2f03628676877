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
            if T <= 128:    # matrix-core attention (csrc/encoder_train.hip) without mask / dropout: 41 us against 141 us
                capi.enc_attention_train(ws["qkv"], B, T, D, L["heads"], None, 0.0, None, 0, ws["att"])
            else:
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


ENC_PARAM_NAMES = ("self_attn.in_proj_weight", "self_attn.in_proj_bias", "self_attn.out_proj.weight",
                   "self_attn.out_proj.bias", "linear1.weight", "linear1.bias", "linear2.weight", "linear2.bias",
                   "norm1.weight", "norm1.bias", "norm2.weight", "norm2.bias")


def encoder_live_groups(encoder, prefix="rnn.0.layers."):
    """FlatParams groups of the encoder for the chained mode (state_dict names of the reference module,
    track_mm/cogmen.py:94-109).  LayerNorm weight | bias share a group: their gradients leave one column-sum launch."""
    groups = []
    for i, lyr in enumerate(encoder.layers):
        pre = "%s%d." % (prefix, i)
        sa = lyr.self_attn
        groups += [[(pre + "self_attn.in_proj_weight", sa.in_proj_weight)], [(pre + "self_attn.in_proj_bias", sa.in_proj_bias)],
                   [(pre + "self_attn.out_proj.weight", sa.out_proj.weight)], [(pre + "self_attn.out_proj.bias", sa.out_proj.bias)],
                   [(pre + "linear1.weight", lyr.linear1.weight)], [(pre + "linear1.bias", lyr.linear1.bias)],
                   [(pre + "linear2.weight", lyr.linear2.weight)], [(pre + "linear2.bias", lyr.linear2.bias)],
                   [(pre + "norm1.weight", lyr.norm1.weight), (pre + "norm1.bias", lyr.norm1.bias)],
                   [(pre + "norm2.weight", lyr.norm2.weight), (pre + "norm2.bias", lyr.norm2.bias)]]
    return groups


class EncoderTrain:
    """The encoder of the chained COGMEN variant: ``rnn.1(rnn.0(x, src_key_padding_mask))`` -- what
    track_mm/cogmen.py:94-109 builds the encoder for -- in place of the reference's computed-and-discarded call
    (cogmen.py:145-147).  Opt-in, not parity with the reference's logits.

    Training-mode layer math of contrib/nn.py:283-305 (post-norm; dropout 0.5 on the attention probabilities, after the
    attention block, inside and after the feed-forward block), bf16 operands with fp32 accumulation, fp32 master weights
    in the module's flat buffer; every gradient is hand-written:

      forward, per layer   qkv GEMM -> masked attention -> out-proj GEMM -> add + dropout + LayerNorm ->
                           FFN GEMM (ReLU + dropout epilogue) -> FFN GEMM -> add + dropout + LayerNorm
      backward, per layer  LayerNorm bwd -> {bias colsum, weight NT product over the token axis, dgrad GEMM (ReLU/dropout
                           mask epilogue)} x 2 -> LayerNorm bwd -> out-proj grads -> attention bwd -> in-proj grads

    Weight-gradient products contract over the B*T token axis: both operands are transposed to [features, tokens]
    (erc_enc_transpose_bf16) so that the one NT GEMM kernel serves them.  bf16 copies of the weights (plain for the
    forward, transposed for the dgrad products) are rebuilt from the fp32 masters after every optimizer step
    (``refresh_shadows``)."""

    RNG_BASE = 0x100

    def __init__(self, encoder, flat, device, drop_p=0.5, prefix="rnn.0.layers."):
        self.flat, self.device, self.drop_p, self.prefix = flat, torch.device(device), drop_p, prefix
        self.n_layers = len(encoder.layers)
        l0 = encoder.layers[0]
        self.heads = l0.self_attn.num_heads
        self.D = l0.self_attn.embed_dim
        self.ffn = l0.linear1.out_features
        self.eps = (l0.norm1.eps, l0.norm2.eps)
        if self.D % 4 or self.ffn % 4:
            raise capi.ErcGraftError("chained encoder: feature width %d / ffn %d must be multiples of 4" % (self.D, self.ffn))
        h = lambda *s: torch.zeros(*s, dtype=torch.bfloat16, device=self.device)
        self.shadow = []
        for i in range(self.n_layers):
            sh = {}
            for key in ("self_attn.in_proj_weight", "self_attn.out_proj.weight", "linear1.weight", "linear2.weight"):
                R, Cn = flat.shapes[self._n(i, key)]
                sh[key] = (h(R, Cn), h(Cn, (R + 7) // 8 * 8))      # plain [out, in], transposed [in, out padded]
            self.shadow.append(sh)
        self._ws = {}
        self.refresh_shadows()

    def _n(self, i, key):
        return "%s%d.%s" % (self.prefix, i, key)

    def w(self, i, key):
        return self.flat.w(self._n(i, key))

    def g(self, i, key):
        return self.flat.g(self._n(i, key))

    def refresh_shadows(self):
        for i, sh in enumerate(self.shadow):
            for key, (plain, tr) in sh.items():
                W = self.w(i, key)
                R, Cn = W.shape
                capi.enc_transpose_bf16(W, Cn, R, Cn, tr, tr.shape[1], plain, Cn)

    def _workspace(self, M):
        ws = self._ws.get(M)
        if ws is None:
            D, F, dev = self.D, self.ffn, self.device
            Mp = (M + 7) // 8 * 8
            h = lambda *s: torch.zeros(*s, dtype=torch.bfloat16, device=dev)
            f = lambda *s: torch.zeros(*s, dtype=torch.float32, device=dev)
            layers = [dict(qkv=h(M, 3 * D), att=h(M, D), s1=f(M, D), st1=f(2 * M), x1f=f(M, D), x1h=h(M, D), hid=h(M, F),
                           s2=f(M, D), st2=f(2 * M), x2f=f(M, D), x2h=h(M, D)) for _ in range(self.n_layers)]
            nb = capi.enc_layernorm_bwd_blocks(M)
            ws = dict(layers=layers, Mp=Mp, xf=f(M, D), xh=h(M, D), tmp=f(M, D),
                      d_a=f(M, D), d_b=f(M, D), d_c=f(M, D), dzh=h(M, D), dpre=h(M, F), datt=h(M, D), dqkv=h(M, 3 * D),
                      tA=h(max(3 * D, F), Mp), tB=h(max(D, F), Mp), partial=f(nb, 2 * D), nb=nb,
                      cs_ws=f(capi.enc_colsum_ws_floats(max(3 * D, F))), inv=torch.zeros(M, dtype=torch.int32, device=dev),
                      dXn=None)
            self._ws[M] = ws
        return ws

    def _stream(self, layer, site):
        return self.RNG_BASE + 4 * layer + site

    def forward(self, x, lengths, training, rng_state):
        """x [B, T, D] fp32 / bf16; lengths int64 [B] (key-padding mask) or None.  Returns the bf16 output [B*T, D]."""
        B, T, D = x.shape
        M, F = B * T, self.ffn
        ws = self._workspace(M)
        p = self.drop_p if training else 0.0
        rng = rng_state if p > 0 else None
        if x.dtype == torch.bfloat16:
            ws["xh"].copy_(x.reshape(M, D))
            ws["xf"].copy_(ws["xh"])
        else:
            ws["xf"].copy_(x.reshape(M, D))
            capi.enc_to_bf16(ws["xf"], M * D, ws["xh"])
        xf, xh = ws["xf"], ws["xh"]
        for i, L in enumerate(ws["layers"]):
            sh = self.shadow[i]
            capi.enc_gemm_bf16(xh, D, sh["self_attn.in_proj_weight"][0], D, self.w(i, "self_attn.in_proj_bias"), None,
                               L["qkv"], 3 * D, M, 3 * D, D)
            capi.enc_attention_train(L["qkv"], B, T, D, self.heads, lengths, p, rng, self._stream(i, 0), L["att"])
            capi.enc_gemm_bf16(L["att"], D, sh["self_attn.out_proj.weight"][0], D, self.w(i, "self_attn.out_proj.bias"),
                               ws["tmp"], None, D, M, D, D)
            capi.enc_add_layernorm_train(xf, ws["tmp"], D, M, self.w(i, "norm1.weight"), self.w(i, "norm1.bias"), self.eps[0],
                                         p, rng, self._stream(i, 1), L["x1f"], L["x1h"], L["s1"], L["st1"])
            capi.enc_gemm_bf16_ex(L["x1h"], D, sh["linear1.weight"][0], D, self.w(i, "linear1.bias"), None, L["hid"], F, M, F,
                                  D, relu=1, epilogue=1 if p > 0 else 0, scale=1.0 / (1.0 - p), drop_p=p, rng_state=rng,
                                  rng_stream=self._stream(i, 2))
            capi.enc_gemm_bf16(L["hid"], F, sh["linear2.weight"][0], F, self.w(i, "linear2.bias"), ws["tmp"], None, D, M, D, F)
            capi.enc_add_layernorm_train(L["x1f"], ws["tmp"], D, M, self.w(i, "norm2.weight"), self.w(i, "norm2.bias"),
                                         self.eps[1], p, rng, self._stream(i, 3), L["x2f"], L["x2h"], L["s2"], L["st2"])
            L["xh_in"] = xh
            xf, xh = L["x2f"], L["x2h"]
        ws["B"], ws["T"], ws["p"], ws["rng"], ws["lengths"] = B, T, p, rng, lengths
        self._last = ws
        return xh

    def _wgrad(self, ws, dy, n_out, x, n_in, M, wgrad, bgrad):
        """wgrad[n_out, n_in] = dy[M, n_out]^T x[M, n_in] (bf16 operands), bgrad = column sums of dy."""
        Mp = ws["Mp"]
        capi.enc_colsum(dy, n_out, M, n_out, bgrad, ws["cs_ws"])
        capi.enc_transpose_bf16(dy, n_out, M, n_out, ws["tA"], Mp)
        capi.enc_transpose_bf16(x, n_in, M, n_in, ws["tB"], Mp)
        capi.enc_gemm_bf16(ws["tA"], Mp, ws["tB"], Mp, None, wgrad, None, n_in, n_out, n_in, Mp)

    def backward(self, d_out, row_map=None):
        """d_out: gradient w.r.t. the encoder output, fp32 [M, D] -- or [N, D] with ``row_map`` int32 [M] giving the
        source row of every output row (-1: zero).  Writes every encoder gradient into the flat gradient buffer."""
        D, F = self.D, self.ffn
        ws = self._last            # the workspace of the forward this backward belongs to
        B, T, p, rng, lengths = ws["B"], ws["T"], ws["p"], ws["rng"], ws["lengths"]
        M = B * T
        ks = 1.0 / (1.0 - p)
        dy_a, dy_map, dy_b = d_out, row_map, None
        for i in reversed(range(self.n_layers)):
            L, sh = ws["layers"][i], self.shadow[i]
            # LayerNorm 2 -> ds2 (residual into x1), dz (through dropout 2)
            capi.enc_layernorm_bwd(dy_a, dy_map, dy_b, L["s2"], L["st2"], self.w(i, "norm2.weight"), D, M, p, rng,
                                   self._stream(i, 3), ws["d_a"], ws["dzh"], ws["partial"])
            capi.enc_colsum(ws["partial"], 2 * D, ws["nb"], 2 * D, self.g(i, "norm2.weight"), ws["cs_ws"])
            self._wgrad(ws, ws["dzh"], D, L["hid"], F, M, self.g(i, "linear2.weight"), self.g(i, "linear2.bias"))
            capi.enc_gemm_bf16_ex(ws["dzh"], D, sh["linear2.weight"][1], sh["linear2.weight"][1].shape[1], None, None,
                                  ws["dpre"], F, M, F, D, epilogue=2, mask_src=L["hid"], ld_mask=F, scale=ks)
            self._wgrad(ws, ws["dpre"], F, L["x1h"], D, M, self.g(i, "linear1.weight"), self.g(i, "linear1.bias"))
            capi.enc_gemm_bf16(ws["dpre"], F, sh["linear1.weight"][1], sh["linear1.weight"][1].shape[1], None, ws["d_b"], None,
                               D, M, D, F)
            # LayerNorm 1 -> ds1 (residual into the layer input), dy (through dropout 1)
            capi.enc_layernorm_bwd(ws["d_a"], None, ws["d_b"], L["s1"], L["st1"], self.w(i, "norm1.weight"), D, M, p, rng,
                                   self._stream(i, 1), ws["d_c"], ws["dzh"], ws["partial"])
            capi.enc_colsum(ws["partial"], 2 * D, ws["nb"], 2 * D, self.g(i, "norm1.weight"), ws["cs_ws"])
            self._wgrad(ws, ws["dzh"], D, L["att"], D, M, self.g(i, "self_attn.out_proj.weight"),
                        self.g(i, "self_attn.out_proj.bias"))
            capi.enc_gemm_bf16(ws["dzh"], D, sh["self_attn.out_proj.weight"][1], sh["self_attn.out_proj.weight"][1].shape[1],
                               None, None, ws["datt"], D, M, D, D)
            capi.enc_attention_bwd(L["qkv"], ws["datt"], B, T, D, self.heads, lengths, p, rng, self._stream(i, 0), ws["dqkv"])
            self._wgrad(ws, ws["dqkv"], 3 * D, L["xh_in"], D, M, self.g(i, "self_attn.in_proj_weight"),
                        self.g(i, "self_attn.in_proj_bias"))
            if i > 0:
                capi.enc_gemm_bf16(ws["dqkv"], 3 * D, sh["self_attn.in_proj_weight"][1],
                                   sh["self_attn.in_proj_weight"][1].shape[1], None, ws["d_b"], None, D, M, D, 3 * D)
                # gradient w.r.t. the previous layer's output = residual part (d_c) + in-proj part (d_b)
                dy_a, dy_map, dy_b = ws["d_c"], None, ws["d_b"]
