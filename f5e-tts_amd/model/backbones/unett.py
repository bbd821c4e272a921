"""UNetT mirror (reference model/backbones/unett.py:107-250): constructor signature, exact ``state_dict`` keys and
the inference ``forward`` (time token in front, x_transformers RMSNorm, U-shaped skip connections) on libf5e_hip.so.

The reference's samplers only ever call ``transformer.sample``, which exists on DiT alone, so UNetT is not
on the sampled path; this is SURVEY section 8 row f4, built from the same kernels as the DiT block (no AdaLN: the gate
of the residual epilogue is a vector of ones) plus ``f5e_l2norm``."""
from __future__ import annotations

from typing import Literal

import torch
from torch import nn

from ... import _C, ops
from ...engine import DiTConfig, DiTEngine
from ..modules import Attention, AttnProcessor, FeedForward, TimestepEmbedding
from .dit import InputEmbedding, RotaryEmbedding, TextEmbedding

BF, F32, I32 = torch.bfloat16, torch.float32, torch.int32


class XRMSNorm(nn.Module):
    """Container for x_transformers.RMSNorm (single parameter ``g``)."""

    def __init__(self, dim):
        super().__init__()
        self.scale = dim ** 0.5
        self.g = nn.Parameter(torch.ones(dim))


class UNetT(nn.Module):
    def __init__(self, *, dim, depth=8, heads=8, dim_head=64, dropout=0.1, ff_mult=4, mel_dim=100,
                 text_num_embeds=256, text_dim=None, text_mask_padding=True, qk_norm=None, conv_layers=0,
                 pe_attn_head=None, skip_connect_type: Literal["add", "concat", "none"] = "concat"):
        super().__init__()
        assert depth % 2 == 0, "UNet-Transformer's depth should be even."
        self.time_embed = TimestepEmbedding(dim)
        if text_dim is None:
            text_dim = mel_dim
        self.text_embed = TextEmbedding(text_num_embeds, text_dim, mask_padding=text_mask_padding,
                                        conv_layers=conv_layers)
        self.text_cond, self.text_uncond = None, None
        self.input_embed = InputEmbedding(mel_dim, text_dim, dim, use_ppg=False)
        self.rotary_embed = RotaryEmbedding(dim_head)
        self.dim, self.depth, self.skip_connect_type = dim, depth, skip_connect_type
        self.layers = nn.ModuleList([])
        for idx in range(depth):
            later = idx >= depth // 2
            skip_proj = nn.Linear(dim * 2, dim, bias=False) if (skip_connect_type == "concat" and later) else None
            self.layers.append(nn.ModuleList([
                skip_proj, XRMSNorm(dim),
                Attention(processor=AttnProcessor(pe_attn_head=pe_attn_head), dim=dim, heads=heads, dim_head=dim_head,
                          dropout=dropout, qk_norm=qk_norm),
                XRMSNorm(dim), FeedForward(dim=dim, mult=ff_mult, dropout=dropout, approximate="tanh")]))
        self.norm_out = XRMSNorm(dim)
        self.proj_out = nn.Linear(dim, mel_dim)

    def clear_cache(self):
        self.text_cond, self.text_uncond = None, None

    # ------------------------------------------------------------------ HIP forward

    def _packed(self):
        tensors = list(self.parameters()) + list(self.buffers())
        dv = tensors[0].device
        sig = tuple((t.data_ptr(), t._version) for t in tensors) + (str(dv),)
        if getattr(self, "_pack", None) is None or self._pack[0] != sig:
            if dv.type != "cuda":
                raise _C.F5EError(f"UNetT lives on {dv}: move it to the GPU (there is no CPU path)")
            f = lambda t: t.detach().to(dv, F32).contiguous()  # noqa: E731
            te = self.text_embed
            cfg = DiTConfig(dim=self.dim, depth=0, heads=self.layers[0][2].heads, mel_dim=self.proj_out.out_features,
                            text_num_embeds=te.text_embed.num_embeddings - 1, text_dim=te.text_embed.embedding_dim,
                            text_mask_padding=te.mask_padding,
                            conv_layers=len(te.text_blocks) if te.extra_modeling else 0)
            front = DiTEngine(self.state_dict(), cfg, dv, blocks=False)
            layers = []
            for skip_proj, attn_norm, attn, ff_norm, ff in self.layers:
                layers.append(dict(
                    skip=f(skip_proj.weight) if skip_proj is not None else None, g_attn=f(attn_norm.g), g_ff=f(ff_norm.g),
                    w_qkv=torch.cat([f(attn.to_q.weight), f(attn.to_k.weight), f(attn.to_v.weight)], 0).to(BF),
                    b_qkv=torch.cat([f(attn.to_q.bias), f(attn.to_k.bias), f(attn.to_v.bias)], 0),
                    qn=f(attn.q_norm.weight) if attn.q_norm is not None else None,
                    kn=f(attn.k_norm.weight) if attn.k_norm is not None else None,
                    w_out=f(attn.to_out[0].weight).to(BF), b_out=f(attn.to_out[0].bias),
                    w1=f(ff.ff[0][0].weight).to(BF), b1=f(ff.ff[0][0].bias),
                    w2=f(ff.ff[2].weight).to(BF), b2=f(ff.ff[2].bias), pe=attn.processor.pe_attn_head))
            self._pack = (sig, dict(front=front, layers=layers, g_out=f(self.norm_out.g),
                                    w_proj=f(self.proj_out.weight).to(BF), b_proj=f(self.proj_out.bias),
                                    ones=torch.ones(1, self.dim, device=dv)))
        return self._pack[1]

    @torch.no_grad()
    def forward(self, x, cond, text, time, drop_audio_cond, drop_text, mask=None, cache=False):
        """reference backbones/unett.py:184-250.  x, cond [b, n, mel]; text int [b, nt]; mask bool [b, n] or None."""
        P = self._packed()
        eng = P["front"]
        dv = eng.device
        B, N = x.shape[0], x.shape[1]
        D, H = self.dim, self.layers[0][2].heads
        if time.ndim == 0:
            time = time.repeat(B)
        t = eng.time_embed(time.to(dv, F32))
        if cache:
            attr = "text_uncond" if drop_text else "text_cond"
            if getattr(self, attr) is None:
                setattr(self, attr, eng.text_embed(text, B, N, drop_text))
            text_embed = getattr(self, attr)
        else:
            text_embed = eng.text_embed(text, B, N, drop_text)
        in_const = torch.empty(B * N, D, device=dv)
        eng.input_const(cond.to(dv, F32).contiguous(), text_embed, None, drop_audio_cond, in_const)
        mel = x.shape[-1]
        h0 = torch.empty(B * N, D, device=dv)
        h0b = torch.empty(B * N, D, device=dv, dtype=BF)
        ops.gemm_f32(x.to(dv, F32).reshape(B * N, mel).contiguous(), eng.in_w[:, :mel], None, out=h0, out_bf16=h0b,
                     addend=in_const)
        c1 = torch.empty_like(h0b)
        xs = torch.empty_like(h0)
        ops.convpos(h0b, eng.cp[0][0], eng.cp[0][1], B, N, out_bf16=c1)
        ops.convpos(c1, eng.cp[1][0], eng.cp[1][1], B, N, out_f32=xs, resid=h0)
        n1 = N + 1
        X = torch.cat([t.unsqueeze(1), xs.view(B, N, D)], dim=1).contiguous().view(B * n1, D)  # time token in front
        lens = None
        if mask is not None:
            lens = (mask.sum(-1) + 1).to(I32).contiguous()
            if not torch.equal(mask, torch.arange(N, device=mask.device)[None] < (lens[:, None] - 1)):
                raise _C.F5EError("UNetT.forward takes key-padding masks of the lens_to_mask form only")
        n_pad = (n1 + 63) // 64 * 64
        cs = eng.rope_table(n1)
        q = torch.zeros(B, H, n_pad, 64, device=dv, dtype=BF)
        k, v = torch.zeros_like(q), torch.zeros_like(q)
        hn = torch.empty(B * n1, D, device=dv, dtype=BF)
        ao = torch.empty(B * n1, H * 64, device=dv, dtype=BF)
        one = torch.ones(1, device=dv)
        depth, skips = len(P["layers"]), []
        for idx, L in enumerate(P["layers"]):
            if idx + 1 <= depth // 2:
                skips.append(X.clone())
            else:
                skip = skips.pop()
                if self.skip_connect_type == "concat":
                    tmp = torch.empty_like(X)
                    ops.gemm_f32(X, L["skip"][:, :D], None, out=tmp)
                    Xn = torch.empty_like(X)
                    ops.gemm_f32(skip, L["skip"][:, D:], None, out=Xn, addend=tmp)
                    X = Xn
                elif self.skip_connect_type == "add":
                    ops.ode_update(skip, 0, 0, 0.0, 0.0, X, X, one, None)  # X += 1.0 * skip
            ops.l2norm(X, hn, L["g_attn"])
            ops.gemm_bf16_qkv_rope(hn, L["w_qkv"], L["b_qkv"], q, k, v, H, H if L["pe"] is None else L["pe"], cs, n1,
                                   q_norm_w=L["qn"], k_norm_w=L["kn"])
            ops.flash_attn(q, k, v, ao, n1, kv_len=lens)
            ops.gemm_bf16_gate_residual(ao, L["w_out"], L["b_out"], X, P["ones"], n1, seq_len=lens)
            ops.l2norm(X, hn, L["g_ff"])
            ff = torch.empty(B * n1, L["w1"].shape[0], device=dv, dtype=BF)
            ops.gemm_bf16_bias(hn, L["w1"], L["b1"], ff, act=ops.ACT_GELU_TANH)
            ops.gemm_bf16_gate_residual(ff, L["w2"], L["b2"], X, P["ones"], n1)
        assert not skips
        ops.l2norm(X, hn, P["g_out"])
        out = torch.empty(B * n1, mel, device=dv)
        ops.gemm_bf16_bias(hn, P["w_proj"], P["b_proj"], out)
        return out.view(B, n1, mel)[:, 1:, :].contiguous()
