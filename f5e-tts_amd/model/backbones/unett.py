"""UNetT constructor / state_dict mirror (reference model/backbones/unett.py:107-183).

The reference's samplers only ever call ``transformer.sample`` which exists on DiT alone, so UNetT is not
on the sampled path: this class keeps the constructor signature and the exact ``state_dict`` keys (so checkpoints
load), and its forward is the "next" row f4 of SURVEY section 8 -- not built yet, and it says so loudly."""
from __future__ import annotations

from typing import Literal

import torch
from torch import nn

from ..modules import Attention, AttnProcessor, FeedForward, TimestepEmbedding
from .dit import InputEmbedding, RotaryEmbedding, TextEmbedding


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

    def forward(self, *args, **kwargs):
        raise NotImplementedError("UNetT forward is SURVEY section 8 row f4 ('next'): the reference samplers cannot "
                                  "drive it (no .sample), so it is not part of the MI355X hot path yet")
