"""Seeded synthetic weights and inputs: F5TTS-shaped state dicts with the reference's
checkpoint key names, a random-init Vocos, band-limited reference waves and token ids.

NEUTRAL module: neither product code (``f5e-tts_amd/`` never imports it) nor part of the oracle.  ``bench.py``,
``tests/``, ``__graft_entry__.smoke()`` and the scripts under ``tools/`` use it to build identical inputs for the HIP path
and for the CPU oracle.  ``cfg`` is any object with the DiT architecture attributes (``oracle.f5e_oracle.DiTConfig`` or
``f5e_tts_amd.engine.DiTConfig``).
"""
from __future__ import annotations

import math
from typing import Dict

import torch
import torch.nn.functional as F

Tensor = torch.Tensor
State = Dict[str, Tensor]


class DiTConfig:
    """Architecture attributes of a DiT (reference backbones/dit.py:184-204 keyword names); defaults = F5TTS_v1_Base."""

    def __init__(self, dim=1024, depth=22, heads=16, dim_head=64, ff_mult=2, mel_dim=100, text_num_embeds=2545,
                 text_dim=512, text_mask_padding=True, qk_norm=None, conv_layers=4, pe_attn_head=None,
                 long_skip_connection=False, use_ppg=False, ppg_dim=256, conv_groups=16, ppg_transformer=False,
                 ppg_heads=4, ppg_ff=512, ppg_layers=2):
        self.__dict__.update(locals())
        del self.__dict__["self"]


def init_dit_state(cfg, seed: int = 1234, std_zeroed: float = 0.02) -> State:
    """Seeded random F5TTS-shaped state dict with the reference's key names.

    nn.Linear/Conv default-init scale (U(-1/sqrt(fan_in), +)) for ordinary layers; the tensors the
    reference zero-initialises (backbones/dit.py:273-283) get N(0, std_zeroed).
    """
    g = torch.Generator().manual_seed(seed)
    sd: State = {}

    def uni(shape, fan_in):
        bound = 1.0 / math.sqrt(fan_in)
        return (torch.rand(shape, generator=g) * 2 - 1) * bound

    def lin(name, out_f, in_f, bias=True):
        sd[name + ".weight"] = uni((out_f, in_f), in_f)
        if bias:
            sd[name + ".bias"] = uni((out_f,), in_f)

    def nrm(name, out_f, in_f):
        sd[name + ".weight"] = torch.randn((out_f, in_f), generator=g) * std_zeroed
        sd[name + ".bias"] = torch.randn((out_f,), generator=g) * std_zeroed

    d, td = cfg.dim, cfg.text_dim
    lin("time_embed.time_mlp.0", d, 256)
    lin("time_embed.time_mlp.2", d, d)
    sd["text_embed.text_embed.weight"] = torch.randn((cfg.text_num_embeds + 1, td), generator=g)
    for i in range(cfg.conv_layers):
        p = f"text_embed.text_blocks.{i}."
        sd[p + "dwconv.weight"] = uni((td, 1, 7), 7)
        sd[p + "dwconv.bias"] = uni((td,), 7)
        sd[p + "norm.weight"] = 1.0 + 0.1 * torch.randn((td,), generator=g)
        sd[p + "norm.bias"] = 0.1 * torch.randn((td,), generator=g)
        lin(p + "pwconv1", 2 * td, td)
        sd[p + "grn.gamma"] = 0.1 * torch.randn((1, 1, 2 * td), generator=g)
        sd[p + "grn.beta"] = 0.1 * torch.randn((1, 1, 2 * td), generator=g)
        lin(p + "pwconv2", td, 2 * td)
    if cfg.use_ppg and getattr(cfg, "ppg_transformer", False):   # PPGEmbedding(use_transformer=True), dit.py:105-119
        pd, pp = cfg.ppg_dim, "ppg_embed.ppg_proj."
        for i in range(cfg.ppg_layers):
            q = pp + f"0.layers.{i}."
            sd[q + "self_attn.in_proj_weight"] = uni((3 * pd, pd), pd)
            sd[q + "self_attn.in_proj_bias"] = uni((3 * pd,), pd)
            lin(q + "self_attn.out_proj", pd, pd)
            lin(q + "linear1", cfg.ppg_ff, pd)
            lin(q + "linear2", pd, cfg.ppg_ff)
            for n_ in ("norm1", "norm2"):
                sd[q + n_ + ".weight"] = 1.0 + 0.1 * torch.randn((pd,), generator=g)
                sd[q + n_ + ".bias"] = 0.1 * torch.randn((pd,), generator=g)
        lin(pp + "1", td, pd)
    elif cfg.use_ppg:
        pd = cfg.ppg_dim
        pp = "ppg_embed.ppg_proj."
        lin(pp + "0", pd, pd)
        for conv_i, bn_i in ((2, 3), (6, 7), (10, 11)):
            sd[pp + f"{conv_i}.weight"] = uni((pd, pd, 5), pd * 5)
            sd[pp + f"{conv_i}.bias"] = uni((pd,), pd * 5)
            sd[pp + f"{bn_i}.weight"] = 1.0 + 0.1 * torch.randn((pd,), generator=g)
            sd[pp + f"{bn_i}.bias"] = 0.1 * torch.randn((pd,), generator=g)
            sd[pp + f"{bn_i}.running_mean"] = 0.1 * torch.randn((pd,), generator=g)
            sd[pp + f"{bn_i}.running_var"] = 1.0 + 0.2 * torch.rand((pd,), generator=g)
            sd[pp + f"{bn_i}.num_batches_tracked"] = torch.tensor(0, dtype=torch.long)
        lin(pp + "15", td, pd)
    in_dim = cfg.mel_dim * 2 + td * (2 if cfg.use_ppg else 1)
    lin("input_embed.proj", d, in_dim)
    cg = d // getattr(cfg, "conv_groups", 16)
    for j in (0, 2):
        sd[f"input_embed.conv_pos_embed.conv1d.{j}.weight"] = uni((d, cg, 31), cg * 31)
        sd[f"input_embed.conv_pos_embed.conv1d.{j}.bias"] = uni((d,), cg * 31)
    sd["rotary_embed.inv_freq"] = 1.0 / (10000 ** (torch.arange(0, getattr(cfg, "dim_head", 64), 2).float() / getattr(cfg, "dim_head", 64)))
    inner = cfg.heads * getattr(cfg, "dim_head", 64)
    for i in range(cfg.depth):
        p = f"transformer_blocks.{i}."
        nrm(p + "attn_norm.linear", 6 * d, d)
        lin(p + "attn.to_q", inner, d)
        lin(p + "attn.to_k", inner, d)
        lin(p + "attn.to_v", inner, d)
        if cfg.qk_norm == "rms_norm":
            sd[p + "attn.q_norm.weight"] = 1.0 + 0.1 * torch.randn((getattr(cfg, "dim_head", 64),), generator=g)
            sd[p + "attn.k_norm.weight"] = 1.0 + 0.1 * torch.randn((getattr(cfg, "dim_head", 64),), generator=g)
        lin(p + "attn.to_out.0", d, inner)
        lin(p + "ff.ff.0.0", d * cfg.ff_mult, d)
        lin(p + "ff.ff.2", d, d * cfg.ff_mult)
    if cfg.long_skip_connection:
        lin("long_skip_connection", d, 2 * d, bias=False)
    nrm("norm_out.linear", 2 * d, d)
    nrm("proj_out", cfg.mel_dim, d)
    return sd


def init_vocos_state(seed: int = 4321, dim: int = 512, inter: int = 1536, layers: int = 8, n_mels: int = 100,
                     n_fft: int = 1024) -> State:
    """Random-init Vocos (charactr/vocos-mel-24khz architecture), gamma = 1/layers."""
    g = torch.Generator().manual_seed(seed)
    vs: State = {}

    def uni(shape, fan_in):
        bound = 1.0 / math.sqrt(fan_in)
        return (torch.rand(shape, generator=g) * 2 - 1) * bound

    vs["backbone.embed.weight"] = uni((dim, n_mels, 7), n_mels * 7)
    vs["backbone.embed.bias"] = uni((dim,), n_mels * 7)
    vs["backbone.norm.weight"] = torch.ones(dim)
    vs["backbone.norm.bias"] = torch.zeros(dim)
    for i in range(layers):
        p = f"backbone.convnext.{i}."
        vs[p + "dwconv.weight"] = uni((dim, 1, 7), 7)
        vs[p + "dwconv.bias"] = uni((dim,), 7)
        vs[p + "norm.weight"] = torch.ones(dim)
        vs[p + "norm.bias"] = torch.zeros(dim)
        vs[p + "pwconv1.weight"] = uni((inter, dim), dim)
        vs[p + "pwconv1.bias"] = uni((inter,), dim)
        vs[p + "pwconv2.weight"] = uni((dim, inter), inter)
        vs[p + "pwconv2.bias"] = uni((dim,), inter)
        vs[p + "gamma"] = torch.full((dim,), 1.0 / layers)
    vs["backbone.final_layer_norm.weight"] = torch.ones(dim)
    vs["backbone.final_layer_norm.bias"] = torch.zeros(dim)
    vs["head.out.weight"] = uni((n_fft + 2, dim), dim)
    vs["head.out.bias"] = uni((n_fft + 2,), dim)
    return vs


def synthetic_ref_wave(n_frames: int, seed: int = 2024, hop: int = 256, batch: int = 1) -> Tensor:
    """0.1*randn, 5-tap smoothed, RMS-normalised to 0.1; nw = hop*n_frames - 1 -> exactly n_frames mel frames
    when center=True (1 + nw // hop)."""
    g = torch.Generator().manual_seed(seed)
    nw = hop * (n_frames - 1) + hop // 2
    w = 0.1 * torch.randn((batch, nw + 4), generator=g)
    w = F.avg_pool1d(w[:, None], 5, stride=1)[:, 0]
    rms = w.pow(2).mean(dim=-1, keepdim=True).sqrt()
    return w * (0.1 / rms)


def synthetic_text_ids(n_total: int, batch: int = 1, seed: int = 7, vocab: int = 2545) -> Tensor:
    """Uniform ids in [1, vocab-1], nt = round(N/8)."""
    g = torch.Generator().manual_seed(seed)
    nt = max(1, round(n_total / 8))
    return torch.randint(1, vocab, (batch, nt), generator=g)
