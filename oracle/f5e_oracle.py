"""CPU oracle for the F5E-TTS flow-matching inference hot path.

TEST INFRASTRUCTURE ONLY.  Nothing under ``oracle/`` is imported by the product
package ``f5e-tts_amd/``; only ``tests/``, ``__graft_entry__.smoke()`` and the
``cpu_baseline`` leg of ``bench.py`` may use it, and there only as the checker
or as the reported CPU baseline -- never as the thing measured or shipped.

This is a plain PyTorch fp32 *restatement* (functional, over a flat
``{name: tensor}`` state dict that uses the reference's checkpoint key names)
of the algorithm the reference runs for this path.  Every function cites the
reference file:line it follows (paths relative to ``/root/reference/src/f5_tts``).

Pinning status (see DESIGN.md "Oracle"):
  * the reference's own arithmetic (DiT, DiTBlock, AdaLN, ConvPositionEmbedding,
    TextEmbedding/ConvNeXtV2/GRN, PPGEmbedding, TimestepEmbedding, CFM.sample*
    prep + CFG combination) is pinned by ``tests/golden/*.npz``, produced by
    ``tests/golden/make_golden.py`` from the reference itself, imported in the
    build container.
  * arithmetic that lives in un-vendored third-party packages (x_transformers
    RoPE, torchdiffeq fixed-grid solvers, torchaudio MelSpectrogram, vocos) is
    restated from the published algorithms; the reference tree holds no golden
    vectors for it, so for those functions: PARITY UNPINNED (cross-checked only
    against torch.stft / torch.istft, which are available).
"""
from __future__ import annotations

import math
from typing import Dict, List, Optional, Sequence, Tuple

import torch
import torch.nn.functional as F

Tensor = torch.Tensor
State = Dict[str, Tensor]


# --------------------------------------------------------------------------
# host-side helpers                                   (model/utils.py:41-100)
# --------------------------------------------------------------------------

def lens_to_mask(lens: Tensor, length: Optional[int] = None) -> Tensor:
    """model/utils.py:41-46 -- ``arange(length) < lens[:, None]``."""
    if length is None:
        length = int(lens.max())
    return torch.arange(length, device=lens.device)[None, :] < lens[:, None]


def list_str_to_idx(text: Sequence, vocab: Dict[str, int], pad: int = -1) -> Tensor:
    """model/utils.py:87-100 -- vocab.get(c, 0), right-pad with -1."""
    rows = [[vocab.get(c, 0) for c in t] for t in text]
    n = max(len(r) for r in rows)
    out = torch.full((len(rows), n), pad, dtype=torch.long)
    for i, r in enumerate(rows):
        out[i, : len(r)] = torch.tensor(r, dtype=torch.long)
    return out


def list_str_to_tensor(text: Sequence[str], pad: int = -1) -> Tensor:
    """model/utils.py:80-83 -- UTF-8 bytes, right-pad with -1."""
    rows = [list(bytes(t, "UTF-8")) for t in text]
    n = max(len(r) for r in rows)
    out = torch.full((len(rows), n), pad, dtype=torch.long)
    for i, r in enumerate(rows):
        out[i, : len(r)] = torch.tensor(r, dtype=torch.long)
    return out


# --------------------------------------------------------------------------
# K1 log-mel front-end      (model/modules.py:75-101; torchaudio, UNPINNED)
# --------------------------------------------------------------------------

def mel_filterbank_htk(n_freqs: int = 513, n_mels: int = 100, sr: int = 24000,
                       f_min: float = 0.0, f_max: Optional[float] = None) -> Tensor:
    """torchaudio.functional.melscale_fbanks(norm=None, mel_scale="htk") -> [n_freqs, n_mels]."""
    f_max = float(sr // 2) if f_max is None else f_max
    all_freqs = torch.linspace(0, sr // 2, n_freqs)
    m_min = 2595.0 * math.log10(1.0 + f_min / 700.0)
    m_max = 2595.0 * math.log10(1.0 + f_max / 700.0)
    m_pts = torch.linspace(m_min, m_max, n_mels + 2)
    f_pts = 700.0 * (10.0 ** (m_pts / 2595.0) - 1.0)
    f_diff = f_pts[1:] - f_pts[:-1]
    slopes = f_pts.unsqueeze(0) - all_freqs.unsqueeze(1)
    down = (-1.0 * slopes[:, :-2]) / f_diff[:-1]
    up = slopes[:, 2:] / f_diff[1:]
    return torch.clamp(torch.min(down, up), min=0.0)


def log_mel_spectrogram(wav: Tensor, n_fft: int = 1024, hop: int = 256, win: int = 1024,
                        n_mels: int = 100, sr: int = 24000) -> Tensor:
    """get_vocos_mel_spectrogram, modules.py:75-101 -> [B, n_mels, 1 + nw // hop]."""
    if wav.ndim == 3:
        wav = wav.squeeze(1)
    window = torch.hann_window(win, periodic=True, dtype=torch.float32)
    spec = torch.stft(wav.float(), n_fft, hop_length=hop, win_length=win, window=window, center=True,
                      pad_mode="reflect", normalized=False, onesided=True, return_complex=True)
    mag = spec.abs()  # power = 1
    fb = mel_filterbank_htk(n_fft // 2 + 1, n_mels, sr)
    mel = torch.matmul(mag.transpose(-1, -2), fb).transpose(-1, -2)
    return mel.clamp(min=1e-5).log()


# --------------------------------------------------------------------------
# K2 time embedding                         (model/modules.py:149-161,721-731)
# --------------------------------------------------------------------------

def sinus_embedding(t: Tensor, dim: int = 256, scale: float = 1000.0) -> Tensor:
    half = dim // 2
    k = math.log(10000) / (half - 1)
    freqs = torch.exp(torch.arange(half).float() * -k)
    arg = scale * t.float().unsqueeze(1) * freqs.unsqueeze(0)
    return torch.cat((arg.sin(), arg.cos()), dim=-1)


def time_embedding(sd: State, t: Tensor, prefix: str = "time_embed.") -> Tensor:
    h = sinus_embedding(t).to(t.dtype)
    h = F.linear(h, sd[prefix + "time_mlp.0.weight"], sd[prefix + "time_mlp.0.bias"])
    h = F.silu(h)
    return F.linear(h, sd[prefix + "time_mlp.2.weight"], sd[prefix + "time_mlp.2.bias"])


# --------------------------------------------------------------------------
# K3 text embedding          (backbones/dit.py:37-87; modules.py:196-269)
# --------------------------------------------------------------------------

def text_pos_table(dim: int, end: int = 4096, theta: float = 10000.0) -> Tensor:
    """precompute_freqs_cis, modules.py:196-207 -> [end, dim] = cos || sin."""
    freqs = 1.0 / (theta ** (torch.arange(0, dim, 2)[: dim // 2].float() / dim))
    ang = torch.outer(torch.arange(end), freqs).float()
    return torch.cat([ang.cos(), ang.sin()], dim=-1)


def grn(x: Tensor, gamma: Tensor, beta: Tensor) -> Tensor:
    """GRN, modules.py:225-234 (L2 norm over the SEQUENCE axis, dim=1)."""
    gx = torch.norm(x, p=2, dim=1, keepdim=True)
    nx = gx / (gx.mean(dim=-1, keepdim=True) + 1e-6)
    return gamma * (x * nx) + beta + x


def convnext_v2_block(sd: State, p: str, x: Tensor) -> Tensor:
    """ConvNeXtV2Block.forward, modules.py:259-269."""
    c = x.shape[-1]
    h = F.conv1d(x.transpose(1, 2), sd[p + "dwconv.weight"], sd[p + "dwconv.bias"], padding=3, groups=c)
    h = h.transpose(1, 2)
    h = F.layer_norm(h, (c,), sd[p + "norm.weight"], sd[p + "norm.bias"], eps=1e-6)
    h = F.linear(h, sd[p + "pwconv1.weight"], sd[p + "pwconv1.bias"])
    h = F.gelu(h)
    h = grn(h, sd[p + "grn.gamma"], sd[p + "grn.beta"])
    h = F.linear(h, sd[p + "pwconv2.weight"], sd[p + "pwconv2.bias"])
    return x + h


def text_embedding(sd: State, text: Optional[Tensor], batch: int, seq_len: int, drop_text: bool,
                   mask_padding: bool = True, prefix: str = "text_embed.") -> Tensor:
    """TextEmbedding.forward, backbones/dit.py:54-87."""
    emb_w = sd[prefix + "text_embed.weight"]
    text_mask = None
    if text is None:
        ids = torch.zeros((batch, seq_len), dtype=torch.long)
    else:
        ids = (text + 1)[:, :seq_len]
        ids = F.pad(ids, (0, seq_len - ids.shape[1]), value=0)
        if mask_padding:
            text_mask = ids == 0  # taken BEFORE the drop (dit.py:62-66)
        if drop_text:
            ids = torch.zeros_like(ids)
    h = F.embedding(ids, emb_w)
    n_conv = 0
    while (prefix + f"text_blocks.{n_conv}.dwconv.weight") in sd:
        n_conv += 1
    if n_conv > 0:
        table = text_pos_table(emb_w.shape[1])
        pos = torch.arange(seq_len).clamp(max=4095)
        h = h + table[pos][None]
        if text_mask is not None:
            keep = (~text_mask).unsqueeze(-1).to(h.dtype)
            h = h * keep
            for i in range(n_conv):
                h = convnext_v2_block(sd, prefix + f"text_blocks.{i}.", h) * keep
        else:
            for i in range(n_conv):
                h = convnext_v2_block(sd, prefix + f"text_blocks.{i}.", h)
    return h


# --------------------------------------------------------------------------
# K17 PPG embedding                                (backbones/dit.py:93-153)
# --------------------------------------------------------------------------

def ppg_embedding_transformer(sd: State, h: Tensor, heads: int, prefix: str) -> Tensor:
    """use_transformer=True variant (backbones/dit.py:105-119): nn.TransformerEncoder of post-norm layers (PyTorch
    defaults: norm_first=False, eps 1e-5) with GELU(erf), batch_first, NO padding mask, then Linear(ppg_dim, text_dim)."""
    i = 0
    while prefix + f"0.layers.{i}.self_attn.in_proj_weight" in sd:
        p = prefix + f"0.layers.{i}."
        b, n, d = h.shape
        dh = d // heads
        q, k, v = F.linear(h, sd[p + "self_attn.in_proj_weight"], sd[p + "self_attn.in_proj_bias"]).chunk(3, dim=-1)
        q, k, v = (t.view(b, n, heads, dh).transpose(1, 2) for t in (q, k, v))
        a = torch.softmax(q @ k.transpose(-2, -1) / math.sqrt(dh), dim=-1) @ v
        a = F.linear(a.transpose(1, 2).reshape(b, n, d), sd[p + "self_attn.out_proj.weight"], sd[p + "self_attn.out_proj.bias"])
        h = F.layer_norm(h + a, (d,), sd[p + "norm1.weight"], sd[p + "norm1.bias"], eps=1e-5)
        f = F.linear(F.gelu(F.linear(h, sd[p + "linear1.weight"], sd[p + "linear1.bias"])), sd[p + "linear2.weight"],
                     sd[p + "linear2.bias"])
        h = F.layer_norm(h + f, (d,), sd[p + "norm2.weight"], sd[p + "norm2.bias"], eps=1e-5)
        i += 1
    return F.linear(h, sd[prefix + "1.weight"], sd[prefix + "1.bias"])


def ppg_embedding(sd: State, ppg: Optional[Tensor], batch: int, seq_len: int, drop_ppg: bool,
                  prefix: str = "ppg_embed.ppg_proj.", heads: int = 4) -> Tensor:
    transformer = prefix + "0.layers.0.self_attn.in_proj_weight" in sd
    w0 = sd[prefix + ("1.weight" if transformer else "0.weight")]
    ppg_dim = w0.shape[1]
    if ppg is None:
        h = torch.zeros((batch, seq_len, ppg_dim), dtype=w0.dtype)
    else:
        h = F.pad(ppg.to(w0.dtype), (0, 0, 0, seq_len - ppg.shape[1]), value=0)
        if drop_ppg:
            h = torch.zeros_like(h)
    if transformer:
        return ppg_embedding_transformer(sd, h, heads, prefix)
    h = F.linear(h, w0, sd[prefix + "0.bias"]).transpose(1, 2)
    for conv_i, bn_i in ((2, 3), (6, 7), (10, 11)):
        h = F.conv1d(h, sd[prefix + f"{conv_i}.weight"], sd[prefix + f"{conv_i}.bias"], padding=2)
        h = F.batch_norm(h, sd[prefix + f"{bn_i}.running_mean"], sd[prefix + f"{bn_i}.running_var"],
                         sd[prefix + f"{bn_i}.weight"], sd[prefix + f"{bn_i}.bias"], training=False, eps=1e-5)
        h = F.relu(h)
    h = h.transpose(1, 2)
    return F.linear(h, sd[prefix + "15.weight"], sd[prefix + "15.bias"])


# --------------------------------------------------------------------------
# K4/K5 input embedding     (backbones/dit.py:159-177; modules.py:167-190)
# --------------------------------------------------------------------------

def conv_pos_embedding(sd: State, p: str, x: Tensor, groups: int = 16) -> Tensor:
    """ConvPositionEmbedding.forward with mask=None (dit.py:176)."""
    h = x.permute(0, 2, 1)
    h = F.mish(F.conv1d(h, sd[p + "conv1d.0.weight"], sd[p + "conv1d.0.bias"], padding=15, groups=groups))
    h = F.mish(F.conv1d(h, sd[p + "conv1d.2.weight"], sd[p + "conv1d.2.bias"], padding=15, groups=groups))
    return h.permute(0, 2, 1)


def input_embedding(sd: State, x: Tensor, cond: Tensor, text_emb: Tensor, ppg_emb: Optional[Tensor],
                    drop_audio_cond: bool, prefix: str = "input_embed.") -> Tensor:
    if drop_audio_cond:
        cond = torch.zeros_like(cond)
    parts = (x, cond, text_emb) if ppg_emb is None else (x, cond, text_emb, ppg_emb)
    h = F.linear(torch.cat(parts, dim=-1), sd[prefix + "proj.weight"], sd[prefix + "proj.bias"])
    return conv_pos_embedding(sd, prefix + "conv_pos_embed.", h) + h


# --------------------------------------------------------------------------
# K6/K9 rotary embedding            (x_transformers, UNPINNED)
# --------------------------------------------------------------------------

def rope_freqs(seq_len: int, dim_head: int = 64, inv_freq: Optional[Tensor] = None) -> Tensor:
    """RotaryEmbedding.forward_from_seq_len -> [1, N, dim_head], pairs interleaved."""
    if inv_freq is None:
        inv_freq = 1.0 / (10000 ** (torch.arange(0, dim_head, 2).float() / dim_head))
    ang = torch.outer(torch.arange(seq_len).float(), inv_freq.float())
    return torch.repeat_interleave(ang, 2, dim=-1)[None]


def apply_rope(t: Tensor, freqs: Tensor) -> Tensor:
    """apply_rotary_pos_emb(t[B,H,N,dh], freqs, scale=1): t*cos + rotate_half(t)*sin on pairs (2i, 2i+1)."""
    rot = freqs.shape[-1]
    tr, tp = t[..., :rot], t[..., rot:]
    x = tr.reshape(*tr.shape[:-1], rot // 2, 2)
    rh = torch.stack((-x[..., 1], x[..., 0]), dim=-1).reshape(tr.shape)
    out = tr * freqs.cos() + rh * freqs.sin()
    return torch.cat((out, tp), dim=-1).to(t.dtype)


# --------------------------------------------------------------------------
# K7-K12 DiT block                              (model/modules.py:301-641)
# --------------------------------------------------------------------------

def rms_norm(x: Tensor, w: Tensor, eps: float = 1e-6) -> Tensor:
    """RMSNorm.forward, modules.py:275-294 (qk_norm='rms_norm' only)."""
    return x * torch.rsqrt(x.float().pow(2).mean(-1, keepdim=True) + eps) * w


def attention(sd: State, p: str, x: Tensor, heads: int, mask: Optional[Tensor], freqs: Tensor,
              pe_attn_head: Optional[int]) -> Tensor:
    """AttnProcessor.__call__, modules.py:442-503."""
    b, n, _ = x.shape
    q = F.linear(x, sd[p + "to_q.weight"], sd[p + "to_q.bias"])
    k = F.linear(x, sd[p + "to_k.weight"], sd[p + "to_k.bias"])
    v = F.linear(x, sd[p + "to_v.weight"], sd[p + "to_v.bias"])
    dh = q.shape[-1] // heads
    q = q.view(b, n, heads, dh).transpose(1, 2)
    k = k.view(b, n, heads, dh).transpose(1, 2)
    v = v.view(b, n, heads, dh).transpose(1, 2)
    if (p + "q_norm.weight") in sd:
        q = rms_norm(q, sd[p + "q_norm.weight"])
        k = rms_norm(k, sd[p + "k_norm.weight"])
    if pe_attn_head is not None:
        q = torch.cat((apply_rope(q[:, :pe_attn_head], freqs), q[:, pe_attn_head:]), dim=1)
        k = torch.cat((apply_rope(k[:, :pe_attn_head], freqs), k[:, pe_attn_head:]), dim=1)
    else:
        q, k = apply_rope(q, freqs), apply_rope(k, freqs)
    s = torch.matmul(q, k.transpose(-1, -2)) * (dh ** -0.5)
    if mask is not None:
        s = s.masked_fill(~mask[:, None, None, :], float("-inf"))
    o = torch.matmul(torch.softmax(s, dim=-1), v)
    o = o.transpose(1, 2).reshape(b, n, heads * dh)
    o = F.linear(o, sd[p + "to_out.0.weight"], sd[p + "to_out.0.bias"])
    if mask is not None:
        o = o.masked_fill(~mask.unsqueeze(-1), 0.0)
    return o


def dit_block(sd: State, p: str, x: Tensor, t: Tensor, heads: int, mask: Optional[Tensor], freqs: Tensor,
              pe_attn_head: Optional[int] = None) -> Tensor:
    """DiTBlock.forward, modules.py:627-641."""
    d = x.shape[-1]
    emb = F.linear(F.silu(t), sd[p + "attn_norm.linear.weight"], sd[p + "attn_norm.linear.bias"])
    shift_msa, scale_msa, gate_msa, shift_mlp, scale_mlp, gate_mlp = torch.chunk(emb, 6, dim=1)
    h = F.layer_norm(x, (d,), eps=1e-6) * (1 + scale_msa[:, None]) + shift_msa[:, None]
    a = attention(sd, p + "attn.", h, heads, mask, freqs, pe_attn_head)
    x = x + gate_msa.unsqueeze(1) * a
    h = F.layer_norm(x, (d,), eps=1e-6) * (1 + scale_mlp[:, None]) + shift_mlp[:, None]
    h = F.linear(h, sd[p + "ff.ff.0.0.weight"], sd[p + "ff.ff.0.0.bias"])
    h = F.gelu(h, approximate="tanh")
    h = F.linear(h, sd[p + "ff.ff.2.weight"], sd[p + "ff.ff.2.bias"])
    return x + gate_mlp.unsqueeze(1) * h


# --------------------------------------------------------------------------
# a3 DiT.sample                                  (backbones/dit.py:417-472)
# --------------------------------------------------------------------------

class DiTConfig:
    def __init__(self, dim=1024, depth=22, heads=16, dim_head=64, ff_mult=2, mel_dim=100, text_num_embeds=2545,
                 text_dim=512, text_mask_padding=True, qk_norm=None, conv_layers=4, pe_attn_head=None,
                 long_skip_connection=False, use_ppg=False, ppg_dim=256, conv_groups=16, ppg_heads=4):
        self.__dict__.update(locals())
        del self.__dict__["self"]


def dit_depth(sd: State) -> int:
    n = 0
    while f"transformer_blocks.{n}.attn.to_q.weight" in sd:
        n += 1
    return n


def dit_sample(sd: State, cfg: DiTConfig, x: Tensor, cond: Tensor, text: Optional[Tensor], ppg: Optional[Tensor],
               time: Tensor, drop_audio_cond: bool, drop_text: bool, drop_ppg: bool,
               mask: Optional[Tensor] = None, cache: Optional[dict] = None) -> Tensor:
    b, n = x.shape[0], x.shape[1]
    if time.ndim == 0:
        time = time.repeat(b)
    t = time_embedding(sd, time)
    key = "uncond" if drop_text else "cond"
    if cache is not None and key in cache:
        text_emb = cache[key]
    else:
        text_emb = text_embedding(sd, text, b, n, drop_text, mask_padding=cfg.text_mask_padding)
        if cache is not None:
            cache[key] = text_emb
    ppg_emb = ppg_embedding(sd, ppg, b, n, drop_ppg, heads=getattr(cfg, "ppg_heads", 4)) if cfg.use_ppg else None
    h = input_embedding(sd, x, cond, text_emb, ppg_emb, drop_audio_cond)
    inv = sd.get("rotary_embed.inv_freq")
    freqs = rope_freqs(n, cfg.dim_head, inv)
    res = h
    for i in range(dit_depth(sd)):
        h = dit_block(sd, f"transformer_blocks.{i}.", h, t, cfg.heads, mask, freqs, cfg.pe_attn_head)
    if "long_skip_connection.weight" in sd:
        h = F.linear(torch.cat((h, res), dim=-1), sd["long_skip_connection.weight"])
    emb = F.linear(F.silu(t), sd["norm_out.linear.weight"], sd["norm_out.linear.bias"])
    scale, shift = torch.chunk(emb, 2, dim=1)  # (scale, shift) order: modules.py:333
    h = F.layer_norm(h, (h.shape[-1],), eps=1e-6) * (1 + scale)[:, None, :] + shift[:, None, :]
    return F.linear(h, sd["proj_out.weight"], sd["proj_out.bias"])


# --------------------------------------------------------------------------
# UNetT.forward                           (backbones/unett.py:184-250; f4 row)
# --------------------------------------------------------------------------

def x_rmsnorm(x: Tensor, g: Tensor) -> Tensor:
    """x_transformers.RMSNorm (UNPINNED third party): F.normalize(x, dim=-1) * sqrt(dim) * g."""
    return F.normalize(x, dim=-1) * (x.shape[-1] ** 0.5) * g


def unett_forward(sd: State, heads: int, x: Tensor, cond: Tensor, text: Tensor, time: Tensor, drop_audio_cond: bool,
                  drop_text: bool, mask: Optional[Tensor] = None, skip_connect_type: str = "concat",
                  text_mask_padding: bool = True, pe_attn_head: Optional[int] = None) -> Tensor:
    b, n = x.shape[0], x.shape[1]
    if time.ndim == 0:
        time = time.repeat(b)
    t = time_embedding(sd, time)
    text_emb = text_embedding(sd, text, b, n, drop_text, mask_padding=text_mask_padding)
    h = input_embedding(sd, x, cond, text_emb, None, drop_audio_cond)
    h = torch.cat([t.unsqueeze(1), h], dim=1)                       # time token in front (unett.py:215)
    if mask is not None:
        mask = F.pad(mask, (1, 0), value=True)
    dh = sd["layers.0.2.to_q.weight"].shape[0] // heads
    freqs = rope_freqs(n + 1, dh, sd.get("rotary_embed.inv_freq"))
    depth = 0
    while f"layers.{depth}.2.to_q.weight" in sd:
        depth += 1
    skips = []
    for idx in range(depth):
        layer = idx + 1
        p = f"layers.{idx}."
        if layer <= depth // 2:
            skips.append(h)
        else:
            skip = skips.pop()
            if skip_connect_type == "concat":
                h = F.linear(torch.cat((h, skip), dim=-1), sd[p + "0.weight"])
            elif skip_connect_type == "add":
                h = h + skip
        h = attention(sd, p + "2.", x_rmsnorm(h, sd[p + "1.g"]), heads, mask, freqs, pe_attn_head) + h
        f = x_rmsnorm(h, sd[p + "3.g"])
        f = F.linear(f, sd[p + "4.ff.0.0.weight"], sd[p + "4.ff.0.0.bias"])
        f = F.gelu(f, approximate="tanh")
        f = F.linear(f, sd[p + "4.ff.2.weight"], sd[p + "4.ff.2.bias"])
        h = f + h
    h = x_rmsnorm(h, sd["norm_out.g"])[:, 1:, :]
    return F.linear(h, sd["proj_out.weight"], sd["proj_out.bias"])


# --------------------------------------------------------------------------
# a19 fixed-grid ODE solvers              (torchdiffeq, UNPINNED; App C2)
# --------------------------------------------------------------------------

def odeint_fixed(fn, y0: Tensor, t: Tensor, method: str = "euler") -> Tensor:
    ys = [y0]
    y = y0
    for i in range(t.shape[0] - 1):
        t0, t1 = t[i], t[i + 1]
        dt = t1 - t0
        if method == "euler":
            y = y + dt * fn(t0, y)
        elif method == "midpoint":
            half = 0.5 * dt
            y_mid = y + fn(t0, y) * half
            y = y + dt * fn(t0 + half, y_mid)
        else:
            raise ValueError(method)
        ys.append(y)
    return torch.stack(ys, dim=0)


# --------------------------------------------------------------------------
# a1/a2 CFM.sample / sample_tts / sample_vc         (model/cfm.py:94-482)
# --------------------------------------------------------------------------

def sway_time_grid(steps: int, sway: Optional[float], t_start: float = 0.0, dtype=torch.float32) -> Tensor:
    """cfm.py:467-469."""
    t = torch.linspace(t_start, 1, steps + 1, dtype=dtype)
    if sway is not None:
        t = t + sway * (torch.cos(torch.pi / 2 * t) - 1 + t)
    return t


def sample_prep(cond: Tensor, text: Optional[Tensor], duration, lens: Optional[Tensor], seed: Optional[int],
                max_duration: int = 4096, no_ref_audio: bool = False, edit_mask: Optional[Tensor] = None,
                num_channels: int = 100, duplicate_test: bool = False, t_inter: float = 0.1):
    """cfm.py:378-428,452-457: masks, padding, duration clamp, seeded noise (CPU generator)."""
    b, nc = cond.shape[:2]
    if lens is None:
        lens = torch.full((b,), nc, dtype=torch.long)
    cond_mask = lens_to_mask(lens)
    if edit_mask is not None:
        cond_mask = cond_mask & edit_mask
    if isinstance(duration, int):
        duration = torch.full((b,), duration, dtype=torch.long)
    if text is not None:
        duration = torch.maximum(torch.maximum((text != -1).sum(dim=-1), lens) + 1, duration)
    else:
        duration = torch.maximum(lens + 1, duration)
    duration = duration.clamp(max=max_duration)
    n = int(duration.amax())
    test_cond = F.pad(cond, (0, 0, nc, n - 2 * nc), value=0.0) if duplicate_test else None  # cfm.py:411-412
    cond = F.pad(cond, (0, 0, 0, n - nc), value=0.0)
    if no_ref_audio:
        cond = torch.zeros_like(cond)
    cond_mask = F.pad(cond_mask, (0, n - cond_mask.shape[-1]), value=False).unsqueeze(-1)
    step_cond = torch.where(cond_mask, cond, torch.zeros_like(cond))
    mask = lens_to_mask(duration) if b > 1 else None
    y0 = torch.zeros((b, n, num_channels), dtype=step_cond.dtype)
    for i, dur in enumerate(duration.tolist()):
        if seed is not None:
            torch.manual_seed(seed)
        y0[i, :dur] = torch.randn(dur, num_channels, dtype=step_cond.dtype)
    t_start = 0.0
    if duplicate_test:  # cfm.py:462-465
        t_start = t_inter
        y0 = (1 - t_start) * y0 + t_start * test_cond
    return dict(cond=cond, cond_mask=cond_mask, step_cond=step_cond, mask=mask, y0=y0, duration=duration, n=n,
                t_start=t_start)


def cfm_sample(sd: State, cfg: DiTConfig, cond: Tensor, text: Optional[Tensor], ppg: Optional[Tensor] = None,
               duration=None, *, lens: Optional[Tensor] = None, steps: int = 32, cfg_strength: float = 1.0,
               sway_sampling_coef: Optional[float] = None, seed: Optional[int] = None, max_duration: int = 4096,
               no_ref_audio: bool = False, edit_mask: Optional[Tensor] = None, method: str = "euler",
               mode: str = "cfg", alpha_a: float = 1.0, alpha_b: float = 1.0, duplicate_test: bool = False,
               t_inter: float = 0.1) -> Tuple[Tensor, Tensor]:
    """CFM.sample (mode='cfg', cfm.py:349-482), sample_tts (mode='tts', :94-223), sample_vc (mode='vc', :226-346).

    ``cond`` is a mel [B, Nc, 100] or a raw wave [B, nw].  Returns (out, trajectory).
    For mode 'tts'/'vc', alpha_a = alpha_spk and alpha_b = alpha_txt / alpha_ppg.
    """
    if cond.ndim == 2:
        cond = log_mel_spectrogram(cond).permute(0, 2, 1)
    cond = cond.float()
    prep = sample_prep(cond, text if mode != "vc" else None, duration, lens, seed, max_duration, no_ref_audio,
                       edit_mask, cond.shape[-1], duplicate_test, t_inter)
    if duplicate_test:
        steps = int(steps * (1 - prep["t_start"]))
    step_cond, mask = prep["step_cond"], prep["mask"]
    cache: dict = {}

    def net(x, t, da, dt_, dp, txt, pg):
        return dit_sample(sd, cfg, x, step_cond, txt, pg, t, da, dt_, dp, mask,
                          cache if mode == "cfg" else None)

    if mode == "cfg":
        def fn(t, x):
            pred = net(x, t, False, False, False, text, ppg)
            if cfg_strength < 1e-5:
                return pred
            null = net(x, t, True, True, True, text, ppg)
            return pred + (pred - null) * cfg_strength
    elif mode == "tts":
        def fn(t, x):  # cfm.py:170-188
            null = net(x, t, True, True, True, text, None)
            txt = net(x, t, True, False, True, text, None)
            spk = net(x, t, False, False, True, text, None)
            return alpha_a * (spk - txt) + alpha_b * (txt - null) + null
    elif mode == "vc":
        def fn(t, x):  # cfm.py:293-311
            null = net(x, t, True, True, True, None, ppg)
            pg = net(x, t, True, True, False, None, ppg)
            spk = net(x, t, False, True, False, None, ppg)
            return alpha_a * (spk - pg) + alpha_b * (pg - null) + null
    else:
        raise ValueError(mode)

    t = sway_time_grid(steps, sway_sampling_coef, prep["t_start"])
    traj = odeint_fixed(fn, prep["y0"], t, method)
    out = torch.where(prep["cond_mask"], prep["cond"], traj[-1])
    return out, traj


# --------------------------------------------------------------------------
# K16 Vocos decode                      (vocos pkg, UNPINNED)
# --------------------------------------------------------------------------

def vocos_backbone(vs: State, mel: Tensor) -> Tensor:
    """VocosBackbone.forward: mel [B,100,T] -> [B,T,512]."""
    h = F.conv1d(mel, vs["backbone.embed.weight"], vs["backbone.embed.bias"], padding=3)
    c = h.shape[1]
    h = F.layer_norm(h.transpose(1, 2), (c,), vs["backbone.norm.weight"], vs["backbone.norm.bias"], eps=1e-6)
    i = 0
    while f"backbone.convnext.{i}.dwconv.weight" in vs:
        p = f"backbone.convnext.{i}."
        r = h
        g = F.conv1d(h.transpose(1, 2), vs[p + "dwconv.weight"], vs[p + "dwconv.bias"], padding=3, groups=c)
        g = F.layer_norm(g.transpose(1, 2), (c,), vs[p + "norm.weight"], vs[p + "norm.bias"], eps=1e-6)
        g = F.linear(g, vs[p + "pwconv1.weight"], vs[p + "pwconv1.bias"])
        g = F.gelu(g)
        g = F.linear(g, vs[p + "pwconv2.weight"], vs[p + "pwconv2.bias"])
        h = r + vs[p + "gamma"] * g
        i += 1
    return F.layer_norm(h, (c,), vs["backbone.final_layer_norm.weight"], vs["backbone.final_layer_norm.bias"],
                        eps=1e-6)


def istft_head(vs: State, h: Tensor, n_fft: int = 1024, hop: int = 256) -> Tensor:
    """ISTFTHead.forward (padding='center'); restated in-tree at
    runtime/triton_trtllm/scripts/export_vocoder_to_onnx.py:45-59."""
    z = F.linear(h, vs["head.out.weight"], vs["head.out.bias"]).transpose(1, 2)
    mag, ph = z.chunk(2, dim=1)
    mag = torch.clip(torch.exp(mag), max=1e2)
    spec = torch.complex(mag * torch.cos(ph), mag * torch.sin(ph))
    window = torch.hann_window(n_fft, periodic=True, dtype=torch.float32)
    return torch.istft(spec, n_fft, hop, n_fft, window, center=True)


def vocos_decode(vs: State, mel: Tensor) -> Tensor:
    """vocoder.decode(mel[B,100,T]) -> wav [B, 256*(T-1)] (utils_infer.py:489)."""
    return istft_head(vs, vocos_backbone(vs, mel.float()))


# --------------------------------------------------------------------------
# K18 GumbelVectorQuantizer eval forward (parity-only; modules.py:881-950)
# --------------------------------------------------------------------------

def gumbel_vq_eval(x: Tensor, layers, codebook: Tensor, groups: int, num_vars: int,
                   combine_groups: bool = False) -> Dict[str, Tensor]:
    """GumbelVectorQuantizer.forward, eval branch (modules.py:881-950), time_first input [B, T, C].

    ``layers``: [(weight, bias, gelu_after)] of weight_proj; ``codebook``: vars [1, (1 or groups)*num_vars, var_dim].
    Returns x, targets, code_perplexity, prob_perplexity.
    """
    bsz, tsz, fsz = x.shape
    h = x.reshape(-1, fsz)
    for w, b, gelu in layers:
        h = F.linear(h, w, b)
        if gelu:
            h = F.gelu(h)
    logits = h.view(bsz * tsz * groups, num_vars)
    k = logits.max(-1)[1]
    hard = torch.zeros_like(logits).scatter_(-1, k.view(-1, 1), 1.0).view(bsz * tsz, groups, num_vars)
    hard_probs = hard.float().mean(dim=0)
    code_ppl = torch.exp(-torch.sum(hard_probs * torch.log(hard_probs + 1e-7), dim=-1)).sum()
    avg_probs = torch.softmax(logits.view(bsz * tsz, groups, num_vars).float(), dim=-1).mean(dim=0)
    prob_ppl = torch.exp(-torch.sum(avg_probs * torch.log(avg_probs + 1e-7), dim=-1)).sum()
    vars_ = codebook.repeat(1, groups, 1) if combine_groups else codebook
    q = (hard.view(bsz * tsz, -1).unsqueeze(-1) * vars_).view(bsz * tsz, groups, num_vars, -1).sum(-2)
    return dict(x=q.view(bsz, tsz, -1), targets=k.view(bsz, tsz, groups), code_perplexity=code_ppl,
                prob_perplexity=prob_ppl)
