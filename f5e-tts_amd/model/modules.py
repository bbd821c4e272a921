"""Host-side mirrors of the reference building blocks (reference model/modules.py): same class names, constructor
signatures and ``state_dict`` keys, so upstream checkpoints load with ``strict=True``.  The nn.Linear / nn.Conv1d
members are *parameter containers only*; ``forward`` never calls them -- it runs the HIP kernels through
``f5e_tts_amd.ops`` and raises if the tensors are not on a gfx950 device (no eager fallback)."""
from __future__ import annotations

from typing import Optional

import torch
from torch import nn

from .. import _C, ops
from ..engine import fft_tables, mel_filterbank

BF, F32, I32 = torch.bfloat16, torch.float32, torch.int32


# ---------------------------------------------------------------- mel front-end (reference modules.py:75-143)

class MelSpec(nn.Module):
    def __init__(self, n_fft=1024, hop_length=256, win_length=1024, n_mel_channels=100, target_sample_rate=24_000,
                 mel_spec_type="vocos"):
        super().__init__()
        assert mel_spec_type in ["vocos", "bigvgan"], "We only support two extract mel backend: vocos or bigvgan"
        if mel_spec_type != "vocos":
            raise _C.F5EError("only the vocos mel front-end is built for MI355X (bigvgan needs librosa's slaney filters)")
        if n_fft != 1024 or win_length != 1024:
            raise _C.F5EError("the STFT kernel is built for n_fft = win_length = 1024")
        self.n_fft, self.hop_length, self.win_length = n_fft, hop_length, win_length
        self.n_mel_channels, self.target_sample_rate = n_mel_channels, target_sample_rate
        self.mel_spec_type = mel_spec_type
        self.register_buffer("dummy", torch.tensor(0), persistent=False)
        self._tables = None

    def _get_tables(self, device):
        if self._tables is None or self._tables[0].device != device:
            win, tw = fft_tables(device)
            fb = mel_filterbank(self.n_fft // 2 + 1, self.n_mel_channels, self.target_sample_rate).to(device)
            self._tables = (win, tw, fb, ops.band_filterbank(fb))
        return self._tables

    def forward(self, wav: torch.Tensor) -> torch.Tensor:
        """wav [b, nw] (or [b, 1, nw]) on the GPU -> log-mel [b, n_mels, 1 + nw // hop]."""
        if wav.ndim == 3:
            wav = wav.squeeze(1)
        assert wav.ndim == 2
        if self.dummy.device != wav.device:
            self.to(wav.device)
        win, tw, fb, banded = self._get_tables(wav.device)
        wav = wav.to(F32).contiguous()
        frames = 1 + wav.shape[1] // self.hop_length
        out = torch.empty(wav.shape[0], frames, self.n_mel_channels, device=wav.device)
        if banded is not None:   # same result bit for bit; the filters' non-zero runs from LDS instead of 513 dense rows
            ops.stft_logmel_banded(wav, win, tw, banded[0], banded[1], out, self.n_fft, self.hop_length)
        else:
            ops.stft_logmel(wav, win, tw, fb, out, self.n_fft, self.hop_length)
        return out.permute(0, 2, 1)


# ---------------------------------------------------------------- parameter containers

class SinusPositionEmbedding(nn.Module):
    def __init__(self, dim):
        super().__init__()
        self.dim = dim


class TimestepEmbedding(nn.Module):
    """reference modules.py:721-731 (keys time_mlp.0 / time_mlp.2)."""

    def __init__(self, dim, freq_embed_dim=256):
        super().__init__()
        self.time_embed = SinusPositionEmbedding(freq_embed_dim)
        self.time_mlp = nn.Sequential(nn.Linear(freq_embed_dim, dim), nn.SiLU(), nn.Linear(dim, dim))


class ConvPositionEmbedding(nn.Module):
    """reference modules.py:167-176 (keys conv1d.0 / conv1d.2)."""

    def __init__(self, dim, kernel_size=31, groups=16):
        super().__init__()
        assert kernel_size % 2 != 0
        self.conv1d = nn.Sequential(
            nn.Conv1d(dim, dim, kernel_size, groups=groups, padding=kernel_size // 2), nn.Mish(),
            nn.Conv1d(dim, dim, kernel_size, groups=groups, padding=kernel_size // 2), nn.Mish())


class GRN(nn.Module):
    def __init__(self, dim):
        super().__init__()
        self.gamma = nn.Parameter(torch.zeros(1, 1, dim))
        self.beta = nn.Parameter(torch.zeros(1, 1, dim))


class ConvNeXtV2Block(nn.Module):
    """reference modules.py:241-257."""

    def __init__(self, dim: int, intermediate_dim: int, dilation: int = 1):
        super().__init__()
        if dilation != 1:
            raise _C.F5EError("dwconv kernel is built for dilation 1")
        self.dwconv = nn.Conv1d(dim, dim, kernel_size=7, padding=3, groups=dim)
        self.norm = nn.LayerNorm(dim, eps=1e-6)
        self.pwconv1 = nn.Linear(dim, intermediate_dim)
        self.act = nn.GELU()
        self.grn = GRN(intermediate_dim)
        self.pwconv2 = nn.Linear(intermediate_dim, dim)


class RMSNorm(nn.Module):
    def __init__(self, dim: int, eps: float):
        super().__init__()
        self.eps = eps
        self.weight = nn.Parameter(torch.ones(dim))


class AdaLayerNorm(nn.Module):
    def __init__(self, dim):
        super().__init__()
        self.silu = nn.SiLU()
        self.linear = nn.Linear(dim, dim * 6)
        self.norm = nn.LayerNorm(dim, elementwise_affine=False, eps=1e-6)


class AdaLayerNorm_Final(nn.Module):
    def __init__(self, dim):
        super().__init__()
        self.silu = nn.SiLU()
        self.linear = nn.Linear(dim, dim * 2)
        self.norm = nn.LayerNorm(dim, elementwise_affine=False, eps=1e-6)


class FeedForward(nn.Module):
    """reference modules.py:342-350 (keys ff.0.0 / ff.2)."""

    def __init__(self, dim, dim_out=None, mult=4, dropout=0.0, approximate: str = "none"):
        super().__init__()
        inner_dim = int(dim * mult)
        dim_out = dim_out if dim_out is not None else dim
        project_in = nn.Sequential(nn.Linear(dim, inner_dim), nn.GELU(approximate=approximate))
        self.ff = nn.Sequential(project_in, nn.Dropout(dropout), nn.Linear(inner_dim, dim_out))


class AttnProcessor:
    def __init__(self, pe_attn_head: Optional[int] = None):
        self.pe_attn_head = pe_attn_head


class Attention(nn.Module):
    """reference modules.py:360-416, self-attention members only (keys to_q/to_k/to_v/to_out.0[/q_norm/k_norm])."""

    def __init__(self, processor, dim: int, heads: int = 8, dim_head: int = 64, dropout: float = 0.0,
                 context_dim: Optional[int] = None, context_pre_only: bool = False, qk_norm: Optional[str] = None):
        super().__init__()
        if context_dim is not None:
            raise _C.F5EError("joint (MMDiT) attention is not on the sampled path (SURVEY F4) and is not built")
        self.processor = processor
        self.dim, self.heads, self.inner_dim, self.dropout = dim, heads, dim_head * heads, dropout
        self.to_q = nn.Linear(dim, self.inner_dim)
        self.to_k = nn.Linear(dim, self.inner_dim)
        self.to_v = nn.Linear(dim, self.inner_dim)
        if qk_norm is None:
            self.q_norm = None
            self.k_norm = None
        elif qk_norm == "rms_norm":
            self.q_norm = RMSNorm(dim_head, eps=1e-6)
            self.k_norm = RMSNorm(dim_head, eps=1e-6)
        else:
            raise ValueError(f"Unimplemented qk_norm: {qk_norm}")
        self.to_out = nn.ModuleList([nn.Linear(self.inner_dim, dim), nn.Dropout(dropout)])


# ---------------------------------------------------------------- DiT block (reference modules.py:610-641)

class DiTBlock(nn.Module):
    def __init__(self, dim, heads, dim_head, ff_mult=4, dropout=0.1, qk_norm=None, pe_attn_head=None):
        super().__init__()
        self.dim, self.heads, self.dim_head = dim, heads, dim_head
        self.attn_norm = AdaLayerNorm(dim)
        self.attn = Attention(processor=AttnProcessor(pe_attn_head=pe_attn_head), dim=dim, heads=heads,
                              dim_head=dim_head, dropout=dropout, qk_norm=qk_norm)
        self.ff_norm = nn.LayerNorm(dim, elementwise_affine=False, eps=1e-6)
        self.ff = FeedForward(dim=dim, mult=ff_mult, dropout=dropout, approximate="tanh")
        self._packed = None

    def _pack(self, device):
        sig = tuple((p.data_ptr(), p._version) for p in self.parameters()) + (str(device),)
        if self._packed is None or self._packed[0] != sig:
            a = self.attn
            f32 = lambda t: t.detach().to(device, F32).contiguous()  # noqa: E731
            w_qkv = torch.cat([f32(a.to_q.weight), f32(a.to_k.weight), f32(a.to_v.weight)], 0).to(BF)
            b_qkv = torch.cat([f32(a.to_q.bias), f32(a.to_k.bias), f32(a.to_v.bias)], 0)
            self._packed = (sig, dict(
                w_ada=f32(self.attn_norm.linear.weight), b_ada=f32(self.attn_norm.linear.bias),
                w_qkv=w_qkv, b_qkv=b_qkv, w_out=f32(a.to_out[0].weight).to(BF), b_out=f32(a.to_out[0].bias),
                w_ff1=f32(self.ff.ff[0][0].weight).to(BF), b_ff1=f32(self.ff.ff[0][0].bias),
                w_ff2=f32(self.ff.ff[2].weight).to(BF), b_ff2=f32(self.ff.ff[2].bias)))
        return self._packed[1]

    @torch.no_grad()
    def forward(self, x, t, mask=None, rope=None):
        """x [b, n, d], t [b, d] time embedding, mask bool [b, n] or None, rope = (freqs [1, n, 64], scale)."""
        if self.dim_head != 64:
            raise _C.F5EError("DiTBlock HIP path: dim_head must be 64")
        ops.require_device()
        B, N, D = x.shape
        dv = x.device
        w = self._pack(dv)
        H = self.heads
        emb = torch.empty(B, 6 * D, device=dv)
        ops.gemm_f32(t.to(F32).contiguous(), w["w_ada"], w["b_ada"], out=emb, a_act=ops.ACT_SILU)
        xs = x.to(F32).reshape(B * N, D).clone()
        hn = torch.empty(B * N, D, device=dv, dtype=BF)
        ops.layernorm(xs, hn, scale=emb[:, D:2 * D], shift=emb[:, 0:D], rows_per_seq=N)
        n_pad = (N + 63) // 64 * 64
        q = torch.zeros(B, H, n_pad, 64, device=dv, dtype=BF)
        k = torch.zeros_like(q)
        vt = torch.zeros(B, H, 64, n_pad, device=dv, dtype=BF)
        cs = torch.zeros(N, 32, 2, device=dv)
        pe = self.attn.processor.pe_attn_head
        rope_heads = 0
        if rope is not None:
            freqs = rope[0]
            rope_heads = H if pe is None else pe
            if N > 1:
                ops.rope_table(freqs[0, 1, 0::2].to(F32).contiguous(), cs)  # angle at position 1 = inv_freq
            else:
                cs[..., 0] = 1.0
        qn = self.attn.q_norm.weight.detach().to(dv, F32).contiguous() if self.attn.q_norm is not None else None
        kn = self.attn.k_norm.weight.detach().to(dv, F32).contiguous() if self.attn.k_norm is not None else None
        ops.gemm_bf16_qkv_rope(hn, w["w_qkv"], w["b_qkv"], q, k, vt, H, rope_heads, cs, N, q_norm_w=qn, k_norm_w=kn)
        lens = mask.sum(-1).to(I32).contiguous() if mask is not None else None
        if mask is not None and not torch.equal(mask, torch.arange(N, device=dv)[None] < lens[:, None]):
            raise _C.F5EError("attention kernel takes key-padding masks of the lens_to_mask form only")
        ao = torch.empty(B * N, H * 64, device=dv, dtype=BF)
        ops.flash_attn(q, k, vt, ao, N, kv_len=lens)
        ops.gemm_bf16_gate_residual(ao, w["w_out"], w["b_out"], xs, emb[:, 2 * D:3 * D], N, seq_len=lens)
        ops.layernorm(xs, hn, scale=emb[:, 4 * D:5 * D], shift=emb[:, 3 * D:4 * D], rows_per_seq=N)
        ff = torch.empty(B * N, w["w_ff1"].shape[0], device=dv, dtype=BF)
        ops.gemm_bf16_bias(hn, w["w_ff1"], w["b_ff1"], ff, act=ops.ACT_GELU_TANH)
        ops.gemm_bf16_gate_residual(ff, w["w_ff2"], w["b_ff2"], xs, emb[:, 5 * D:6 * D], N)
        return xs.view(B, N, D)


# ---------------------------------------------------------------- codebook (reference modules.py:744-950, eval only)

class GumbelVectorQuantizer(nn.Module):
    """Constructor / state_dict mirror of the reference quantizer; ``forward`` is the EVAL branch only (per-group
    argmax -> codebook row) on libf5e_hip.so.  The reference never calls it from ``DiT.sample``: this is
    the parity-only op BASELINE config 5 names."""

    def __init__(self, dim, num_vars, temp, groups, combine_groups, vq_dim, time_first, activation=nn.GELU(),
                 weight_proj_depth=1, weight_proj_factor=1, hard=True, std=0):
        super().__init__()
        self.groups, self.combine_groups, self.input_dim, self.num_vars = groups, combine_groups, dim, num_vars
        self.time_first, self.hard = time_first, hard
        assert vq_dim % groups == 0, f"dim {vq_dim} must be divisible by groups {groups} for concatenation"
        var_dim = vq_dim // groups
        num_groups = groups if not combine_groups else 1
        self.vars = nn.Parameter(torch.empty(1, num_groups * num_vars, var_dim))
        nn.init.uniform_(self.vars) if std == 0 else nn.init.normal_(self.vars, mean=0, std=std)
        if weight_proj_depth > 1:
            if not isinstance(activation, nn.GELU):
                raise _C.F5EError("weight_proj activation other than GELU is not built")
            inner = self.input_dim * weight_proj_factor
            blocks = [nn.Sequential(nn.Linear(self.input_dim if i == 0 else inner, inner), activation)
                      for i in range(weight_proj_depth - 1)]
            self.weight_proj = nn.Sequential(*blocks, nn.Linear(inner, groups * num_vars))
        else:
            self.weight_proj = nn.Linear(self.input_dim, groups * num_vars)
            nn.init.normal_(self.weight_proj.weight, mean=0, std=1)
            nn.init.zeros_(self.weight_proj.bias)
        if isinstance(temp, str):
            import ast
            temp = ast.literal_eval(temp)
        assert len(temp) == 3, f"{temp}, {len(temp)}"
        self.max_temp, self.min_temp, self.temp_decay = temp
        self.curr_temp = self.max_temp

    def set_num_updates(self, num_updates):
        self.curr_temp = max(self.max_temp * self.temp_decay ** num_updates, self.min_temp)

    @torch.no_grad()
    def forward(self, x, produce_targets=False):
        if self.training:
            raise NotImplementedError("gumbel-softmax training branch is out of scope (SURVEY section 8)")
        ops.require_device()
        result = {"num_vars": self.num_vars * self.groups}
        if not self.time_first:
            x = x.transpose(1, 2)
        bsz, tsz, fsz = x.shape
        dv = x.device
        h = x.reshape(-1, fsz).to(F32).contiguous()
        layers = [self.weight_proj] if isinstance(self.weight_proj, nn.Linear) else list(self.weight_proj)
        for layer in layers:
            lin, act = (layer, ops.ACT_NONE) if isinstance(layer, nn.Linear) else (layer[0], ops.ACT_GELU_ERF)
            o = torch.empty(h.shape[0], lin.out_features, device=dv)
            ops.gemm_f32(h, lin.weight.detach().to(dv, F32).contiguous(), lin.bias.detach().to(dv, F32).contiguous(),
                         out=o, act=act)
            h = o
        vd = self.vars.shape[-1]
        out = torch.empty(bsz * tsz, self.groups * vd, device=dv)
        targets = torch.empty(bsz * tsz, self.groups, device=dv, dtype=I32)
        stats = torch.empty(2, device=dv)
        ops.vq_eval(h, self.vars.detach().to(dv, F32)[0].contiguous(), self.combine_groups, out, targets, stats,
                    self.groups, self.num_vars)
        result["code_perplexity"], result["prob_perplexity"] = stats[0], stats[1]
        result["temp"] = self.curr_temp
        if produce_targets:
            result["targets"] = targets.view(bsz, tsz, self.groups).long()
        xq = out.view(bsz, tsz, -1)
        result["x"] = xq if self.time_first else xq.transpose(1, 2)
        return result
