"""Host-side engine of the hot path: weight repack, workspace planning, once-per-call embeddings/tables and the
hipGraph-driven ODE loop.  All arithmetic is done by libf5e_hip.so (``ops``); torch is used for device memory,
integer/bool index preparation and streams only.

Mirrors, at the level of *what is computed*: DiT.sample (reference backbones/dit.py:417-472) and the loop body of
CFM.sample / sample_tts / sample_vc (model/cfm.py:430-471), restructured for MI355X:
  * the CFG branches of one step are batched into ONE S = branches*B forward (weights stream once per step);
  * everything that depends only on t (time MLP, 22 AdaLN tables, final AdaLN) is built for the whole time grid before
    the loop; everything that depends only on (cond, text, ppg) (text/PPG embeddings, their share of the input
    projection) once per call and branch -- the reference recomputes the PPG embedding every forward (dit.py:448);
  * one ODE step (network + CFG + Euler/midpoint update) is captured as a hipGraph and replayed; per-step scalars come
    from device tables indexed by a device-side evaluation counter.
"""
from __future__ import annotations

import math
import os
import threading
from dataclasses import dataclass
from typing import Dict, List, Optional, Tuple

import torch

from . import _C, ops

Tensor = torch.Tensor
BF, F32, I32 = torch.bfloat16, torch.float32, torch.int32


@dataclass
class DiTConfig:
    dim: int = 1024
    depth: int = 22
    heads: int = 16
    dim_head: int = 64
    ff_mult: int = 2
    mel_dim: int = 100
    text_num_embeds: int = 2545
    text_dim: int = 512
    text_mask_padding: bool = True
    qk_norm: Optional[str] = None
    conv_layers: int = 4
    pe_attn_head: Optional[int] = None
    long_skip_connection: bool = False
    use_ppg: bool = False
    ppg_dim: int = 256
    ppg_transformer: bool = False   # PPGEmbedding(use_transformer=True): nn.TransformerEncoder + Linear
    ppg_heads: int = 4

    def validate(self) -> None:
        if self.dim_head != 64:
            raise _C.F5EError("HIP path is built for dim_head = 64")
        if self.dim % 256 or self.text_dim % 256:
            raise _C.F5EError(f"HIP path needs dim and text_dim to be multiples of 256 (got {self.dim}, {self.text_dim})")
        if (self.dim // 16) % 16 or self.dim // 16 > 64:
            raise _C.F5EError("ConvPositionEmbedding kernel needs dim/16 in {16, 32, 48, 64} channels per group; "
                              f"got dim = {self.dim}")
        if self.qk_norm not in (None, "rms_norm"):
            raise _C.F5EError(f"unknown qk_norm {self.qk_norm!r}")
        if self.mel_dim % 4:
            raise _C.F5EError("mel_dim must be a multiple of 4")


def fft_tables(device) -> Tuple[Tensor, Tensor]:
    """hann window (periodic, as torch.hann_window) and 1024-point twiddles (cos, -sin), built in float64."""
    k = torch.arange(512, dtype=torch.float64)
    tw = torch.stack((torch.cos(2 * math.pi * k / 1024), -torch.sin(2 * math.pi * k / 1024)), -1).float()
    return torch.hann_window(1024).to(device), tw.to(device).contiguous()


def mel_filterbank(n_freqs: int, n_mels: int, sr: int) -> Tensor:
    """HTK triangular filterbank, norm=None (torchaudio.functional.melscale_fbanks) -> [n_freqs, n_mels]."""
    all_freqs = torch.linspace(0, sr // 2, n_freqs)
    m_max = 2595.0 * math.log10(1.0 + (sr // 2) / 700.0)
    m_pts = torch.linspace(0.0, m_max, n_mels + 2)
    f_pts = 700.0 * (10.0 ** (m_pts / 2595.0) - 1.0)
    f_diff = f_pts[1:] - f_pts[:-1]
    slopes = f_pts.unsqueeze(0) - all_freqs.unsqueeze(1)
    return torch.clamp(torch.min(-slopes[:, :-2] / f_diff[:-1], slopes[:, 2:] / f_diff[1:]), min=0.0).contiguous()


def text_pos_table(dim: int, end: int = 4096, theta: float = 10000.0) -> Tensor:
    """precompute_freqs_cis (model/modules.py:196-207): constant [end, dim] = cos || sin table."""
    freqs = 1.0 / (theta ** (torch.arange(0, dim, 2)[: dim // 2].float() / dim))
    ang = torch.outer(torch.arange(end), freqs).float()
    return torch.cat([ang.cos(), ang.sin()], dim=-1).contiguous()


class DiTEngine:
    """Device-resident, repacked weights of one DiT + the code that runs it through the C ABI."""

    def __init__(self, sd: Dict[str, Tensor], cfg: DiTConfig, device, blocks: bool = True):
        """blocks=False loads only the front-end shared with UNetT (time MLP, text embedding, input embedding)."""
        cfg.validate()
        ops.require_device()
        self.cfg = cfg
        self.device = torch.device(device)
        dv = self.device
        f = lambda k: sd[k].detach().to(dv, F32).contiguous()  # noqa: E731
        D, L = cfg.dim, cfg.depth
        self.L = L
        self.inner = cfg.heads * 64
        self.FF = D * cfg.ff_mult
        # --- once-per-call fp32 weights
        self.tm_w0, self.tm_b0 = f("time_embed.time_mlp.0.weight"), f("time_embed.time_mlp.0.bias")
        self.tm_w2, self.tm_b2 = f("time_embed.time_mlp.2.weight"), f("time_embed.time_mlp.2.bias")
        half = 128
        self.sinus_freqs = torch.exp(torch.arange(half).float() * -(math.log(10000) / (half - 1))).to(dv)
        self.text_table = f("text_embed.text_embed.weight")
        self.text_blocks = []
        for i in range(cfg.conv_layers):
            p = f"text_embed.text_blocks.{i}."
            self.text_blocks.append(dict(
                dw_w=f(p + "dwconv.weight")[:, 0, :].t().contiguous(), dw_b=f(p + "dwconv.bias"),
                ln_g=f(p + "norm.weight"), ln_b=f(p + "norm.bias"),
                w1=f(p + "pwconv1.weight"), b1=f(p + "pwconv1.bias"),
                grn_g=f(p + "grn.gamma").view(-1).contiguous(), grn_b=f(p + "grn.beta").view(-1).contiguous(),
                w2=f(p + "pwconv2.weight"), b2=f(p + "pwconv2.bias")))
        self.text_pos = text_pos_table(cfg.text_dim).to(dv) if cfg.conv_layers > 0 else None
        self.in_w, self.in_b = f("input_embed.proj.weight"), f("input_embed.proj.bias")
        self.ppg = None
        if cfg.use_ppg and cfg.ppg_transformer:
            pp = "ppg_embed.ppg_proj."
            layers = []
            i = 0
            while pp + f"0.layers.{i}.self_attn.in_proj_weight" in sd:
                q = pp + f"0.layers.{i}."
                layers.append({k: f(q + n) for k, n in dict(
                    w_in="self_attn.in_proj_weight", b_in="self_attn.in_proj_bias", w_o="self_attn.out_proj.weight",
                    b_o="self_attn.out_proj.bias", w1="linear1.weight", b1="linear1.bias", w2="linear2.weight",
                    b2="linear2.bias", g1="norm1.weight", be1="norm1.bias", g2="norm2.weight", be2="norm2.bias").items()})
                i += 1
            if cfg.ppg_dim % cfg.ppg_heads or (cfg.ppg_dim // cfg.ppg_heads) % 4:
                raise _C.F5EError("PPG transformer embedding: ppg_dim / nhead must be a multiple of 4")
            self.ppg = dict(layers=layers, w_out=f(pp + "1.weight"), b_out=f(pp + "1.bias"))
        elif cfg.use_ppg:
            pp = "ppg_embed.ppg_proj."
            convs = []
            for ci, bi in ((2, 3), (6, 7), (10, 11)):
                w, b = f(pp + f"{ci}.weight"), f(pp + f"{ci}.bias")
                g, be = f(pp + f"{bi}.weight"), f(pp + f"{bi}.bias")
                mu, var = f(pp + f"{bi}.running_mean"), f(pp + f"{bi}.running_var")
                s = g / torch.sqrt(var + 1e-5)  # eval-mode BatchNorm1d folded into the conv (weights-only transform)
                wf = (w * s[:, None, None]).permute(0, 2, 1).reshape(w.shape[0], -1).contiguous()  # [oc][k*ic]
                convs.append((wf, ((b - mu) * s + be).contiguous()))
            self.ppg = dict(w0=f(pp + "0.weight"), b0=f(pp + "0.bias"), convs=convs,
                            w15=f(pp + "15.weight"), b15=f(pp + "15.bias"))
        # conv position embedding: [D][D/16][31] -> [16 groups][tap][64 oc][64 ic] bf16 (zero padded)
        self.cp = []
        for j in (0, 2):
            w = f(f"input_embed.conv_pos_embed.conv1d.{j}.weight")
            self.cp.append((ops.pack_convpos_weight(w, 16), f(f"input_embed.conv_pos_embed.conv1d.{j}.bias")))
        inv = sd.get("rotary_embed.inv_freq")
        self.inv_freq = (inv.detach().float() if inv is not None
                         else 1.0 / (10000 ** (torch.arange(0, 64, 2).float() / 64))).to(dv).contiguous()
        # fold the AdaLN LayerNorms into the GEMMs in run_ode when the architecture allows it (F5E_FUSE_LN=0: A/B switch)
        self.fuse_ln = os.environ.get("F5E_FUSE_LN", "1") != "0"
        self.mall_prefetch = os.environ.get("F5E_MALL_PREFETCH", "1") != "0"  # Infinity-Cache weight prefetch at small M
        self._loops = threading.local()   # per-thread LRU of persistent loop states (see _LoopState)
        self._tables: Dict[tuple, Tensor] = {}
        self._lock = threading.Lock()
        if not blocks:
            return
        # --- per-block weights: AdaLN linears stay fp32 (tables), the four block linears go bf16
        self.adaln = []
        self.blocks_keep = []
        self.block_arr = (_C.BlockWeights * L)()
        for i in range(L):
            p = f"transformer_blocks.{i}."
            self.adaln.append((f(p + "attn_norm.linear.weight"), f(p + "attn_norm.linear.bias")))
            w_qkv = torch.cat([f(p + "attn.to_q.weight"), f(p + "attn.to_k.weight"), f(p + "attn.to_v.weight")], 0).to(BF)
            b_qkv = torch.cat([f(p + "attn.to_q.bias"), f(p + "attn.to_k.bias"), f(p + "attn.to_v.bias")], 0)
            w_out, b_out = f(p + "attn.to_out.0.weight").to(BF), f(p + "attn.to_out.0.bias")
            w_ff1, b_ff1 = f(p + "ff.ff.0.0.weight").to(BF), f(p + "ff.ff.0.0.bias")
            w_ff2, b_ff2 = f(p + "ff.ff.2.weight").to(BF), f(p + "ff.ff.2.bias")
            keep = [w_qkv, b_qkv, w_out, b_out, w_ff1, b_ff1, w_ff2, b_ff2]
            names = ["w_qkv", "b_qkv", "w_out", "b_out", "w_ff1", "b_ff1", "w_ff2", "b_ff2"]
            if cfg.qk_norm == "rms_norm":
                keep += [f(p + "attn.q_norm.weight"), f(p + "attn.k_norm.weight")]
                names += ["q_norm_w", "k_norm_w"]
            self.blocks_keep.append(keep)
            for name, t in zip(names, keep):
                setattr(self.block_arr[i], name, t.data_ptr())
        self.skip_w = f("long_skip_connection.weight") if cfg.long_skip_connection else None
        self.final_w, self.final_b = f("norm_out.linear.weight"), f("norm_out.linear.bias")
        self.proj_w, self.proj_b = f("proj_out.weight").to(BF), f("proj_out.bias")
        self.row_stride = L * 6 * D + 2 * D
        self.rope_heads = cfg.heads if cfg.pe_attn_head is None else cfg.pe_attn_head

    # ------------------------------------------------------------------ once-per-call pieces

    # Above this many rows per launch the 128x128 GEMM tiles win (QKV / FF1 at M = 3752: 41 / 23 us against 54 / 31 us for
    # the 64x64 tile the fusion needs, i.e. more than the two LayerNorm launches it saves): keep LayerNorm separate there.
    # (A large-M form of the fusion on the 256x256 kernel was built and measured twice in round 2: break-even at C3 -- the
    # fused form moves as many bytes as the roofline-bound LayerNorm it replaces, in less favourable patterns)
    LN_FUSE_MAX_ROWS = 2800

    def fuse_rows_ok(self, M: int) -> bool:
        return M <= self.LN_FUSE_MAX_ROWS

    @property
    def can_fuse_ln(self) -> bool:
        cfg = self.cfg
        return (cfg.qk_norm is None and not cfg.long_skip_connection and cfg.dim % 256 == 0 and cfg.dim <= 1024
                and self.inner == cfg.dim)

    def cd_tables(self, mod: Tensor) -> Tensor:
        """Tables of the fused AdaLN (f5e_ln_fuse): for every evaluation row of `mod` and every linear that follows a
        modulated LayerNorm,  c[n] = sum_k W[n][k] (1 + scale[k])  and  d[n] = sum_k W[n][k] shift[k] + bias[n],  with
        W the bf16 weights the MFMA kernels use.  Layout per row: L x (c_qkv | d_qkv | c_ff1 | d_ff1), c_proj | d_proj."""
        cfg, dv = self.cfg, self.device
        D, L, FF, inner, mel = cfg.dim, self.L, self.FF, self.inner, cfg.mel_dim
        E, rows, _ = mod.shape
        m2 = mod.view(E * rows, self.row_stride)
        ls = 6 * inner + 2 * FF
        stride = (L * ls + 2 * mel + 3) // 4 * 4
        cd = torch.zeros(E * rows, stride, device=dv)
        one_plus = torch.empty(E * rows, D, device=dv)

        def pair(scale, shift, w_bf, bias, off, n):
            w = ops.cast_f32(w_bf, torch.empty(w_bf.shape, device=dv))     # the bf16 values the MFMA kernels read
            ops.axpby(scale.contiguous(), None, one_plus, 1.0, 0.0, 1.0)
            ops.gemm_f32(one_plus, w, None, out=cd[:, off:off + n])
            ops.gemm_f32(shift, w, bias, out=cd[:, off + n:off + 2 * n])

        for l in range(L):
            w_qkv, b_qkv, _, _, w_ff1, b_ff1 = self.blocks_keep[l][:6]
            mb = m2[:, l * 6 * D:(l + 1) * 6 * D]
            pair(mb[:, D:2 * D], mb[:, :D], w_qkv, b_qkv, l * ls, 3 * inner)
            pair(mb[:, 4 * D:5 * D], mb[:, 3 * D:4 * D], w_ff1, b_ff1, l * ls + 6 * inner, FF)
        mf = m2[:, L * 6 * D:]
        pair(mf[:, :D], mf[:, D:2 * D], self.proj_w, self.proj_b, L * ls, mel)
        return cd.view(E, rows, stride)

    def time_tables_cached(self, t_host: Tensor, coef_host: Optional[Tensor] = None):
        """(mod, cd, coef) tables: they depend on the time grid only, so calls that share (steps, sway, solver) share
        them.  The cache is keyed by the exact grid values and is read-only once built.  Every entry carries the event
        recorded behind its build kernels: a caller on ANOTHER stream (a second thread sampling inside
        `with torch.cuda.stream(s)`) orders its stream behind that event before it reads the tables -- without it the
        hit would race the build still queued on the first caller's stream.  cd is None when the architecture rules the
        fused AdaLN out; coef is the device copy of `coef_host` (step coefficients of the solver)."""
        key = tuple(t_host.tolist()) + (("c",) + tuple(coef_host.tolist()) if coef_host is not None else ())
        with self._lock:
            hit = self._tables.get(key)
        if hit is None:
            mod = self.time_tables(h2d(t_host, self.device))
            cd = self.cd_tables(mod) if self.can_fuse_ln else None
            coef = h2d(coef_host, self.device).contiguous() if coef_host is not None else None
            ev = torch.cuda.Event()
            ev.record(torch.cuda.current_stream(self.device))
            hit = (mod, cd, coef, ev)
            with self._lock:
                if len(self._tables) >= 8:
                    self._tables.pop(next(iter(self._tables)))
                self._tables[key] = hit
        else:
            torch.cuda.current_stream(self.device).wait_event(hit[3])
        return hit[:3]

    def time_embed(self, flat_t: Tensor) -> Tensor:
        """TimestepEmbedding (modules.py:721-731): f32 [n] -> f32 [n, dim]."""
        cfg, dv = self.cfg, self.device
        n = flat_t.shape[0]
        sin = torch.empty(n, 256, device=dv)
        ops.sinus_embed(flat_t.contiguous(), self.sinus_freqs, sin)
        h = torch.empty(n, cfg.dim, device=dv)
        ops.gemm_f32(sin, self.tm_w0, self.tm_b0, out=h, act=ops.ACT_SILU)
        temb = torch.empty(n, cfg.dim, device=dv)
        ops.gemm_f32(h, self.tm_w2, self.tm_b2, out=temb)
        return temb

    def time_tables(self, t: Tensor) -> Tensor:
        """t: f32 [E] (or [E, rows]) -> modulation table [E, rows, L*6D + 2D] f32 (K2 + the AdaLN linears)."""
        cfg, dv = self.cfg, self.device
        t2 = t.reshape(t.shape[0], -1)
        E, rows = t2.shape
        temb = self.time_embed(t2.reshape(-1))
        mod = torch.empty(E * rows, self.row_stride, device=dv)
        D = cfg.dim
        for l, (w, b) in enumerate(self.adaln):
            ops.gemm_f32(temb, w, b, out=mod[:, l * 6 * D:(l + 1) * 6 * D], a_act=ops.ACT_SILU)
        ops.gemm_f32(temb, self.final_w, self.final_b, out=mod[:, self.L * 6 * D:], a_act=ops.ACT_SILU)
        self._last_temb = temb
        return mod.view(E, rows, self.row_stride)

    def text_embed(self, text: Optional[Tensor], B: int, N: int, drop_text: bool) -> Tensor:
        """TextEmbedding.forward (backbones/dit.py:54-87) -> f32 [B, N, text_dim]."""
        return self.text_embed_multi(text, B, N, (drop_text,))[0]

    def text_embed_multi(self, text: Optional[Tensor], B: int, N: int, drops: Tuple[bool, ...]) -> List[Tensor]:
        """TextEmbedding.forward for several drop_text settings of ONE call in one pass: the guidance branches embed the same
        ids once kept and once dropped (all filler tokens, masked at the same padded positions: dit.py:62-66), so the
        variants are stacked along the batch axis and every ConvNeXt launch serves all of them (the GRN statistics are per
        batch item) -> [f32 [B, N, text_dim]] in the order of `drops`."""
        cfg, dv = self.cfg, self.device
        V = len(drops)
        keep = None
        if text is None:
            ids = torch.zeros(V * B, N, dtype=I32, device=dv)
        else:
            if text.device.type == "cpu" and text[:, :N].numel():
                # nn.Embedding raises IndexError on an id outside the table (reference backbones/dit.py:59,68: after the
                # truncation to N and the +1 shift): a vocabulary / checkpoint mismatch must not turn into wrong audio.
                # Host-side ids: no sync.
                hi, lo = int(text[:, :N].max()), int(text[:, :N].min())
                if hi + 1 >= self.text_table.shape[0] or lo < -1:
                    raise IndexError(f"index out of range in self: token id {hi if hi + 1 >= self.text_table.shape[0] else lo} "
                                     f"with text_num_embeds = {self.text_table.shape[0] - 1}")
            ids = (h2d(text, dv) + 1)[:, :N]
            ids = torch.nn.functional.pad(ids, (0, N - ids.shape[1]), value=0)
            if cfg.text_mask_padding:
                keep = (ids != 0).to(F32).repeat(V, 1).contiguous()  # mask taken BEFORE the drop (dit.py:62-66)
            ids = torch.cat([torch.zeros_like(ids) if d else ids for d in drops], 0).to(I32).contiguous()
        TD = cfg.text_dim
        VB = V * B
        h = torch.empty(VB, N, TD, device=dv)
        # reference masks only when there are conv blocks (the masked_fill sits inside `if self.extra_modeling`)
        ops.text_gather(ids, self.text_table, self.text_pos, keep if self.text_blocks else None, h)
        if self.text_blocks:
            c = torch.empty_like(h)
            n = torch.empty_like(h)
            p1 = torch.empty(VB, N, 2 * TD, device=dv)
            g = torch.empty_like(p1)
            gws = torch.empty(VB, 2 * TD, device=dv)
            keep_flat = keep.view(-1) if keep is not None else None
            for blk in self.text_blocks:
                ops.dwconv7(h, blk["dw_w"], blk["dw_b"], c)
                ops.layernorm(c.view(VB * N, TD), n.view(VB * N, TD), gamma=blk["ln_g"], beta=blk["ln_b"])
                ops.gemm_f32(n.view(VB * N, TD), blk["w1"], blk["b1"], out=p1.view(VB * N, 2 * TD), act=ops.ACT_GELU_ERF)
                ops.grn(p1, g, blk["grn_g"], blk["grn_b"], gws)
                h_new = torch.empty_like(h)
                ops.gemm_f32(g.view(VB * N, 2 * TD), blk["w2"], blk["b2"], out=h_new.view(VB * N, TD),
                             addend=h.view(VB * N, TD), row_scale=keep_flat)
                h = h_new
        return [h[v * B:(v + 1) * B] for v in range(V)]

    def ppg_embed(self, ppg: Optional[Tensor], B: int, N: int, drop_ppg: bool) -> Tensor:
        """PPGEmbedding.forward (backbones/dit.py:140-153), BatchNorm folded -> f32 [B, N, text_dim]."""
        cfg, dv, P = self.cfg, self.device, self.ppg
        pd = cfg.ppg_dim
        x = torch.zeros(B, N, pd, device=dv)
        if ppg is not None and not drop_ppg:
            x[:, : ppg.shape[1]] = ppg.to(dv, F32)[:, :N]
        if cfg.ppg_transformer:
            return self._ppg_transformer(x.view(B * N, pd), B, N)
        h = torch.empty(B * N, pd, device=dv)
        ops.gemm_f32(x.view(B * N, pd), P["w0"], P["b0"], out=h)
        col = torch.empty(B, N, 5 * pd, device=dv)
        for wf, bf in P["convs"]:
            ops.im2col(h.view(B, N, pd), col, 5, 2)
            h2 = torch.empty_like(h)
            ops.gemm_f32(col.view(B * N, 5 * pd), wf, bf, out=h2, act=ops.ACT_RELU)
            h = h2
        out = torch.empty(B, N, cfg.text_dim, device=dv)
        ops.gemm_f32(h, P["w15"], P["b15"], out=out.view(B * N, cfg.text_dim))
        return out

    def _ppg_transformer(self, x: Tensor, B: int, N: int) -> Tensor:
        """PPGEmbedding(use_transformer=True) (backbones/dit.py:105-119): post-norm nn.TransformerEncoderLayer stack
        (self-attention over the whole padded sequence, no mask, GELU(erf), LayerNorm eps 1e-5) + Linear, fp32."""
        cfg, dv, P = self.cfg, self.device, self.ppg
        d, H = cfg.ppg_dim, cfg.ppg_heads
        dh = d // H
        Np = (N + 3) // 4 * 4
        qkv = torch.empty(B * N, 3 * d, device=dv)
        vt = torch.zeros(d, Np, device=dv)
        sc, pr = torch.empty(N, Np, device=dv), torch.empty(N, Np, device=dv)
        ctx, t1 = torch.empty(B * N, d, device=dv), torch.empty(B * N, d, device=dv)
        scale = 1.0 / math.sqrt(dh)
        for L in P["layers"]:
            ops.gemm_f32(x, L["w_in"], L["b_in"], out=qkv)
            for b in range(B):
                r0, r1 = b * N, (b + 1) * N
                # V^T [d, N] = Wv x^T; the value bias is added after P.V (softmax rows sum to one)
                ops.gemm_f32(L["w_in"][2 * d:], x[r0:r1], None, out=vt[:, :N])
                for h in range(H):
                    c0, c1 = h * dh, (h + 1) * dh
                    ops.gemm_f32(qkv[r0:r1, c0:c1], qkv[r0:r1, d + c0:d + c1], None, out=sc[:, :N])
                    ops.softmax_rows(sc, pr, N, scale)
                    ops.gemm_f32(pr, vt[c0:c1], L["b_in"][2 * d + c0:2 * d + c1], out=ctx[r0:r1, c0:c1])
            ops.gemm_f32(ctx, L["w_o"], L["b_o"], out=t1, addend=x)
            h1 = torch.empty(B * N, d, device=dv)
            ops.layernorm(t1, h1, gamma=L["g1"], beta=L["be1"], eps=1e-5)
            mid = torch.empty(B * N, L["w1"].shape[0], device=dv)
            ops.gemm_f32(h1, L["w1"], L["b1"], out=mid, act=ops.ACT_GELU_ERF)
            ops.gemm_f32(mid, L["w2"], L["b2"], out=t1, addend=h1)
            x = torch.empty(B * N, d, device=dv)
            ops.layernorm(t1, x, gamma=L["g2"], beta=L["be2"], eps=1e-5)
        out = torch.empty(B, N, cfg.text_dim, device=dv)
        ops.gemm_f32(x, P["w_out"], P["b_out"], out=out.view(B * N, cfg.text_dim))
        return out

    def input_const(self, cond: Tensor, text_emb: Tensor, ppg_emb: Optional[Tensor], drop_audio_cond: bool,
                    out: Tensor) -> None:
        """The (cond, text, ppg) columns of InputEmbedding.proj + bias (backbones/dit.py:169-175) -> out f32 [B*N, D]."""
        cfg = self.cfg
        B, N, mel = cond.shape
        TD = cfg.text_dim
        M = B * N
        w = self.in_w
        cur = None
        if ppg_emb is not None:
            cur = torch.empty(M, cfg.dim, device=self.device)
            ops.gemm_f32(ppg_emb.view(M, TD), w[:, 2 * mel + TD: 2 * mel + 2 * TD], None, out=cur)
        last_is_text = drop_audio_cond
        dst = out if last_is_text else torch.empty(M, cfg.dim, device=self.device)
        ops.gemm_f32(text_emb.view(M, TD), w[:, 2 * mel: 2 * mel + TD], self.in_b, out=dst, addend=cur)
        if not drop_audio_cond:
            ops.gemm_f32(cond.reshape(M, mel), w[:, mel: 2 * mel], None, out=out, addend=dst)

    def rope_table(self, N: int) -> Tensor:
        cs = torch.empty(N, 32, 2, device=self.device)
        ops.rope_table(self.inv_freq, cs)
        return cs

    # ------------------------------------------------------------------ plan / workspace

    def plan_fuses(self, M: int, cd: Optional[Tensor]) -> bool:
        return bool(cd is not None and self.can_fuse_ln and self.fuse_rows_ok(M) and cd.shape[1] == 1)

    def plan_shape(self, S: int, B: int, N: int, mod_rows: int, fuse: bool) -> "_C.DitPlan":
        """The shape half of a plan (what f5e_workspace_bytes reads)."""
        cfg = self.cfg
        p = _C.DitPlan()
        p.S, p.B, p.N, p.n_pad, p.D, p.H = S, B, N, (N + 63) // 64 * 64, cfg.dim, cfg.heads
        p.rope_heads, p.FF, p.L, p.mel, p.mod_rows = self.rope_heads, self.FF, self.L, cfg.mel_dim, mod_rows
        p.fuse_ln = 1 if fuse else 0
        if self.skip_w is not None:
            p.w_skip = self.skip_w.data_ptr()
        return p

    def workspace_layout(self, p: "_C.DitPlan") -> "_C.DitWorkspace":
        import ctypes as C
        w = _C.DitWorkspace()
        _C.check(_C.lib().f5e_workspace_bytes(C.byref(p), C.byref(w)), "f5e_workspace_bytes")
        return w

    @staticmethod
    def zero_pads(w: "_C.DitWorkspace", arena: Tensor) -> None:
        """q / k / vt pad rows are never written by the kernels: the buffers must start out zero."""
        q0 = int(w.offset[_C.WS_NAMES.index("q")])
        q1 = int(w.offset[_C.WS_NAMES.index("vt")] + w.bytes[_C.WS_NAMES.index("vt")])
        arena[q0:q1].zero_()

    def workspace(self, p: "_C.DitPlan", arena: Optional[Tensor] = None) -> Tuple["_C.DitWorkspace", Tensor]:
        """ONE arena sized and laid out by the library (f5e_workspace_bytes).  Allocated here with its q / k / vt pads
        zero-filled -- or carved by the caller (run_ode's per-thread pool), who then zeroes the pads when it must."""
        w = self.workspace_layout(p)
        if arena is None:
            arena = torch.empty(int(w.total), dtype=torch.uint8, device=self.device)
            self.zero_pads(w, arena)
        elif arena.numel() < int(w.total):
            raise _C.F5EError(f"workspace arena of {arena.numel()} bytes, {int(w.total)} needed")
        return w, arena

    def make_plan(self, S: int, B: int, N: int, y: Tensor, in_const: Tensor, mod: Tensor, eval_ptr: Optional[Tensor],
                  rope_cs: Tensor, seq_len: Optional[Tensor], pred: Optional[Tensor] = None,
                  cd: Optional[Tensor] = None, arena: Optional[Tensor] = None) -> "_Plan":
        cfg = self.cfg
        M = S * N
        fuse = self.plan_fuses(M, cd)
        p = self.plan_shape(S, B, N, mod.shape[1], fuse)
        w, arena = self.workspace(p, arena)
        assert int(w.n_pad) == p.n_pad
        base = arena.data_ptr()
        for i, name in enumerate(_C.WS_NAMES):
            if w.bytes[i]:
                setattr(p, name, base + int(w.offset[i]))
        o, nb = int(w.offset[_C.WS_NAMES.index("pred")]), int(w.bytes[_C.WS_NAMES.index("pred")])
        pred_t = pred if pred is not None else arena[o:o + nb].view(F32).view(M, cfg.mel_dim)
        p.pred = pred_t.data_ptr()
        p.y = y.data_ptr()
        p.w_x, p.ldw_x = self.in_w.data_ptr(), self.in_w.stride(0)
        p.in_const = in_const.data_ptr()
        p.convpos_w1, p.convpos_b1 = self.cp[0][0].data_ptr(), self.cp[0][1].data_ptr()
        p.convpos_w2, p.convpos_b2 = self.cp[1][0].data_ptr(), self.cp[1][1].data_ptr()
        p.convpos_groups = 16
        p.rope_cs = rope_cs.data_ptr()
        p.seq_len = seq_len.data_ptr() if seq_len is not None else None
        p.mod = mod.data_ptr()
        p.eval_ptr = eval_ptr.data_ptr() if eval_ptr is not None else None
        p.blocks = self.block_arr
        p.w_proj, p.b_proj = self.proj_w.data_ptr(), self.proj_b.data_ptr()
        if fuse:
            p.cd, p.cd_stride = cd.data_ptr(), cd.shape[2]
            # small-M chains stream every weight from HBM: let each launch prefetch for the one after next (f5e_abi.h)
            p.mall_prefetch = 1 if (M <= self.LN_FUSE_MAX_ROWS and self.mall_prefetch) else 0
        return _Plan(p, dict(arena=arena, pred=pred_t), (y, in_const, mod, eval_ptr, rope_cs, seq_len, cd), w)

    def forward(self, plan: "_Plan") -> Tensor:
        ops.dit_forward(plan.c)
        return plan.ws["pred"]


class _Plan:
    def __init__(self, c_plan, ws, keep, layout=None):
        self.c, self.ws, self.keep, self.layout = c_plan, ws, keep, layout


class _LoopState:
    """Everything one (thread, shape) pair needs to integrate again WITHOUT allocating or capturing: the workspace
    arenas, the once-per-call buffers the captured kernels read (in_const, rope table, lengths, counters, ODE state,
    trajectory) and the instantiated hipGraphs.  The first call with a key launches eagerly (nothing is captured for
    shapes that never come back, e.g. an eval stream of distinct lengths), the second captures ONE step and replays it
    `steps` times, and from the third call on the WHOLE loop is one graph launch -- no capture, and none of the 8.6 us of
    idle the GPU spends between two graph launches.  A state belongs to the thread (= side stream) that made it: all writes to its buffers are ordered on
    that stream, which is what makes the reuse safe without host synchronisation.  (Round 3: "that stream" is the caller's
    current stream -- part of the cache key -- and the side stream only captures.)"""

    def __init__(self):
        self.uses = 0
        self.nbytes = 0
        self.pool: Optional["_Pool"] = None
        self.step_graph: Optional[ops.Graph] = None
        self.loop_graph: Optional[ops.Graph] = None
        self.buf: dict = {}
        self.plans_a = self.plans_b = None

    def retire(self):
        for g in (self.step_graph, self.loop_graph):
            if g is not None:
                g.retire()
        self.step_graph = self.loop_graph = None


class _Pool:
    """Device memory of the persistent loop states of one (thread, stream).  Their runs are ordered on that stream and never
    overlap in time, so the states of different shapes are all carved from offset 0 of the same buffer: the footprint is
    the largest shape's, not one arena per cached shape (~1.7 GB per C3-size state, times the LRU depth).
    `owner` = the state whose run last used the buffer: another state must re-zero its q / k / vt pads."""

    def __init__(self, nbytes: int, device):
        self.buf = torch.empty(nbytes, dtype=torch.uint8, device=device)
        self.owner = None


LOOP_CACHE_ENTRIES = 8   # per thread; a C4-style stream of distinct lengths just cycles through them
LOOP_CACHE_BYTES = 24 << 30   # ... and at most this much device memory per thread (a C3-size state pins ~1.7 GB of arena)


@dataclass
class SamplerInputs:
    """Everything the loop needs, already on the device (built by CFM.sample*)."""
    step_cond: Tensor            # [B, N, mel] f32
    text: Optional[Tensor]       # [B, nt] int64 (pad -1) or None
    ppg: Optional[Tensor]        # [B, np, ppg_dim] or None
    y0: Tensor                   # [B, N, mel] f32
    t: Tensor                    # [steps+1] f32 time grid (host or device)
    seq_len: Optional[Tensor]    # [B] int32 durations, or None when B == 1 (reference: mask=None)
    branches: List[Tuple[bool, bool, bool]]  # (drop_audio_cond, drop_text, drop_ppg) per CFG branch, p0 first
    mode: int                    # 0 plain, 1 cfg, 2 three-branch
    w0: float = 0.0
    w1: float = 0.0
    method: str = "euler"


class KernelTimer:
    """HIP-event pairs recorded by f5e_dit_forward around every launch of one op class (eager launches only)."""

    def __init__(self, op, capacity: int = 4096):
        """op: one F5E_OP_* class, or a list of classes timed in the same pass (read_by_op)."""
        import ctypes as C
        if not isinstance(op, int):
            m = 0
            for o in op:
                m |= 1 << int(o)
            op = -m
        self.op, self.capacity = op, capacity
        self.handle = C.c_void_p()
        _C.check(_C.lib().f5e_timer_create(capacity, C.byref(self.handle)), "f5e_timer_create")

    def reset(self):
        _C.check(_C.lib().f5e_timer_reset(self.handle), "f5e_timer_reset")

    def read_ms(self) -> List[float]:
        import ctypes as C
        buf = (C.c_float * self.capacity)()
        cnt = C.c_int(0)
        _C.check(_C.lib().f5e_timer_read(self.handle, buf, self.capacity, C.byref(cnt)), "f5e_timer_read")
        return [buf[i] for i in range(cnt.value)]

    def read_by_op(self) -> Dict[int, List[float]]:
        """{op class: [ms per launch]} of a multi-class timer."""
        import ctypes as C
        ms = self.read_ms()
        buf = (C.c_int * self.capacity)()
        cnt = C.c_int(0)
        _C.check(_C.lib().f5e_timer_read_ops(self.handle, buf, self.capacity, C.byref(cnt)), "f5e_timer_read_ops")
        out: Dict[int, List[float]] = {}
        for i in range(min(cnt.value, len(ms))):
            out.setdefault(int(buf[i]), []).append(ms[i])
        return out

    def __del__(self):
        try:
            _C.lib().f5e_timer_destroy(self.handle)
        except Exception:
            pass


def h2d(t: Tensor, device, dtype=None) -> Tensor:
    """Host tensor -> device through pinned memory, without blocking the host (a pageable hipMemcpyAsync waits for the
    stream to drain, which would serialise the next call's prep behind the previous call's ODE loop)."""
    if t.device.type != "cpu":
        return t.to(device, dtype) if dtype is not None else t.to(device)
    if dtype is not None:
        t = t.to(dtype)
    return t.contiguous().pin_memory().to(device, non_blocking=True)


def ode_setup(engine: DiTEngine, inp: SamplerInputs) -> dict:
    """Host -> device part of an integration: time grid -> (evaluation times, step coefficients), the cached
    modulation / fused-AdaLN tables, and the small pinned uploads.  Call it on the CALLER's stream, never on the stream
    that will capture the hipGraph: the pinned-memory allocator later polls the events of these copies
    (hipEventQuery), and polling an event of a stream that is capturing invalidates that capture (seen with two
    threads sampling on one model).  It also puts the shared tables on a stream every capture stream waits for."""
    dv = engine.device
    t = inp.t.detach().to("cpu", torch.float32)
    steps = t.shape[0] - 1
    if inp.method == "euler":
        eps_per_step = 1
        t_eval = t[:-1].clone()
        coef = (t[1:] - t[:-1]).clone()
    elif inp.method == "midpoint":
        eps_per_step = 2
        dt = t[1:] - t[:-1]
        half = 0.5 * dt
        t_eval = torch.stack((t[:-1], t[:-1] + half), 1).reshape(-1)
        coef = torch.stack((half, dt), 1).reshape(-1)
    else:
        raise _C.F5EError(f"unsupported ODE method {inp.method!r} (euler, midpoint)")
    mod, cd, coef_d = engine.time_tables_cached(t_eval, coef)     # [E, 1, row_stride], [E, 1, cd_stride] or None, [E]
    seq_len = h2d(inp.seq_len, dv, I32) if inp.seq_len is not None else None
    return dict(steps=steps, eps_per_step=eps_per_step, mod=mod, cd=cd if engine.fuse_ln else None,
                coef_d=coef_d, seq_len=seq_len)


def _loop_state(engine: DiTEngine, key: tuple, persistent: bool) -> Tuple[_LoopState, bool]:
    """The calling thread's state for `key` (LRU), or a throw-away one for eager / instrumented runs.  A fresh state is NOT
    yet in the cache: run_ode publishes it (_publish_state) once its buffers, plans and first capture exist, so a failed
    allocation never leaves a half-built entry behind."""
    if not persistent:
        return _LoopState(), True
    cache = getattr(engine._loops, "cache", None)
    if cache is None:
        cache = engine._loops.cache = {}
    st = cache.pop(key, None)
    if st is not None:
        cache[key] = st                                 # most recently used last
        return st, False
    return _LoopState(), True


def _cache_bytes(cache: dict) -> int:
    """Device memory pinned by a thread's cached loop states: every DISTINCT pool block once (all states of one (thread,
    stream) are carved from the same block) plus the private buffers of un-pooled states."""
    pools = {id(x.pool): x.pool.buf.numel() for x in cache.values() if x.pool is not None}
    return sum(pools.values()) + sum(x.nbytes for x in cache.values() if x.pool is None)


def _pending_list(engine: DiTEngine) -> list:
    pending = getattr(engine._loops, "pending", None)
    if pending is None:
        pending = engine._loops.pending = []
    return pending


def _publish_state(engine: DiTEngine, key: tuple, st: _LoopState) -> None:
    """Insert a fully built state and evict least-recently-used ones beyond LOOP_CACHE_ENTRIES entries or
    LOOP_CACHE_BYTES of device memory (_cache_bytes: a pool block counts once, whoever shares it).  Runs on the CALLER's
    stream, outside any capture; evicted states are only queued here (engine._loops.pending) and run_ode parks their graphs
    behind an event on that stream once the call's last launch is queued (retire_pending)."""
    st.nbytes = 0 if st.pool is not None else sum(t.numel() * t.element_size() for t in st.buf.values() if isinstance(t, Tensor))
    cache = engine._loops.cache
    cache[key] = st
    pending = _pending_list(engine)
    while len(cache) > 1 and (len(cache) > LOOP_CACHE_ENTRIES or _cache_bytes(cache) > LOOP_CACHE_BYTES):
        pending.append(cache.pop(next(iter(cache))))


def retire_pending(engine: DiTEngine) -> None:
    """Park the graphs of evicted loop states behind an event on the CURRENT stream and free the finished ones.  Runs on the
    caller's stream (run_ode's last act; everything of a call executes there), never while capturing."""
    pending = getattr(engine._loops, "pending", None)
    while pending:
        pending.pop().retire()
    ops.Graph.reap()


def run_ode(engine: DiTEngine, inp: SamplerInputs, use_graph: bool = True, want_trajectory: bool = True,
            timer: Optional[KernelTimer] = None, chains: Optional[int] = None, setup: Optional[dict] = None,
            capture_stream: Optional["torch.cuda.Stream"] = None) -> Tensor:
    """Integrates dy/dt = v(t, y) on the given grid (torchdiffeq fixed-grid euler / midpoint).

    Returns the trajectory [steps+1, B, N, mel] (or [2, ...] = (y0, y_final) when want_trajectory is False).
    `setup` = ode_setup(engine, inp) made on the caller's stream (required when this runs on a capture stream that other
    threads' calls may overlap); made here when omitted.
    """
    cfg, dv = engine.cfg, engine.device
    B, N, mel = inp.y0.shape
    nb = len(inp.branches)
    S = nb * B
    if setup is None:
        setup = ode_setup(engine, inp)
    steps, eps_per_step, mod, cd = setup["steps"], setup["eps_per_step"], setup["mod"], setup["cd"]
    coef_d = setup["coef_d"]
    n_chains = chains or 1
    if nb % n_chains:
        raise _C.F5EError(f"chains={n_chains} must divide the number of CFG branches {nb}")
    per = nb // n_chains  # branches per chain
    engine.last_n_chains = n_chains
    n = B * N * mel
    masked = setup["seq_len"] is not None

    # Chains (opt-in): the CFG branches are independent until the combine, so they can run as PARALLEL chains of the
    # captured graph.  Measured on MI355X at C2 (tools/chain_ab.py): 57-70 ms/pass with high run-to-run variance vs
    # a stable 60.5 ms for the single batched forward, so the default keeps ONE forward over all branches (weights
    # stream once per step); results are bit-identical either way.
    persistent = use_graph and steps > 1 and timer is None
    # a state's buffers are written and read in the order of ONE stream: the caller's current stream is part of the key
    run_stream = torch.cuda.current_stream(dv).cuda_stream
    key = (S, B, N, nb, inp.mode, float(inp.w0), float(inp.w1), inp.method, steps, want_trajectory, n_chains, masked,
           mod.data_ptr(), cd.data_ptr() if cd is not None else 0, coef_d.data_ptr(), run_stream)
    st, fresh = _loop_state(engine, key, persistent)
    bf = st.buf
    if fresh:
        bf["keep"] = (mod, cd, coef_d)      # the captured kernels read these: keep them alive past a table-cache eviction
        bf["rope_cs"] = engine.rope_table(N)
        traj_rows = (steps + 1) if want_trajectory else 2
        fuse = engine.plan_fuses(per * B * N, cd)
        ws_bytes = int(engine.workspace_layout(engine.plan_shape(per * B, B, N, mod.shape[1], fuse)).total)
        n_ws = n_chains * eps_per_step
        sizes = [("in_const", S * N * cfg.dim * 4), ("seq_len", S * 4 if masked else 0), ("ctr", 8),
                 ("traj", traj_rows * n * 4), ("y", n * 4), ("y_mid", n * 4 if eps_per_step == 2 else 0),
                 ("pred_all", S * N * mel * 4)] + [(f"ws{i}", ws_bytes) for i in range(n_ws)]
        offs, total = {}, 0
        for name, nb_ in sizes:
            offs[name] = (total, nb_)
            total += (nb_ + 255) // 256 * 256
        if persistent:
            # every persistent state of this thread lives at offset 0 of ONE pool (see _Pool); a larger shape gets a new
            # pool, the states built on the old one keep it alive
            pools = getattr(engine._loops, "pools", None)
            if pools is None:
                pools = engine._loops.pools = {}
            pool = pools.get(run_stream)          # one pool per (thread, stream): its users never overlap in time
            if pool is None or pool.buf.numel() < total:
                if pool is not None:
                    # superseded by a larger block: the cached states carved from the old one would keep it pinned (a
                    # stream of growing shapes would hold several multi-GB blocks at once) -- drop them with it
                    cache_ = getattr(engine._loops, "cache", {})
                    for k_ in [k_ for k_, s_ in cache_.items() if s_.pool is pool]:
                        _pending_list(engine).append(cache_.pop(k_))
                pool = pools[run_stream] = _Pool(total + total // 4, dv)
            st.pool, base = pool, pool.buf
        else:
            base = torch.empty(total, dtype=torch.uint8, device=dv)

        def carve(name, dtype, *shape):
            o, nb_ = offs[name]
            return base[o:o + nb_].view(dtype).view(*shape) if nb_ else None

        bf["in_const"] = carve("in_const", F32, S * N, cfg.dim)
        bf["seq_len"] = carve("seq_len", I32, S)
        bf["ctr"] = carve("ctr", I32, 2)     # [0] evaluation counter, [1] ode_update's arrival counter
        bf["traj"] = carve("traj", F32, traj_rows, B, N, mel)
        bf["y"] = carve("y", F32, B, N, mel)
        bf["y_mid"] = carve("y_mid", F32, B, N, mel)
        bf["pred_all"] = carve("pred_all", F32, S * N, mel)
        bf["ws"] = [carve(f"ws{i}", torch.uint8, ws_bytes) for i in range(n_ws)]
    in_const, rope_cs, seq_len, traj, y, y_mid, pred_all = (bf[k] for k in ("in_const", "rope_cs", "seq_len", "traj", "y",
                                                                          "y_mid", "pred_all"))
    eval_ptr, done = bf["ctr"][0:1], bf["ctr"][1:2]

    # once-per-call tensors, written into the persistent buffers
    cache: Dict[tuple, Tensor] = {}
    drops = tuple(sorted({dt_ for _, dt_, _ in inp.branches}))     # every text variant of the call in ONE pass of the ConvNeXt stack
    for dt_, emb in zip(drops, engine.text_embed_multi(inp.text, B, N, drops)):
        cache[("t", dt_)] = emb
    for bi, (da, dt_, dp) in enumerate(inp.branches):
        tk = ("t", dt_)
        pe = None
        if cfg.use_ppg:
            pk = ("p", dp)
            if pk not in cache:
                cache[pk] = engine.ppg_embed(inp.ppg, B, N, dp)
            pe = cache[pk]
        engine.input_const(inp.step_cond, cache[tk], pe, da, in_const[bi * B * N:(bi + 1) * B * N])
    if masked:
        seq_len.copy_(setup["seq_len"].repeat(nb))
    bf["ctr"].zero_()
    y.copy_(inp.y0)
    traj[0].copy_(inp.y0)

    if fresh:
        def plans_for(y_in, first_ws):
            out = []
            for c in range(n_chains):
                lo, hi = c * per * B * N, (c + 1) * per * B * N
                sl = seq_len[c * per * B:(c + 1) * per * B] if seq_len is not None else None
                out.append(engine.make_plan(per * B, B, N, y_in, in_const[lo:hi], mod, eval_ptr, rope_cs, sl,
                                            pred=pred_all[lo:hi], cd=cd, arena=bf["ws"][first_ws + c]))
            return out

        st.plans_a = plans_for(y, 0)
        st.plans_b = plans_for(y_mid, n_chains) if y_mid is not None else None
    if st.pool is None or st.pool.owner is not st:
        # the buffer was last used by a state of another shape (or never): the q / k / vt pad rows must read as zeros
        for pl in st.plans_a + (st.plans_b or []):
            engine.zero_pads(pl.layout, pl.ws["arena"])
        if st.pool is not None:
            st.pool.owner = st
    plans_a, plans_b = st.plans_a, st.plans_b
    if timer is not None:
        if use_graph:
            raise _C.F5EError("KernelTimer brackets eager launches only (use_graph=False)")
        for pl in plans_a + (plans_b or []):
            pl.c.timer, pl.c.timer_op = timer.handle, timer.op

    aux_streams = [torch.cuda.Stream(device=dv) for _ in range(n_chains - 1)] if use_graph else []

    def forward_all(plans):
        if not aux_streams:
            for pl in plans:
                engine.forward(pl)
            return
        main = torch.cuda.current_stream(dv)
        fork = torch.cuda.Event()
        fork.record(main)
        joins = []
        for pl, st_ in zip(plans[1:], aux_streams):
            st_.wait_event(fork)
            with torch.cuda.stream(st_):
                engine.forward(pl)
                ev = torch.cuda.Event()
                ev.record(st_)
            joins.append(ev)
        engine.forward(plans[0])
        for ev in joins:
            main.wait_event(ev)

    def one_step(traj_row: Optional[Tensor], whole_traj: bool = False):
        # whole_traj: the captured step writes row (eval + 1) // eps_per_step of the trajectory itself
        kw = dict(traj_stride=n, traj_div=eps_per_step) if whole_traj else {}
        forward_all(plans_a)
        if eps_per_step == 1:
            ops.ode_update(pred_all, n, inp.mode, inp.w0, inp.w1, y, y, coef_d, eval_ptr, traj_row, done, **kw)
        else:
            ops.ode_update(pred_all, n, inp.mode, inp.w0, inp.w1, y, y_mid, coef_d, eval_ptr, None, done)
            forward_all(plans_b)
            ops.ode_update(pred_all, n, inp.mode, inp.w0, inp.w1, y, y, coef_d, eval_ptr, traj_row, done, **kw)

    # One forward over all branches (the default): the step sequence is assembled by the LIBRARY (f5e_sample_loop), so a
    # reference-side binder needs nothing of this module; the opt-in parallel chains keep the Python-assembled step.
    loop = None
    if n_chains == 1:
        import ctypes as C
        loop = _C.LoopPlan()
        loop.eval_a = C.pointer(plans_a[0].c)
        loop.eval_b = C.pointer(plans_b[0].c) if plans_b else None
        loop.mode, loop.w0, loop.w1, loop.n = inp.mode, float(inp.w0), float(inp.w1), n
        loop.y, loop.y_mid = y.data_ptr(), (y_mid.data_ptr() if y_mid is not None else None)
        loop.pred, loop.coef = pred_all.data_ptr(), coef_d.data_ptr()
        loop.eval_ptr, loop.done_ctr = eval_ptr.data_ptr(), done.data_ptr()
        loop.traj = traj.data_ptr() if want_trajectory else None

    def enqueue(reps: int):
        if loop is not None:
            loop.steps = reps
            ops.sample_loop(loop)
        else:
            for _ in range(reps):
                one_step(traj if want_trajectory else None, whole_traj=want_trajectory)

    def capture(reps: int) -> ops.Graph:
        # Capture needs a non-default stream; it executes nothing, so it needs no ordering against the caller's stream
        # either.  Everything that RUNS -- the once-per-call kernels above, eager steps, graph launches -- stays on the
        # caller's current stream: no cross-queue hand-off per call (measured at C2 / C4 below).
        import contextlib
        gr = ops.Graph()
        with (torch.cuda.stream(capture_stream) if capture_stream is not None else contextlib.nullcontext()):
            gr.begin()
            try:
                enqueue(reps)
            finally:
                gr.end()
        return gr

    if persistent:
        # A shape's FIRST call runs eagerly: capturing + instantiating even one step costs ~1.2 ms (C4 stream of distinct
        # lengths: 39.3 ms per utterance with a capture per shape, 38.1 without), and the eager launches of one C call per
        # evaluation are not host-bound.  The second call captures ONE step and replays it; from the third call on the
        # whole loop is one graph launch.  All three forms give the same bits.
        st.uses += 1
        if st.loop_graph is None and st.uses >= 3 and LOOP_GRAPH:
            st.loop_graph = capture(steps)         # this shape keeps coming back: from now on one launch per call
        if st.loop_graph is not None:
            st.loop_graph.launch()
        elif st.uses == 1:
            _publish_state(engine, key, st)        # buffers and plans exist: the state may be reused
            enqueue(steps)
        else:
            if st.step_graph is None:
                st.step_graph = capture(1)
            for i in range(steps):
                st.step_graph.launch()
    else:
        enqueue(steps)
    if not want_trajectory:
        traj[1].copy_(y)
    # graphs of loop states this call evicted: parked behind an event on this (the caller's) stream, after the call's last
    # launch and outside any capture -- every caller of run_ode gets it, not only CFM._integrate
    retire_pending(engine)
    # the state's buffers are rewritten by this thread's next call: hand back a copy (6 MB at C2, a few us)
    return traj.clone() if persistent else traj


# F5E_LOOP_GRAPH=0: keep replaying the one-step graph (A/B switch for the whole-loop graph)
LOOP_GRAPH = os.environ.get("F5E_LOOP_GRAPH", "1") != "0"
