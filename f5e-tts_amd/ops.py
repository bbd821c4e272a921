"""Torch-tensor front for the C ABI: pointer extraction, shape inference and loud validation.  PyTorch here only owns
device memory and the stream; every computation happens in libf5e_hip.so."""
from __future__ import annotations

import ctypes as C
import threading
from typing import Optional

import torch

from . import _C
from ._C import ACT_GELU_ERF, ACT_GELU_TANH, ACT_MISH, ACT_NONE, ACT_RELU, ACT_SILU, check, lib  # noqa: F401

Tensor = torch.Tensor
_checked_device = False


def _stream() -> int:
    return torch.cuda.current_stream().cuda_stream


def require_device() -> None:
    """Raise unless a gfx950 device is current and the library is loadable."""
    global _checked_device
    if not _checked_device:
        if not torch.cuda.is_available():
            raise _C.F5EError("f5e_tts_amd needs a ROCm GPU (gfx950); torch.cuda.is_available() is False and there "
                              "is no CPU fallback")
        check(lib().f5e_check_device(), "f5e_check_device")
        _checked_device = True


def _p(t: Optional[Tensor], dtype=None, name="tensor"):
    if t is None:
        return None
    if not t.is_cuda:
        raise _C.F5EError(f"{name} must live on the GPU (got {t.device}); there is no CPU path")
    if dtype is not None and t.dtype != dtype:
        raise _C.F5EError(f"{name} must be {dtype} (got {t.dtype})")
    if not t.is_contiguous():
        raise _C.F5EError(f"{name} must be contiguous")
    return C.c_void_p(t.data_ptr())


BF, F32, I32, U8 = torch.bfloat16, torch.float32, torch.int32, torch.uint8


def ln_consumer(stats: Tensor, c: Tensor, d: Tensor, rows_per_seq: int, row_mean: Tensor, eval_ptr: Optional[Tensor] = None,
                cd_eval_stride: int = 0, eps: float = 1e-6) -> "_C.LnFuse":
    """Consumer side of the fused AdaLN (see f5e_ln_fuse): stats f32 [M, parts, 2] (tile means relative to row_mean);
    c, d f32 [cd_rows, N] views; row_mean f32 [M]: the rows' centring offsets, moved along by this launch."""
    if c.stride(0) != d.stride(0) or c.shape != d.shape:
        raise _C.F5EError("c and d tables must share shape and row stride")
    f = _C.LnFuse()
    f.stats, f.parts = _p(stats, F32, "stats").value, stats.shape[1]
    f.c, f.d, f.cd_stride, f.cd_rows = c.data_ptr(), d.data_ptr(), c.stride(0), c.shape[0]
    f.cd_eval_stride, f.rows_per_seq, f.eps = cd_eval_stride, rows_per_seq, eps
    f.eval_ptr = eval_ptr.data_ptr() if eval_ptr is not None else None
    f.row_mean = _p(row_mean, F32, "row_mean").value
    f._keep = (stats, c, d, eval_ptr, row_mean)
    return f


def ln_producer(xs_out: Tensor, next_scale: Tensor, stats_out: Tensor, row_mean: Tensor) -> "_C.LnFuse":
    """Producer side: xs_out bf16 [M, N]; next_scale f32 [gate_rows, N] view with the gate's row stride;
    stats_out f32 [M, N // 64, 2]; row_mean f32 [M]: the rows' centring offsets (read)."""
    f = _C.LnFuse()
    f.xs_out, f.ld_xs = _p(xs_out, BF, "xs_out").value, xs_out.stride(0)
    f.next_scale, f.stats_out = next_scale.data_ptr(), _p(stats_out, F32, "stats_out").value
    f.row_mean = _p(row_mean, F32, "row_mean").value
    f._keep = (xs_out, next_scale, stats_out, row_mean)
    return f


def _ln_ref(ln):
    return C.byref(ln) if ln is not None else None


def adaln_pre(x: Tensor, xs: Tensor, scale: Tensor, stats: Tensor, row_mean: Tensor, rows_per_seq: int,
              eval_ptr: Optional[Tensor] = None, eval_stride: int = 0):
    """Head of the fused-AdaLN chain: row_mean[row] = mean(x[row]), xs = bf16((x - mean) (1 + scale[r])), stats[row] =
    `parts` equal shares of (0, M2) (tile means relative to row_mean)."""
    require_device()
    rows, D = x.shape
    check(lib().f5e_adaln_pre(_stream(), _p(x, F32, "x"), x.stride(0), _p(xs, BF, "xs"), xs.stride(0),
                              C.c_void_p(scale.data_ptr()), scale.stride(0), scale.shape[0], rows_per_seq,
                              _p(eval_ptr, I32, "eval_ptr"), eval_stride, _p(stats, F32, "stats"), stats.shape[1],
                              _p(row_mean, F32, "row_mean"), rows, D), "f5e_adaln_pre")


def gemm_bf16_bias(a: Tensor, w: Tensor, bias: Optional[Tensor], out: Tensor, act: int = ACT_NONE, tile_hint: int = 0,
                   ln=None):
    """out[M,N] = act(a[M,K] @ w[N,K].T + bias); out bf16 or f32 (no activation).  ln: ``ln_consumer(...)``."""
    require_device()
    M, K = a.shape
    N = w.shape[0]
    check(lib().f5e_gemm_bf16_bias_ln(_stream(), _p(a, BF, "a"), a.stride(0), _p(w, BF, "w"), w.stride(0),
                                      _p(bias, F32, "bias"), _p(out, None, "out"), out.stride(0), M, N, K, act,
                                      1 if out.dtype == F32 else 0, tile_hint, _ln_ref(ln)), "f5e_gemm_bf16_bias")
    return out


def gemm_bf16_gate_residual(a: Tensor, w: Tensor, bias: Optional[Tensor], resid: Tensor, gate: Tensor,
                            rows_per_seq: int, seq_len: Optional[Tensor] = None, eval_ptr: Optional[Tensor] = None,
                            eval_stride: int = 0, tile_hint: int = 0, ln=None):
    """resid[M,N] += gate[seq % rows] * (a @ w.T + bias), rows past seq_len skipped. gate: [rows, N] f32 view.
    ln: ``ln_producer(...)`` to also emit the next fused-AdaLN consumer's inputs."""
    require_device()
    M, K = a.shape
    N = w.shape[0]
    check(lib().f5e_gemm_bf16_gate_residual_ln(
        _stream(), _p(a, BF, "a"), a.stride(0), _p(w, BF, "w"), w.stride(0), _p(bias, F32, "bias"),
        _p(resid, F32, "resid"), resid.stride(0), C.c_void_p(gate.data_ptr()), gate.stride(0), gate.shape[0],
        _p(eval_ptr, I32, "eval_ptr"), eval_stride, rows_per_seq, _p(seq_len, I32, "seq_len"), M, N, K, tile_hint,
        _ln_ref(ln)), "f5e_gemm_bf16_gate_residual")
    return resid


def gemm_bf16_qkv_rope(a: Tensor, w: Tensor, bias: Tensor, q: Tensor, k: Tensor, vt: Tensor, heads: int,
                       rope_heads: int, cos_sin: Tensor, rows_per_seq: int, tile_hint: int = 0,
                       q_norm_w: Optional[Tensor] = None, k_norm_w: Optional[Tensor] = None, ln=None):
    require_device()
    M, K = a.shape
    n_pad = q.shape[2]
    check(lib().f5e_gemm_bf16_qkv_rope_ln(
        _stream(), _p(a, BF, "a"), a.stride(0), _p(w, BF, "w"), w.stride(0), _p(bias, F32, "bias"), _p(q, BF, "q"),
        _p(k, BF, "k"), _p(vt, BF, "vt"), n_pad, heads, rope_heads, _p(cos_sin, F32, "cos_sin"),
        _p(q_norm_w, F32, "q_norm_w"), _p(k_norm_w, F32, "k_norm_w"), rows_per_seq, M, K, tile_hint, _ln_ref(ln)),
        "f5e_gemm_bf16_qkv_rope")


def qk_frag_index(n_pad: int) -> Tensor:
    """offset[pos, d] of the fragment-major Q/K layout (attention.hip): [pos/32][d/16][pos%32][(d/8)%2][d%8]."""
    pos = torch.arange(n_pad)[:, None]
    d = torch.arange(64)[None, :]
    return (((pos // 32) * 4 + d // 16) * 32 + pos % 32) * 16 + ((d // 8) % 2) * 8 + d % 8


def v_frag_index(n_pad: int) -> Tensor:
    """offset[key, d] of the fragment-major V layout: [key/32][(key%32)/16][d/32][d%32][h][j], key%16 = 8(j>>2)+4h+(j&3)."""
    key = torch.arange(n_pad)[:, None]
    d = torch.arange(64)[None, :]
    k16 = key % 16
    j = (k16 // 8) * 4 + k16 % 4
    h = (k16 // 4) % 2
    return (((((key // 32) * 2 + (key % 32) // 16) * 2 + d // 32) * 32 + d % 32) * 2 + h) * 8 + j


def flash_attn(q: Tensor, k: Tensor, vt: Tensor, out: Tensor, rows_per_seq: int, kv_len: Optional[Tensor] = None,
               waves: int = 0):
    """q, k, vt: [S, H, n_pad, 64]-sized bf16 buffers in the fragment-major layouts; ``waves`` = KV splits (0 auto)."""
    require_device()
    S, H, n_pad, _ = q.shape
    check(lib().f5e_flash_attn(_stream(), _p(q, BF, "q"), _p(k, BF, "k"), _p(vt, BF, "vt"), _p(out, BF, "out"),
                               out.stride(0), _p(kv_len, I32, "kv_len"), S, H, rows_per_seq, n_pad, waves),
          "f5e_flash_attn")
    return out


def layernorm(x: Tensor, out: Tensor, gamma: Optional[Tensor] = None, beta: Optional[Tensor] = None,
              scale: Optional[Tensor] = None, shift: Optional[Tensor] = None, rows_per_seq: int = 1,
              eval_ptr: Optional[Tensor] = None, eval_stride: int = 0, eps: float = 1e-6):
    """x f32 [rows, D]; scale/shift: f32 [mod_rows, D] views (row stride free)."""
    require_device()
    rows, D = x.shape
    mod_rows = scale.shape[0] if scale is not None else 0
    mod_stride = scale.stride(0) if scale is not None else 0
    if scale is not None and shift.stride(0) != mod_stride:
        raise _C.F5EError("scale and shift must share a row stride")
    sp = C.c_void_p(scale.data_ptr()) if scale is not None else None
    hp = C.c_void_p(shift.data_ptr()) if shift is not None else None
    check(lib().f5e_layernorm(_stream(), _p(x, F32, "x"), x.stride(0), _p(out, None, "out"), out.stride(0),
                              1 if out.dtype == BF else 0, _p(gamma, F32, "gamma"), _p(beta, F32, "beta"), sp, hp,
                              mod_stride, mod_rows, rows_per_seq, _p(eval_ptr, I32, "eval_ptr"), eval_stride, rows, D,
                              eps), "f5e_layernorm")
    return out


def l2norm(x: Tensor, out: Tensor, g: Tensor):
    """out = x / ||x|| * sqrt(D) * g (x_transformers.RMSNorm); x f32 [rows, D], out bf16 or f32."""
    require_device()
    rows, D = x.shape
    check(lib().f5e_l2norm(_stream(), _p(x, F32, "x"), x.stride(0), _p(out, None, "out"), out.stride(0),
                           1 if out.dtype == BF else 0, _p(g, F32, "g"), rows, D), "f5e_l2norm")
    return out


def grn(x: Tensor, out: Tensor, gamma: Tensor, beta: Tensor, ws: Tensor):
    require_device()
    B, T, Cc = x.shape
    check(lib().f5e_grn(_stream(), _p(x, F32, "x"), _p(out, F32, "out"), _p(ws, F32, "ws"), _p(gamma, F32, "gamma"),
                        _p(beta, F32, "beta"), B, T, Cc), "f5e_grn")
    return out


def gemm_f32(a: Tensor, w: Tensor, bias: Optional[Tensor] = None, *, out: Optional[Tensor] = None,
             out_bf16: Optional[Tensor] = None, M: Optional[int] = None, a_act: int = ACT_NONE, act: int = ACT_NONE,
             ch_scale: Optional[Tensor] = None, addend: Optional[Tensor] = None, row_scale: Optional[Tensor] = None,
             K: Optional[int] = None):
    """fp32 MFMA GEMM; ``a`` [a_rows, >=K] and ``w`` [N, >=K] may be column-sliced views (row stride = ld)."""
    require_device()
    a_rows = a.shape[0]
    K = a.shape[1] if K is None else K
    N = w.shape[0]
    M = a_rows if M is None else M
    for t, nm in ((a, "a"), (w, "w")):
        if not t.is_cuda or t.dtype != F32 or t.stride(1) != 1:
            raise _C.F5EError(f"gemm_f32: {nm} must be an f32 GPU tensor with unit column stride")
    ad = addend
    check(lib().f5e_gemm_f32(
        _stream(), C.c_void_p(a.data_ptr()), a.stride(0), a_rows, a_act, C.c_void_p(w.data_ptr()), w.stride(0),
        _p(bias, F32, "bias"), act, _p(ch_scale, F32, "ch_scale"),
        C.c_void_p(ad.data_ptr()) if ad is not None else None, ad.stride(0) if ad is not None else 0,
        ad.shape[0] if ad is not None else 0, _p(row_scale, F32, "row_scale"),
        C.c_void_p(out.data_ptr()) if out is not None else None, out.stride(0) if out is not None else 0,
        C.c_void_p(out_bf16.data_ptr()) if out_bf16 is not None else None,
        out_bf16.stride(0) if out_bf16 is not None else 0, M, N, K), "f5e_gemm_f32")
    return out if out is not None else out_bf16


def pack_convpos_weight(w: Tensor, groups: int = 16) -> Tensor:
    """Conv1d weight [D][D/groups][31] -> bf16 [groups][31][64 oc][64 ic] (zero padded to 64 x 64 per group)."""
    D, cpg, taps = w.shape
    out = torch.zeros(groups, taps, 64, 64, dtype=w.dtype, device=w.device)
    out[:, :, :cpg, :cpg] = w.view(groups, cpg, cpg, taps).permute(0, 3, 1, 2)
    return out.to(BF).contiguous()


def convpos(x: Tensor, w_packed: Tensor, bias: Tensor, S: int, N: int, *, out_bf16: Optional[Tensor] = None,
            out_f32: Optional[Tensor] = None, resid: Optional[Tensor] = None):
    require_device()
    D = x.shape[1]
    mode = 0 if out_f32 is None else 1
    check(lib().f5e_convpos(_stream(), _p(x, BF, "x"), x.stride(0), _p(w_packed, BF, "w_packed"), _p(bias, F32, "bias"),
                            mode, _p(out_bf16, BF, "out_bf16"), out_bf16.stride(0) if out_bf16 is not None else 0,
                            _p(out_f32, F32, "out_f32"), out_f32.stride(0) if out_f32 is not None else 0,
                            _p(resid, F32, "resid"), resid.stride(0) if resid is not None else 0, S, N, D,
                            w_packed.shape[0]),
          "f5e_convpos")


def dwconv7(x: Tensor, w_t: Tensor, bias: Tensor, out: Tensor):
    require_device()
    B, T, Cc = x.shape
    check(lib().f5e_dwconv7(_stream(), _p(x, F32, "x"), _p(w_t, F32, "w_t"), _p(bias, F32, "bias"), _p(out, F32, "out"),
                            B, T, Cc), "f5e_dwconv7")
    return out


def im2col(x: Tensor, col: Tensor, ksize: int, pad: int):
    require_device()
    B, T, Cin = x.shape
    check(lib().f5e_im2col(_stream(), _p(x, F32, "x"), _p(col, F32, "col"), B, T, Cin, ksize, pad), "f5e_im2col")
    return col


def sinus_embed(t: Tensor, freqs: Tensor, out: Tensor, scale: float = 1000.0):
    require_device()
    E, dim = out.shape
    check(lib().f5e_sinus_embed(_stream(), _p(t, F32, "t"), _p(freqs, F32, "freqs"), _p(out, F32, "out"), E, dim, scale),
          "f5e_sinus_embed")
    return out


def rope_table(inv_freq: Tensor, out: Tensor):
    require_device()
    N, half, _ = out.shape
    check(lib().f5e_rope_table(_stream(), _p(inv_freq, F32, "inv_freq"), _p(out, F32, "out"), N, half), "f5e_rope_table")
    return out


def text_gather(ids: Tensor, table: Tensor, pos: Optional[Tensor], keep: Optional[Tensor], out: Tensor):
    require_device()
    B, N, TD = out.shape
    check(lib().f5e_text_gather(_stream(), _p(ids, I32, "ids"), _p(table, F32, "table"), _p(pos, F32, "pos"),
                                _p(keep, F32, "keep"), _p(out, F32, "out"), B, N, TD,
                                pos.shape[0] if pos is not None else 1, table.shape[0]), "f5e_text_gather")
    return out


def ode_update(pred: Tensor, branch_stride: int, mode: int, w0: float, w1: float, base: Tensor, dst: Tensor,
               coef: Tensor, eval_ptr: Optional[Tensor], traj: Optional[Tensor] = None,
               done_ctr: Optional[Tensor] = None, traj_stride: int = 0, traj_div: int = 1):
    """done_ctr (int32[1], zero): the kernel itself advances *eval_ptr after every block has read it.
    traj_stride > 0: ``traj`` is the whole [rows, ...] trajectory and row (*eval_ptr + 1) // traj_div gets the copy."""
    require_device()
    check(lib().f5e_ode_update_traj(_stream(), _p(pred, F32, "pred"), branch_stride, mode, w0, w1, _p(base, F32, "base"),
                                    _p(dst, F32, "dst"), C.c_void_p(traj.data_ptr()) if traj is not None else None,
                                    traj_stride, traj_div, _p(coef, F32, "coef"), _p(eval_ptr, I32, "eval_ptr"),
                                    _p(done_ctr, I32, "done_ctr"), base.numel()), "f5e_ode_update")
    return dst


def sample_loop(loop: "_C.LoopPlan"):
    """Enqueue loop.steps ODE steps (network evaluation(s) + guidance combine + Euler / midpoint update) on the current stream."""
    require_device()
    check(lib().f5e_sample_loop(_stream(), C.byref(loop)), "f5e_sample_loop")


def advance_eval(eval_ptr: Tensor):
    require_device()
    check(lib().f5e_advance_eval(_stream(), _p(eval_ptr, I32, "eval_ptr")), "f5e_advance_eval")


def stitch(cond: Tensor, y: Tensor, mask_u8: Tensor, out: Tensor):
    require_device()
    Cc = cond.shape[-1]
    check(lib().f5e_stitch(_stream(), _p(cond, F32, "cond"), _p(y, F32, "y"), _p(mask_u8, U8, "mask"),
                           _p(out, F32, "out"), cond.numel() // Cc, Cc), "f5e_stitch")
    return out


def cast_bf16(x: Tensor, out: Tensor):
    require_device()
    check(lib().f5e_cast_bf16(_stream(), _p(x, F32, "x"), _p(out, BF, "out"), x.numel()), "f5e_cast_bf16")
    return out


def cast_f32(x: Tensor, out: Tensor):
    require_device()
    check(lib().f5e_cast_f32(_stream(), _p(x, BF, "x"), _p(out, F32, "out"), x.numel()), "f5e_cast_f32")
    return out


def axpby(x: Tensor, y: Optional[Tensor], out: Tensor, a: float = 1.0, b: float = 1.0, c: float = 0.0):
    """out = a x + b y + c (y optional), f32 contiguous tensors of equal size."""
    require_device()
    if y is not None and y.numel() != x.numel() or out.numel() != x.numel():
        raise _C.F5EError("axpby: size mismatch")
    check(lib().f5e_axpby(_stream(), _p(x, F32, "x"), _p(y, F32, "y"), _p(out, F32, "out"), a, b, c, x.numel()),
          "f5e_axpby")
    return out


def vq_eval(logits: Tensor, vars_: Tensor, combine_groups: bool, out: Tensor, targets: Tensor,
            stats: Optional[Tensor], groups: int, num_vars: int):
    require_device()
    rows = logits.shape[0]
    check(lib().f5e_vq_eval(_stream(), _p(logits, F32, "logits"), logits.stride(0), _p(vars_, F32, "vars"),
                            1 if combine_groups else 0, _p(out, F32, "out"), _p(targets, I32, "targets"),
                            _p(stats, F32, "stats"), rows, groups, num_vars, vars_.shape[-1]), "f5e_vq_eval")
    return out


def stft_logmel(wav: Tensor, window: Tensor, twiddle: Tensor, fb: Tensor, out: Tensor, n_fft: int, hop: int):
    require_device()
    B, nw = wav.shape
    check(lib().f5e_stft_logmel(_stream(), _p(wav, F32, "wav"), nw, wav.stride(0), _p(window, F32, "window"),
                                _p(twiddle, F32, "twiddle"), _p(fb, F32, "fb"), _p(out, F32, "out"), B, n_fft, hop,
                                fb.shape[1]), "f5e_stft_logmel")
    return out


def band_filterbank(fb: Tensor):
    """Dense [n_freqs, n_mels] filterbank -> (compact weights f32 [nnz], band i32 [n_mels, 3] = lo, cnt, offset) on fb's
    device, or None when a filter has interior zeros / the total exceeds the kernel's LDS table (dense kernel then)."""
    h = fb.detach().to("cpu", F32)
    nz = h != 0
    bands, chunks, off = [], [], 0
    for m in range(h.shape[1]):
        idx = torch.nonzero(nz[:, m]).flatten()
        if idx.numel() == 0:
            bands.append((0, 0, off))
            continue
        lo, hi = int(idx[0]), int(idx[-1])
        bands.append((lo, hi - lo + 1, off))       # interior zeros (if any) stay in the run: they add +0 like the dense loop
        chunks.append(h[lo:hi + 1, m])
        off += hi - lo + 1
    if off == 0 or off > 2048:
        return None
    return torch.cat(chunks).contiguous().to(fb.device), torch.tensor(bands, dtype=I32).contiguous().to(fb.device)


def stft_logmel_banded(wav: Tensor, window: Tensor, twiddle: Tensor, fb_compact: Tensor, fb_band: Tensor, out: Tensor,
                       n_fft: int, hop: int):
    require_device()
    B, nw = wav.shape
    check(lib().f5e_stft_logmel_banded(_stream(), _p(wav, F32, "wav"), nw, wav.stride(0), _p(window, F32, "window"),
                                       _p(twiddle, F32, "twiddle"), _p(fb_compact, F32, "fb_compact"),
                                       _p(fb_band, I32, "fb_band"), fb_compact.numel(), _p(out, F32, "out"), B, n_fft, hop,
                                       fb_band.shape[0]), "f5e_stft_logmel_banded")
    return out


def istft_head(z: Tensor, window: Tensor, twiddle: Tensor, frames_ws: Tensor, out: Tensor, B: int, T: int, n_fft: int,
               hop: int):
    require_device()
    check(lib().f5e_istft_head(_stream(), _p(z, F32, "z"), z.stride(0), _p(window, F32, "window"),
                               _p(twiddle, F32, "twiddle"), _p(frames_ws, F32, "frames_ws"), _p(out, F32, "out"), B, T,
                               n_fft, hop), "f5e_istft_head")
    return out


def kaldi_fbank(wav: Tensor, window: Tensor, twiddle: Tensor, fb: Tensor, out: Tensor, win: int, shift: int,
                in_scale: float = 32768.0, preemph: float = 0.97, eps: float = 1.1920928955078125e-07):
    require_device()
    B, nw = wav.shape
    check(lib().f5e_kaldi_fbank(_stream(), _p(wav, F32, "wav"), nw, wav.stride(0), _p(window, F32, "window"),
                                _p(twiddle, F32, "twiddle"), _p(fb, F32, "fb"), _p(out, F32, "out"), B, win, shift,
                                fb.shape[1], in_scale, preemph, eps), "f5e_kaldi_fbank")
    return out


def glu(x: Tensor, out: Tensor):
    """out[r, c] = x[r, c] * sigmoid(x[r, C + c]); x f32 [rows, 2C], out f32 [rows, C] (row strides free)."""
    require_device()
    rows, C = out.shape
    check(lib().f5e_glu(_stream(), _p(x, F32, "x"), x.stride(0), _p(out, F32, "out"),
                        out.stride(0), rows, C), "f5e_glu")
    return out


def dwconv(x: Tensor, w_t: Tensor, bias: Tensor, out: Tensor, keep: Optional[Tensor] = None):
    """Depthwise conv over time, channels-last f32 [B, T, C]; w_t [K, C]."""
    require_device()
    B, T, Cc = x.shape
    check(lib().f5e_dwconv(_stream(), _p(x, F32, "x"), _p(w_t, F32, "w_t"), _p(bias, F32, "bias"), _p(keep, F32, "keep"),
                           _p(out, F32, "out"), B, T, Cc, w_t.shape[0]), "f5e_dwconv")
    return out


def softmax_rows(x: Tensor, out: Tensor, L: int, scale: float, kv_len: Optional[Tensor] = None, rows_per_seq: int = 1):
    """Row softmax of scale * x[:, :len] (zeros beyond); x, out f32 [rows, >= L]."""
    require_device()
    rows = x.shape[0]
    check(lib().f5e_softmax_rows(_stream(), _p(x, F32, "x"), x.stride(0), _p(out, F32, "out"), out.stride(0),
                                 _p(kv_len, I32, "kv_len"), rows, rows_per_seq if kv_len is not None else max(rows, 1), L,
                                 scale), "f5e_softmax_rows")
    return out


def dit_forward(plan: "_C.DitPlan"):
    require_device()
    check(lib().f5e_dit_forward(_stream(), C.byref(plan)), "f5e_dit_forward")


class Graph:
    """hipGraph captured through the C ABI on the current (non-default) torch stream.

    An executable graph owns the kernel-argument storage of its nodes, so it must outlive the launches that are still
    queued: ``retire()`` parks it behind an event recorded on the launch stream and ``reap()`` destroys the parked
    graphs whose event has completed (callers never block on the GPU for this)."""

    _parked = []
    _parked_lock = threading.Lock()

    def __init__(self):
        self.handle = C.c_void_p()

    def retire(self):
        ev = torch.cuda.Event()
        ev.record(torch.cuda.current_stream())
        with Graph._parked_lock:
            Graph._parked.append((self, ev))

    @staticmethod
    def reap():
        with Graph._parked_lock:
            state = [(x, x[1].query()) for x in Graph._parked]    # ONE query per event: it may complete in between
            done = [x for x, finished in state if finished]
            Graph._parked[:] = [x for x, finished in state if not finished]
        for g, _ in done:
            g.destroy()

    def destroy(self):
        if self.handle:
            lib().f5e_graph_destroy(self.handle)
            self.handle = C.c_void_p()

    def begin(self):
        require_device()
        check(lib().f5e_graph_begin(_stream()), "f5e_graph_begin")

    def end(self):
        check(lib().f5e_graph_end(_stream(), C.byref(self.handle)), "f5e_graph_end")

    def launch(self):
        check(lib().f5e_graph_launch(self.handle, _stream()), "f5e_graph_launch")

    def __del__(self):
        try:
            self.destroy()
        except Exception:
            pass
