"""ctypes binding of libf5e_hip.so (declared in include/f5e_abi.h).  Fails loudly: no fallback of any kind."""
from __future__ import annotations

import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# F5E_HIP_LIB: diagnostics only (tools/convpos_time.py loads the -DF5E_TOOLS build); the product loads the in-tree library
LIB_PATH = os.environ.get("F5E_HIP_LIB") or os.path.join(_HERE, "libf5e_hip.so")

ABI_VERSION = 2   # F5E_ABI_VERSION of include/f5e_abi.h

ACT_NONE, ACT_SILU, ACT_GELU_ERF, ACT_GELU_TANH, ACT_RELU, ACT_MISH = range(6)

_P, _I, _F, _LL = C.c_void_p, C.c_int, C.c_float, C.c_longlong

# name -> argtypes (restype is int unless listed in _RESTYPE)
SIGNATURES = {
    "f5e_abi_version": [],
    "f5e_last_error": [],
    "f5e_check_device": [],
    "f5e_gemm_bf16_bias": [_P, _P, _I, _P, _I, _P, _P, _I, _I, _I, _I, _I, _I, _I],
    "f5e_gemm_bf16_gate_residual": [_P, _P, _I, _P, _I, _P, _P, _I, _P, _I, _I, _P, _I, _I, _P, _I, _I, _I, _I],
    "f5e_gemm_bf16_qkv_rope": [_P, _P, _I, _P, _I, _P, _P, _P, _P, _I, _I, _I, _P, _P, _P, _I, _I, _I, _I],
    "f5e_gemm_bf16_bias_ln": [_P, _P, _I, _P, _I, _P, _P, _I, _I, _I, _I, _I, _I, _I, _P],
    "f5e_gemm_bf16_gate_residual_ln": [_P, _P, _I, _P, _I, _P, _P, _I, _P, _I, _I, _P, _I, _I, _P, _I, _I, _I, _I, _P],
    "f5e_gemm_bf16_qkv_rope_ln": [_P, _P, _I, _P, _I, _P, _P, _P, _P, _I, _I, _I, _P, _P, _P, _I, _I, _I, _I, _P],
    "f5e_adaln_pre": [_P, _P, _I, _P, _I, _P, _I, _I, _I, _P, _I, _P, _I, _P, _I, _I],
    "f5e_flash_attn": [_P, _P, _P, _P, _P, _I, _P, _I, _I, _I, _I, _I],
    "f5e_layernorm": [_P, _P, _I, _P, _I, _I, _P, _P, _P, _P, _I, _I, _I, _P, _I, _I, _I, _F],
    "f5e_grn": [_P, _P, _P, _P, _P, _P, _I, _I, _I],
    "f5e_l2norm": [_P, _P, _I, _P, _I, _I, _P, _I, _I],
    "f5e_gemm_f32": [_P, _P, _I, _I, _I, _P, _I, _P, _I, _P, _P, _I, _I, _P, _P, _I, _P, _I, _I, _I, _I],
    "f5e_convpos": [_P, _P, _I, _P, _P, _I, _P, _I, _P, _I, _P, _I, _I, _I, _I, _I],
    "f5e_dwconv7": [_P, _P, _P, _P, _P, _I, _I, _I],
    "f5e_im2col": [_P, _P, _P, _I, _I, _I, _I, _I],
    "f5e_sinus_embed": [_P, _P, _P, _P, _I, _I, _F],
    "f5e_rope_table": [_P, _P, _P, _I, _I],
    "f5e_text_gather": [_P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _I],
    "f5e_ode_update": [_P, _P, _LL, _I, _F, _F, _P, _P, _P, _P, _P, _P, _LL],
    "f5e_ode_update_traj": [_P, _P, _LL, _I, _F, _F, _P, _P, _P, _LL, _I, _P, _P, _P, _LL],
    "f5e_advance_eval": [_P, _P],
    "f5e_stitch": [_P, _P, _P, _P, _P, _LL, _I],
    "f5e_cast_bf16": [_P, _P, _P, _LL],
    "f5e_cast_f32": [_P, _P, _P, _LL],
    "f5e_axpby": [_P, _P, _P, _P, _F, _F, _F, _LL],
    "f5e_vq_eval": [_P, _P, _I, _P, _I, _P, _P, _P, _I, _I, _I, _I],
    "f5e_stft_logmel": [_P, _P, _I, _I, _P, _P, _P, _P, _I, _I, _I, _I],
    "f5e_stft_logmel_banded": [_P, _P, _I, _I, _P, _P, _P, _P, _I, _P, _I, _I, _I, _I],
    "f5e_istft_head": [_P, _P, _I, _P, _P, _P, _P, _I, _I, _I, _I],
    "f5e_kaldi_fbank": [_P, _P, _I, _I, _P, _P, _P, _P, _I, _I, _I, _I, _F, _F, _F],
    "f5e_glu": [_P, _P, _I, _P, _I, _LL, _I],
    "f5e_dwconv": [_P, _P, _P, _P, _P, _P, _I, _I, _I, _I],
    "f5e_softmax_rows": [_P, _P, _I, _P, _I, _P, _LL, _I, _I, _F],
    "f5e_dit_forward": [_P, _P],
    "f5e_sample_loop": [_P, _P],
    "f5e_workspace_bytes": [_P, _P],
    "f5e_timer_create": [_I, C.POINTER(C.c_void_p)],
    "f5e_timer_destroy": [_P],
    "f5e_timer_reset": [_P],
    "f5e_timer_read": [_P, C.POINTER(C.c_float), _I, C.POINTER(C.c_int)],
    "f5e_timer_read_ops": [_P, C.POINTER(C.c_int), _I, C.POINTER(C.c_int)],
    "f5e_graph_begin": [_P],
    "f5e_graph_end": [_P, C.POINTER(C.c_void_p)],
    "f5e_graph_launch": [_P, _P],
    "f5e_graph_destroy": [_P],
}
_RESTYPE = {"f5e_last_error": C.c_char_p}


class BlockWeights(C.Structure):
    _fields_ = [(n, _P) for n in ("w_qkv", "b_qkv", "w_out", "b_out", "w_ff1", "b_ff1", "w_ff2", "b_ff2",
                                  "q_norm_w", "k_norm_w")]


class LnFuse(C.Structure):
    """f5e_ln_fuse: one side (consumer: stats..eps, producer: xs_out..stats_out) is filled per launch."""
    _fields_ = [("stats", _P), ("parts", _I), ("c", _P), ("d", _P), ("cd_stride", _I), ("cd_rows", _I),
                ("cd_eval_stride", _I), ("eval_ptr", _P), ("rows_per_seq", _I), ("eps", _F),
                ("xs_out", _P), ("ld_xs", _I), ("next_scale", _P), ("stats_out", _P), ("row_mean", _P)]


class DitPlan(C.Structure):
    _fields_ = (
        [(n, _I) for n in ("S", "B", "N", "n_pad", "D", "H", "rope_heads", "FF", "L", "mel", "mod_rows")]
        + [("y", _P), ("w_x", _P), ("ldw_x", _I), ("in_const", _P)]
        + [(n, _P) for n in ("convpos_w1", "convpos_b1", "convpos_w2", "convpos_b2")] + [("convpos_groups", _I)]
        + [(n, _P) for n in ("rope_cs", "seq_len", "mod", "eval_ptr")]
        + [("blocks", C.POINTER(BlockWeights)), ("w_proj", _P), ("b_proj", _P)]
        + [("w_skip", _P), ("skip_res", _P), ("skip_tmp", _P)]
        + [(n, _P) for n in ("h0", "h0_bf16", "c1", "x", "hn", "q", "k", "vt", "ao", "ff", "pred")]
        + [("timer", _P), ("timer_op", _I)]
        + [("fuse_ln", _I), ("ln_stats", _P), ("ln_rowmean", _P), ("cd", _P), ("cd_stride", _I)]
        + [("mall_prefetch", _I)]
    )

class LoopPlan(C.Structure):
    """f5e_loop_plan: the fixed-grid ODE loop around one (euler) or two (midpoint) f5e_dit_plan evaluations."""
    _fields_ = [("eval_a", C.POINTER(DitPlan)), ("eval_b", C.POINTER(DitPlan)), ("steps", _I), ("mode", _I), ("w0", _F),
                ("w1", _F), ("n", _LL), ("y", _P), ("y_mid", _P), ("pred", _P), ("coef", _P), ("eval_ptr", _P),
                ("done_ctr", _P), ("traj", _P)]


WS_NAMES = ("h0", "h0_bf16", "c1", "x", "hn", "q", "k", "vt", "ao", "ff", "pred", "ln_stats", "skip_res", "skip_tmp",
            "ln_rowmean")


class DitWorkspace(C.Structure):
    """f5e_dit_workspace: byte size / arena offset of every caller-owned buffer of a plan (order = WS_NAMES)."""
    _fields_ = [("n_pad", _I), ("bytes", C.c_ulonglong * len(WS_NAMES)), ("offset", C.c_ulonglong * len(WS_NAMES)),
                ("total", C.c_ulonglong)]


OP_NONE, OP_INPROJ, OP_CONVPOS, OP_LN, OP_QKV, OP_ATTN, OP_OUT, OP_FF1, OP_FF2, OP_FINAL = range(10)


_lib = None


def lib() -> C.CDLL:
    """The loaded library; raises if it has not been built (``python -c 'import __graft_entry__ as g; g.build()'``)."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(
                f"{LIB_PATH} is missing: the HIP extension is required (no CPU/PyTorch fallback exists). "
                "Build it with `make -C f5e-tts_amd/csrc` or `__graft_entry__.build()`."
            )
        handle = C.CDLL(LIB_PATH)
        for name, argtypes in SIGNATURES.items():
            fn = getattr(handle, name)  # AttributeError here = header/library mismatch, also loud
            fn.argtypes = argtypes
            fn.restype = _RESTYPE.get(name, C.c_int)
        if handle.f5e_abi_version() != ABI_VERSION:
            raise RuntimeError(f"libf5e_hip ABI version {handle.f5e_abi_version()} != {ABI_VERSION}")
        _lib = handle
    return _lib


class F5EError(RuntimeError):
    pass


def check(rc: int, name: str) -> None:
    if rc != 0:
        msg = lib().f5e_last_error()
        raise F5EError(f"{name} failed ({rc}): {msg.decode() if msg else '?'}")
