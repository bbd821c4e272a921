"""PPG extractor mirror (reference ppg/ppg_model.py:11-168, ppg/asr_model.py:221-244 + 814-859, ppg/wenet/transformer/*):
kaldi fbank -> GlobalCMVN -> wenet ConformerEncoder -> 256-d ``linear`` head (the "PPG") -> ``ce.fc`` logits, behind the
reference's ``build_ppg_model`` / ``PPGModelWapper`` API.  It runs once per utterance in front of ``CFM.sample_vc`` /
``sample_tts`` (reference eval/eval_infer_batch_vc.py, model/trainer.py:385-389).

Everything is computed by libf5e_hip.so: the linears, the per-head score / context products and the subsampling conv
(as one Toeplitz-expanded GEMM) on the exact-fp32 MFMA GEMM, LayerNorms, GLU, depthwise conv, row softmax and the fbank
FFT as HIP kernels; torch only holds memory.  Weights-only transforms done once at load: eval-mode BatchNorm and the
global CMVN are folded into the adjacent convolution weights, ``pos_bias_u / v`` into the query projection's bias.

What is built is the configuration the reference ships (``encoder: conformer``, ``input_layer: conv2d`` = 1/2
subsampling for 20 ms PPG frames, ``rel_pos``, macaron feed-forward, convolution module with batch norm, swish, pre-norm,
no chunking: ``extract(stream=False)`` decodes with the full context).  Other encoder variants raise F5EError.
"""
from __future__ import annotations

import json
import math
import os
from typing import Dict, Optional, Tuple

import numpy as np
import torch
from torch import nn

from .. import _C, ops

F32, I32 = torch.float32, torch.int32
Tensor = torch.Tensor


def load_cmvn(cmvn_file: str, is_json: bool) -> Tuple[np.ndarray, np.ndarray]:
    """(mean, 1 / std) from accumulated statistics (reference ppg/wenet/utils/cmvn.py:21-92; json or kaldi text)."""
    if is_json:
        with open(cmvn_file) as f:
            st = json.load(f)
        means, var, count = list(st["mean_stat"]), list(st["var_stat"]), st["frame_num"]
    else:
        with open(cmvn_file, "r") as f:
            arr = f.read().split()
        if not (arr[0] == "[" and arr[-2] == "0" and arr[-1] == "]"):
            raise ValueError("kaldi cmvn: expected the text format of compute-cmvn-stats --binary=false")
        dim = (len(arr) - 4) // 2
        means = [float(v) for v in arr[1:dim + 1]]
        count = float(arr[dim + 1])
        var = [float(v) for v in arr[dim + 2:2 * dim + 2]]
    for i in range(len(means)):
        means[i] /= count
        var[i] = max(var[i] / count - means[i] * means[i], 1.0e-20)
        var[i] = 1.0 / math.sqrt(var[i])
    return np.array(means), np.array(var)


# ------------------------------------------------------------------ parameter containers (reference state_dict names)

class _GlobalCMVN(nn.Module):
    def __init__(self, mean: Tensor, istd: Tensor):
        super().__init__()
        self.register_buffer("mean", mean)
        self.register_buffer("istd", istd)


class _Subsampling2(nn.Module):
    def __init__(self, idim, odim):
        super().__init__()
        self.conv = nn.Sequential(nn.Conv2d(1, odim, 3, 2), nn.ReLU())
        self.out = nn.Sequential(nn.Linear(odim * ((idim - 1) // 2), odim))


class _RelMHA(nn.Module):
    def __init__(self, heads, d):
        super().__init__()
        self.linear_q, self.linear_k, self.linear_v = nn.Linear(d, d), nn.Linear(d, d), nn.Linear(d, d)
        self.linear_out = nn.Linear(d, d)
        self.linear_pos = nn.Linear(d, d, bias=False)
        self.pos_bias_u = nn.Parameter(torch.empty(heads, d // heads))
        self.pos_bias_v = nn.Parameter(torch.empty(heads, d // heads))
        nn.init.xavier_uniform_(self.pos_bias_u)
        nn.init.xavier_uniform_(self.pos_bias_v)


class _FFN(nn.Module):
    def __init__(self, d, units):
        super().__init__()
        self.w_1, self.w_2 = nn.Linear(d, units), nn.Linear(units, d)


class _ConvModule(nn.Module):
    def __init__(self, d, k):
        super().__init__()
        self.pointwise_conv1 = nn.Conv1d(d, 2 * d, 1)
        self.depthwise_conv = nn.Conv1d(d, d, k, padding=(k - 1) // 2, groups=d)
        self.norm = nn.BatchNorm1d(d)
        self.pointwise_conv2 = nn.Conv1d(d, d, 1)


class _ConformerLayer(nn.Module):
    def __init__(self, d, heads, units, k):
        super().__init__()
        self.self_attn = _RelMHA(heads, d)
        self.feed_forward, self.feed_forward_macaron = _FFN(d, units), _FFN(d, units)
        self.conv_module = _ConvModule(d, k)
        for n in ("norm_ff", "norm_mha", "norm_ff_macaron", "norm_conv", "norm_final"):
            setattr(self, n, nn.LayerNorm(d, eps=1e-5))
        self.concat_linear = nn.Linear(2 * d, d)      # a checkpoint key of the reference layer; unused (concat_after False)


class _Encoder(nn.Module):
    def __init__(self, idim, d, heads, units, blocks, k, cmvn):
        super().__init__()
        self.global_cmvn = cmvn
        self.embed = _Subsampling2(idim, d)
        self.after_norm = nn.LayerNorm(d, eps=1e-5)
        self.encoders = nn.ModuleList([_ConformerLayer(d, heads, units, k) for _ in range(blocks)])


class _CE(nn.Module):
    def __init__(self, d, n):
        super().__init__()
        self.fc = nn.Linear(d, n)


class ConformerPPG(nn.Module):
    """The part of the reference ``ASRModel`` that ``extract`` touches (asr_model.py:221-244), same state_dict names:
    ``encoder.*``, ``linear.*``, ``ce.fc.*``.  Decoder / CTC keys of a checkpoint are not on this path and are skipped
    by ``build_ppg_model`` exactly as the reference's key filter does."""

    def __init__(self, input_dim: int = 80, vocab_size: int = 218, output_size: int = 256, attention_heads: int = 4,
                 linear_units: int = 2048, num_blocks: int = 6, cnn_module_kernel: int = 15,
                 global_cmvn: Optional[Tuple[Tensor, Tensor]] = None):
        super().__init__()
        cm = _GlobalCMVN(global_cmvn[0].float(), global_cmvn[1].float()) if global_cmvn is not None else None
        self.encoder = _Encoder(input_dim, output_size, attention_heads, linear_units, num_blocks, cnn_module_kernel, cm)
        self.linear = nn.Linear(output_size, output_size)
        self.ce = _CE(output_size, vocab_size + 1)
        self.input_dim, self.heads, self.dim = input_dim, attention_heads, output_size
        self._engine = None

    @classmethod
    def from_config(cls, configs: dict) -> "ConformerPPG":
        """``init_asr_model`` (asr_model.py:814-859) for the supported encoder family."""
        enc = dict(configs.get("encoder_conf") or {})
        if configs.get("encoder", "conformer") != "conformer":
            raise _C.F5EError("PPG extractor: only `encoder: conformer` is built for MI355X")
        bad = {k: enc[k] for k, want in dict(input_layer="conv2d", pos_enc_layer_type="rel_pos", normalize_before=True,
                                             concat_after=False, macaron_style=True, use_cnn_module=True, causal=False,
                                             cnn_module_norm="batch_norm", activation_type="swish", use_emb=False,
                                             positionwise_conv_kernel_size=1).items() if k in enc and enc[k] != want}
        if bad:
            raise _C.F5EError(f"PPG extractor: unsupported encoder_conf entries {bad}")
        cmvn = None
        if configs.get("cmvn_file") is not None:
            mean, istd = load_cmvn(configs["cmvn_file"], configs["is_json_cmvn"])
            cmvn = (torch.from_numpy(mean).float(), torch.from_numpy(istd).float())
        return cls(configs["input_dim"], configs["output_dim"], enc.get("output_size", 256), enc.get("attention_heads", 4),
                   enc.get("linear_units", 2048), enc.get("num_blocks", 6), enc.get("cnn_module_kernel", 15), cmvn)

    def _apply(self, fn, *a, **kw):
        self._engine = None
        return super()._apply(fn, *a, **kw)

    def load_state_dict(self, *a, **kw):
        self._engine = None
        return super().load_state_dict(*a, **kw)

    def engine(self) -> "ConformerEngine":
        dev = next(self.parameters()).device
        if self._engine is None or self._engine.device != dev:
            if dev.type != "cuda":
                raise _C.F5EError(f"PPG model lives on {dev}: move it to the GPU (there is no CPU path)")
            self._engine = ConformerEngine(self.state_dict(), self.heads, dev, self.input_dim)
        return self._engine

    @torch.no_grad()
    def extract(self, speech: Tensor, speech_lengths: Tensor, stream: bool = False) -> Tuple[Tensor, Tensor]:
        """reference asr_model.py:221-244 -> (ppg [B, T', D], logits [B*T', vocab + 1])."""
        if stream:
            raise _C.F5EError("PPG extractor: the chunk-by-chunk streaming mode is not built (extract(stream=False))")
        assert speech.shape[0] == speech_lengths.shape[0]
        return self.engine().forward(speech, speech_lengths)


class ConformerEngine:
    """Repacked fp32 weights + the launch sequence of ``BaseEncoder.forward`` (wenet/transformer/encoder.py:141-209)."""

    def __init__(self, sd: Dict[str, Tensor], heads: int, device, idim: int):
        ops.require_device()
        self.device = dv = torch.device(device)
        f = lambda k: sd[k].detach().to(dv, F32).contiguous()   # noqa: E731
        self.heads = heads
        cw, cb = f("encoder.embed.conv.0.weight"), f("encoder.embed.conv.0.bias")        # [C, 1, 3, 3]
        C = cw.shape[0]
        ow = f("encoder.embed.out.0.weight")                                             # [D, C * F2]
        self.dim = D = ow.shape[0]
        F2 = ow.shape[1] // C
        self.idim = idim
        if (idim - 1) // 2 != F2:
            raise _C.F5EError(f"PPG extractor: embed.out expects {F2} subsampled bins, input_dim {idim} gives {(idim - 1) // 2}")
        # Conv2d(1, C, 3, stride 2) over [T, idim] as ONE dense GEMM per frame triple: row n = c * F2 + f' of a Toeplitz
        # matrix holds w[c, 0, i, j] at column i * idim + 2 f' + j.  Global CMVN ((x - mean) * istd, per feature) is
        # linear in x, so it folds into these weights and a per-row bias (weights-only transform, done once).
        toe = torch.zeros(C * F2, 3 * idim, device=dv)
        n = torch.arange(C * F2, device=dv)
        c_idx, f_idx = n // F2, n % F2
        for i in range(3):
            for j in range(3):
                toe[n, i * idim + 2 * f_idx + j] = cw[c_idx, 0, i, j]
        bias = cb[c_idx].clone()
        if "encoder.global_cmvn.mean" in sd:
            mean, istd = f("encoder.global_cmvn.mean"), f("encoder.global_cmvn.istd")
            s3, m3 = istd.repeat(3), mean.repeat(3)
            bias = bias - (toe * (s3 * m3)[None, :]).sum(1)
            toe = toe * s3[None, :]
        self.sub_w, self.sub_b = toe.contiguous(), bias.contiguous()
        self.sub_k = 3 * idim
        self.out_w, self.out_b = ow, f("encoder.embed.out.0.bias")
        self.xscale = math.sqrt(D)
        dk = D // heads
        self.layers = []
        i = 0
        while f"encoder.encoders.{i}.norm_mha.weight" in sd:
            p = f"encoder.encoders.{i}."
            wq, bq = f(p + "self_attn.linear_q.weight"), f(p + "self_attn.linear_q.bias")
            u, v = f(p + "self_attn.pos_bias_u").reshape(-1), f(p + "self_attn.pos_bias_v").reshape(-1)
            cm = p + "conv_module."
            dw, db = f(cm + "depthwise_conv.weight")[:, 0, :], f(cm + "depthwise_conv.bias")       # [D, k]
            s = f(cm + "norm.weight") / torch.sqrt(f(cm + "norm.running_var") + 1e-5)             # eval-mode BatchNorm1d
            self.layers.append(dict(
                ln={n_: (f(p + n_ + ".weight"), f(p + n_ + ".bias"))
                    for n_ in ("norm_ff", "norm_mha", "norm_ff_macaron", "norm_conv", "norm_final")},
                ffm=tuple(f(p + "feed_forward_macaron." + n_) for n_ in ("w_1.weight", "w_1.bias", "w_2.weight", "w_2.bias")),
                ff=tuple(f(p + "feed_forward." + n_) for n_ in ("w_1.weight", "w_1.bias", "w_2.weight", "w_2.bias")),
                # (q + pos_bias_u | q + pos_bias_v) = x Wq^T + (bq + u | bq + v): one GEMM with the weight stacked twice
                wq2=torch.cat((wq, wq), 0).contiguous(), bq2=torch.cat((bq + u, bq + v), 0).contiguous(),
                wk=f(p + "self_attn.linear_k.weight"), bk=f(p + "self_attn.linear_k.bias"),
                wv=f(p + "self_attn.linear_v.weight"), bv=f(p + "self_attn.linear_v.bias"),
                wo=f(p + "self_attn.linear_out.weight"), bo=f(p + "self_attn.linear_out.bias"),
                wp=f(p + "self_attn.linear_pos.weight"),
                pw1=f(cm + "pointwise_conv1.weight")[:, :, 0].contiguous(), pb1=f(cm + "pointwise_conv1.bias"),
                dw=(dw * s[:, None]).t().contiguous(),
                db=((db - f(cm + "norm.running_mean")) * s + f(cm + "norm.bias")).contiguous(),
                pw2=f(cm + "pointwise_conv2.weight")[:, :, 0].contiguous(), pb2=f(cm + "pointwise_conv2.bias")))
            i += 1
        self.after = (f("encoder.after_norm.weight"), f("encoder.after_norm.bias"))
        self.lin_w, self.lin_b = f("linear.weight"), f("linear.bias")
        self.ce_w, self.ce_b = f("ce.fc.weight"), f("ce.fc.bias")
        self.half = torch.full((max(D, 1),), 0.5, device=dv)
        self.dk = dk
        self._pe: Dict[int, Tensor] = {}

    def pos_table(self, t: int) -> Tensor:
        """PositionalEncoding.pe[:, :t] (embedding.py:34-46), a constant table built in fp32 like the reference."""
        if t not in self._pe:
            d = self.dim
            pe = torch.zeros(t, d)
            pos = torch.arange(0, t, dtype=F32).unsqueeze(1)
            div = torch.exp(torch.arange(0, d, 2, dtype=F32) * -(math.log(10000.0) / d))
            pe[:, 0::2], pe[:, 1::2] = torch.sin(pos * div), torch.cos(pos * div)
            self._pe = {t: pe.to(self.device).contiguous()}
        return self._pe[t]

    def forward(self, feats: Tensor, lens: Tensor) -> Tuple[Tensor, Tensor]:
        dv, D, H, dk = self.device, self.dim, self.heads, self.dk
        B, T, idim = feats.shape
        if idim != self.idim:
            raise _C.F5EError(f"PPG extractor: features have {idim} bins, the model expects {self.idim}")
        if T < 3:
            raise _C.F5EError("PPG extractor: needs at least 3 feature frames")
        x = feats.to(dv, F32).contiguous()
        T2 = (T - 3) // 2 + 1
        lens_h = lens.detach().to("cpu", torch.long)
        len2 = torch.tensor([int(((torch.arange(0, T - 2, 2)) < int(n)).sum()) for n in lens_h], dtype=I32)  # mask[:, :, :-2:2]
        M = B * T2
        # --- Conv2dSubsampling2 (+ folded CMVN): patches of 3 frames at stride 2 -> ReLU(Toeplitz GEMM) -> Linear
        col = torch.empty(B, T, self.sub_k, device=dv)
        ops.im2col(x, col, 3, 0)
        patches = col.view(B * T, self.sub_k)
        h = torch.empty(M, self.sub_w.shape[0], device=dv)
        for b in range(B):
            ops.gemm_f32(patches[b * T:(b + 1) * T:2], self.sub_w, self.sub_b, out=h[b * T2:(b + 1) * T2], M=T2,
                         act=ops.ACT_RELU)
        xs = torch.empty(M, D, device=dv)
        ops.gemm_f32(h, self.out_w, self.out_b, out=xs)
        ops.axpby(xs, None, xs, self.xscale, 0.0, 0.0)                         # RelPositionalEncoding: x * sqrt(d)
        pos = self.pos_table(T2)
        # masks whenever ANY item is shorter than the padded length (reference BaseEncoder always builds them from xs_lens,
        # ppg/wenet/transformer/encoder.py: also for a single padded utterance); all-full batches need none
        ragged = bool(int(len2.min()) < T2)
        kv_len = len2.to(dv) if ragged else None
        keep = None
        if ragged:
            keep = (torch.arange(T2)[None, :] < len2[:, None].long()).to(F32).to(dv).contiguous()      # mask_pad
        keep_flat = keep.view(-1) if keep is not None else None
        Tp = (T2 + 3) // 4 * 4
        hn = torch.empty(M, D, device=dv)
        hm = torch.empty(M, D, device=dv) if ragged else None
        units = self.layers[0]["ffm"][0].shape[0] if self.layers else D
        mid = torch.empty(M, units, device=dv)
        qu = torch.empty(M, 2 * D, device=dv)
        kb = torch.empty(M, D, device=dv)
        pb = torch.empty(T2, D, device=dv)
        vt = torch.zeros(D, Tp, device=dv)
        sc = torch.empty(T2, Tp, device=dv)
        pr = torch.empty(T2, Tp, device=dv)
        ctx = torch.empty(M, D, device=dv)
        pw = torch.empty(M, 2 * D, device=dv)
        gl = torch.empty(B, T2, D, device=dv)
        dwo = torch.empty(B, T2, D, device=dv)
        scale = 1.0 / math.sqrt(dk)
        for L in self.layers:
            ln = L["ln"]
            # macaron feed-forward: x += 0.5 * W2 swish(W1 LN(x))
            ops.layernorm(xs, hn, gamma=ln["norm_ff_macaron"][0], beta=ln["norm_ff_macaron"][1], eps=1e-5)
            ops.gemm_f32(hn, L["ffm"][0], L["ffm"][1], out=mid, act=ops.ACT_SILU)
            ops.gemm_f32(mid, L["ffm"][2], L["ffm"][3], out=xs, ch_scale=self.half[:D], addend=xs)
            # relative-position self-attention (attention.py:172-222), one sequence and head at a time on the fp32 GEMM
            ops.layernorm(xs, hn, gamma=ln["norm_mha"][0], beta=ln["norm_mha"][1], eps=1e-5)
            ops.gemm_f32(hn, L["wq2"], L["bq2"], out=qu)
            ops.gemm_f32(hn, L["wk"], L["bk"], out=kb)
            ops.gemm_f32(pos, L["wp"], None, out=pb)
            for b in range(B):
                r0, r1 = b * T2, (b + 1) * T2
                # V^T [D, T2] = Wv . LN(x)^T (bias added after P.V: softmax rows sum to one)
                ops.gemm_f32(L["wv"], hn[r0:r1], None, out=vt[:, :T2])
                for hd in range(H):
                    c0, c1 = hd * dk, (hd + 1) * dk
                    ops.gemm_f32(qu[r0:r1, c0:c1], kb[r0:r1, c0:c1], None, out=sc[:, :T2])               # (q + u) k^T
                    ops.gemm_f32(qu[r0:r1, D + c0:D + c1], pb[:, c0:c1], None, out=sc[:, :T2], addend=sc[:, :T2])  # + (q + v) p^T
                    ops.softmax_rows(sc, pr, T2, scale, kv_len=kv_len[b:b + 1] if kv_len is not None else None,
                                     rows_per_seq=T2)
                    ops.gemm_f32(pr, vt[c0:c1], L["bv"][c0:c1], out=ctx[r0:r1, c0:c1], K=Tp)
            ops.gemm_f32(ctx, L["wo"], L["bo"], out=xs, addend=xs)
            # convolution module (convolution.py:84-133): mask, pointwise -> GLU -> depthwise (+BN) -> swish -> pointwise, mask
            ops.layernorm(xs, hn, gamma=ln["norm_conv"][0], beta=ln["norm_conv"][1], eps=1e-5)
            if keep_flat is not None:
                # the reference zeroes padded frames BEFORE pointwise_conv1 (whose bias then makes them non-zero again for the
                # depthwise conv's neighbours): mask LN(x) first (row_scale acts on the output side, hence the identity GEMM)
                ops.gemm_f32(hn, self._eye(D), None, out=hm, row_scale=keep_flat)
                ops.gemm_f32(hm, L["pw1"], L["pb1"], out=pw)
            else:
                ops.gemm_f32(hn, L["pw1"], L["pb1"], out=pw)
            ops.glu(pw, gl.view(M, D))
            ops.dwconv(gl, L["dw"], L["db"], dwo)
            ops.gemm_f32(dwo.view(M, D), L["pw2"], L["pb2"], out=hn, a_act=ops.ACT_SILU, row_scale=keep_flat)
            ops.axpby(xs, hn, xs, 1.0, 1.0)
            # feed-forward + final norm
            ops.layernorm(xs, hn, gamma=ln["norm_ff"][0], beta=ln["norm_ff"][1], eps=1e-5)
            ops.gemm_f32(hn, L["ff"][0], L["ff"][1], out=mid, act=ops.ACT_SILU)
            ops.gemm_f32(mid, L["ff"][2], L["ff"][3], out=xs, ch_scale=self.half[:D], addend=xs)
            ops.layernorm(xs, xs, gamma=ln["norm_final"][0], beta=ln["norm_final"][1], eps=1e-5)
        ops.layernorm(xs, hn, gamma=self.after[0], beta=self.after[1], eps=1e-5)
        ppg = torch.empty(M, D, device=dv)
        ops.gemm_f32(hn, self.lin_w, self.lin_b, out=ppg)
        logits = torch.empty(M, self.ce_w.shape[0], device=dv)
        ops.gemm_f32(ppg, self.ce_w, self.ce_b, out=logits)
        return ppg.view(B, T2, D), logits

    def _eye(self, d: int) -> Tensor:
        if getattr(self, "_eye_t", None) is None or self._eye_t.shape[0] != d:
            self._eye_t = torch.eye(d, device=self.device)
        return self._eye_t


# ------------------------------------------------------------------ features + wrapper (reference ppg/ppg_model.py)

def _kaldi_mel_banks(num_bins: int, padded: int, sr: float, low: float = 20.0, high: float = 0.0) -> Tensor:
    """torchaudio.compliance.kaldi.get_mel_banks (vtln_warp 1) -> [padded / 2 + 1, num_bins] (Nyquist row zero)."""
    nyq = 0.5 * sr
    high = high + nyq if high <= 0.0 else high
    mel_low, mel_high = 1127.0 * math.log(1.0 + low / 700.0), 1127.0 * math.log(1.0 + high / 700.0)
    delta = (mel_high - mel_low) / (num_bins + 1)
    b = torch.arange(num_bins).unsqueeze(1)
    left, center, right = mel_low + b * delta, mel_low + (b + 1.0) * delta, mel_low + (b + 2.0) * delta
    m = (1127.0 * torch.log(1.0 + (sr / padded) * torch.arange(padded // 2) / 700.0)).unsqueeze(0)
    bins = torch.max(torch.zeros(1), torch.min((m - left) / (center - left), (right - m) / (right - center)))
    return torch.nn.functional.pad(bins, (0, 1)).t().contiguous()


class kaldiFbank(nn.Module):
    """reference ppg/wenet/dataset/feats.py:49-83 (kaldi.fbank per utterance: 80 bins, 25 ms / 10 ms, dither 0)."""

    def __init__(self, sample_rate=16000, n_fft=512, win_length=400, hop_length=160, n_mels=80, spec_mask_time=[5, 10],
                 spec_mask_freq=[5, 10]):
        super().__init__()
        self.f_length, self.f_shift = int(win_length / sample_rate * 1000), int(hop_length / sample_rate * 1000)
        self.sample_rate, self.n_fft, self.n_mels = sample_rate, n_fft, n_mels
        self.win, self.shift = int(sample_rate * self.f_length * 0.001), int(sample_rate * self.f_shift * 0.001)
        if (1 << (self.win - 1).bit_length()) != 512:
            raise _C.F5EError("kaldiFbank: the HIP kernel is built for a 512-point padded window (25 ms at 16 kHz)")
        k = torch.arange(256, dtype=torch.float64)
        self.register_buffer("window", torch.hann_window(self.win, periodic=False).pow(0.85), persistent=False)
        self.register_buffer("twiddle", torch.stack((torch.cos(2 * math.pi * k / 512), -torch.sin(2 * math.pi * k / 512)),
                                                    -1).float(), persistent=False)
        self.register_buffer("fb", _kaldi_mel_banks(n_mels, 512, float(sample_rate)), persistent=False)

    @torch.no_grad()
    def forward(self, x: Tensor, is_spec_aug=[]):
        if len(is_spec_aug) != 0:
            raise _C.F5EError("kaldiFbank: spec augmentation is a training feature (out of scope)")
        if self.window.device.type != "cuda":
            self.to(x.device if x.is_cuda else "cuda")
        wav = x.to(self.window.device, F32).contiguous()
        T = 1 + (wav.shape[1] - self.win) // self.shift
        out = torch.empty(wav.shape[0], T, self.n_mels, device=wav.device)
        ops.kaldi_fbank(wav, self.window, self.twiddle, self.fb, out, self.win, self.shift)
        return out, torch.tensor([T])


def build_ppg_model(ppg_model_path, ppg_config, device="cpu"):
    """reference ppg/ppg_model.py:11-29: yaml -> model (cmvn path fallback next to the checkpoint) -> checkpoint keys that
    exist in the model are loaded, the rest (decoder, CTC) ignored."""
    import yaml
    with open(ppg_config, "r") as fin:
        ppg_configs = yaml.safe_load(fin)
    if ppg_configs.get("cmvn_file") is not None and not os.path.exists(ppg_configs["cmvn_file"]):
        old = ppg_configs["cmvn_file"]
        ppg_configs["cmvn_file"] = os.path.join(os.path.dirname(ppg_model_path), "global_cmvn")
        print(f"{old} not exist, use {ppg_configs['cmvn_file']}")
    model = ConformerPPG.from_config(ppg_configs)
    checkpoint = torch.load(ppg_model_path, map_location="cpu", weights_only=True)
    model_dict = model.state_dict()
    model_dict.update({k: v for k, v in checkpoint.items() if k in model_dict})
    model.load_state_dict(model_dict)
    return model.to(device).eval()


def make_pad_mask(lengths: Tensor, max_len: int = 0) -> Tensor:
    """reference ppg/ppg_model.py:31-56."""
    max_len = max_len if max_len > 0 else int(lengths.max().item())
    return torch.arange(0, max_len, dtype=torch.int64, device=lengths.device)[None, :] >= lengths.unsqueeze(-1)


class PPGModelWapper(object):
    """reference ppg/ppg_model.py:58-168 (name as spelled there)."""

    def __init__(self, ppg_model_path, ppg_config, device, output_type="ppg", map_mix_ratio=1.0, ppg_frame_length=20,
                 mel_f_shift=10, global_phn_center_path=None, para_softmax_path=None):
        print(f"loading ppg model from {ppg_model_path}")
        print(f"output_type : {output_type}")
        self.ppg_model = build_ppg_model(ppg_model_path, ppg_config, device)
        self.device, self.output_type, self.map_mix_ratio = device, output_type, map_mix_ratio
        self.ppg_frame_length, self.mel_f_shift = ppg_frame_length, mel_f_shift
        self.featCal = kaldiFbank().eval()
        if self.output_type == "map":
            import pickle
            self.global_phn_center = torch.from_numpy(np.load(global_phn_center_path)).to(device).to(F32)
            with open(para_softmax_path, "rb") as f:
                para = pickle.load(f)
            self.para_softmax = {"w": torch.from_numpy(para["w"]).to(device).float().contiguous(),
                                 "b": torch.from_numpy(para["b"]).to(device).float().contiguous()}

    @staticmethod
    def norm_ppg(ppg, length):
        raise NotImplementedError("norm_ppg is dead code in the reference (commented out at ppg_model.py:121,128)")

    def ppg_to_target(self, ppg, true_len):
        """reference :112-131: optional map through the phone posteriors' centres, then zero the padded frames."""
        B, T, D = ppg.shape
        keep = (~make_pad_mask(true_len.to(ppg.device), T)).to(F32).reshape(-1).contiguous()
        if self.output_type == "map":
            w, bb, cen = self.para_softmax["w"], self.para_softmax["b"], self.global_phn_center
            logit = torch.empty(B * T, w.shape[0], device=ppg.device)
            ops.gemm_f32(ppg.reshape(B * T, D), w, bb, out=logit)
            Lp = (w.shape[0] + 3) // 4 * 4
            soft = torch.zeros(B * T, Lp, device=ppg.device)
            ops.softmax_rows(logit, soft, w.shape[0], 1.0)
            cen_t = torch.zeros(cen.shape[1], Lp, device=ppg.device)
            cen_t[:, :cen.shape[0]] = cen.t()
            mapped = torch.empty(B * T, cen.shape[1], device=ppg.device)
            ops.gemm_f32(soft, cen_t, None, out=mapped)
            if self.map_mix_ratio != 1.0:
                ops.axpby(ppg.reshape(B * T, D).contiguous(), mapped, mapped, 1 - self.map_mix_ratio, self.map_mix_ratio)
            src = mapped
        elif self.output_type == "ppg":
            src = ppg.reshape(B * T, D).contiguous()
        else:
            raise _C.F5EError(f"unknown output_type {self.output_type!r}")
        out = torch.empty(B * T, src.shape[1], device=ppg.device)
        ops.gemm_f32(src, self._eye(src.shape[1], ppg.device), None, out=out, row_scale=keep)
        return out.view(B, T, -1)

    def _eye(self, d, dv):
        if getattr(self, "_eye_t", None) is None or self._eye_t.shape[0] != d:
            self._eye_t = torch.eye(d, device=dv)
        return self._eye_t

    @torch.no_grad()
    def mel_to_ppg(self, mel, mel_lens):
        ppg, _logits = self.ppg_model.extract(mel, mel_lens, stream=False)
        true_len = (mel_lens.to("cpu") / (self.ppg_frame_length / self.mel_f_shift)).long().clamp(max=ppg.shape[1])
        return self.ppg_to_target(ppg, true_len), true_len.to(ppg.device)

    @torch.no_grad()
    def audio_to_mel(self, audio, sr=None):
        """audio: wav path or [1, l] tensor (reference :142-158)."""
        from ..infer import audio as A
        from ..infer.utils_infer import load_wav
        if isinstance(audio, str):
            audio, sr = load_wav(audio)
        if audio.ndim == 1:
            audio = audio.unsqueeze(0)
        if sr != 16000:
            audio = A.resample(audio.cpu(), sr, 16000)
        feats, feats_len = self.featCal(audio.to(self.device))
        return feats, feats_len.to(self.device)

    @torch.no_grad()
    def audio_to_ppg(self, audio, sr=None):
        feats, feats_len = self.audio_to_mel(audio, sr)
        return self.mel_to_ppg(feats, feats_len)
