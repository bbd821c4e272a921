"""CPU oracle for the PPG extractor that feeds BASELINE config 5: kaldi fbank -> wenet Conformer encoder
-> 256-d "content embedding" head, and the ``PPGModelWapper.mel_to_ppg`` glue.

TEST INFRASTRUCTURE ONLY (same rule as f5e_oracle.py): imported by ``tests/`` only, never by the product package.

Plain PyTorch fp32 restatement over a flat state dict with the reference's key names (``ASRModel.state_dict()``:
``encoder.embed.*``, ``encoder.encoders.{i}.*``, ``encoder.after_norm.*``, ``linear.*``, ``ce.fc.*``,
``encoder.global_cmvn.{mean,istd}``).  Paths below are relative to ``/root/reference/src/f5_tts/ppg``.

Pinning: the encoder / head / glue are pinned by ``tests/golden/ppg_*.npz`` produced from the reference's own classes
(``tests/golden/make_golden.py ppg``).  ``kaldi_fbank`` restates ``torchaudio.compliance.kaldi.fbank`` (third-party, not
in the tree, absent from the image): PARITY UNPINNED for that function.
"""
from __future__ import annotations

import math
from typing import Dict, Optional, Tuple

import torch
import torch.nn.functional as F

Tensor = torch.Tensor
State = Dict[str, Tensor]


# ------------------------------------------------------------------ kaldi fbank (torchaudio.compliance.kaldi, UNPINNED)

def kaldi_mel_banks(num_bins: int = 80, padded: int = 512, sr: float = 16000.0, low: float = 20.0, high: float = 0.0) -> Tensor:
    """get_mel_banks (vtln_warp = 1) -> [num_bins, padded // 2 + 1] (last column zero: the Nyquist bin is dropped)."""
    nfft_bins = padded // 2
    nyq = 0.5 * sr
    if high <= 0.0:
        high += nyq
    width = sr / padded
    mel = lambda f: 1127.0 * torch.log(1.0 + f / 700.0)          # noqa: E731
    mel_low, mel_high = 1127.0 * math.log(1.0 + low / 700.0), 1127.0 * math.log(1.0 + high / 700.0)
    delta = (mel_high - mel_low) / (num_bins + 1)
    b = torch.arange(num_bins).unsqueeze(1)
    left, center, right = mel_low + b * delta, mel_low + (b + 1.0) * delta, mel_low + (b + 2.0) * delta
    m = mel(width * torch.arange(nfft_bins)).unsqueeze(0)
    up, down = (m - left) / (center - left), (right - m) / (right - center)
    bins = torch.max(torch.zeros(1), torch.min(up, down))
    return F.pad(bins, (0, 1))


def povey_window(n: int = 400) -> Tensor:
    return torch.hann_window(n, periodic=False).pow(0.85)


def kaldi_fbank(wav: Tensor, num_mel_bins: int = 80, frame_length: float = 25.0, frame_shift: float = 10.0,
                sample_frequency: float = 16000.0, preemph: float = 0.97) -> Tensor:
    """kaldi.fbank(wav[1, n] * 2^15, num_mel_bins, frame_length, frame_shift, dither=0, energy_floor=0, sample_frequency)
    as called at wenet/dataset/feats.py:66-72 -> [frames, num_mel_bins].  ``wav`` is the raw [-1, 1] wave [n]."""
    x = wav.reshape(-1).float() * float(1 << 15)
    shift, win = int(sample_frequency * frame_shift * 0.001), int(sample_frequency * frame_length * 0.001)
    padded = 1 << (win - 1).bit_length()
    m = 1 + (x.numel() - win) // shift
    frames = x.unfold(0, win, shift)[:m]                               # snip_edges
    frames = frames - frames.mean(dim=1, keepdim=True)                 # remove_dc_offset
    prev = F.pad(frames.unsqueeze(0), (1, 0), mode="replicate").squeeze(0)[:, :-1]
    frames = (frames - preemph * prev) * povey_window(win)
    frames = F.pad(frames, (0, padded - win))
    spec = torch.fft.rfft(frames).abs().pow(2.0)
    fb = kaldi_mel_banks(num_mel_bins, padded, sample_frequency)
    return torch.max(spec @ fb.T, torch.tensor(torch.finfo(torch.float32).eps)).log()


# ------------------------------------------------------------------ wenet Conformer encoder (wenet/transformer/*)

def rel_pos_table(t: int, d: int) -> Tensor:
    """PositionalEncoding.pe[:, :t] (embedding.py:34-46): sin on even, cos on odd channels."""
    pe = torch.zeros(t, d)
    pos = torch.arange(0, t, dtype=torch.float32).unsqueeze(1)
    div = torch.exp(torch.arange(0, d, 2, dtype=torch.float32) * -(math.log(10000.0) / d))
    pe[:, 0::2] = torch.sin(pos * div)
    pe[:, 1::2] = torch.cos(pos * div)
    return pe.unsqueeze(0)


def conv2d_subsampling2(sd: State, p: str, x: Tensor, mask: Tensor) -> Tuple[Tensor, Tensor, Tensor]:
    """Conv2dSubsampling2.forward (subsampling.py:96-118) + RelPositionalEncoding (embedding.py:96-109)."""
    h = F.relu(F.conv2d(x.unsqueeze(1), sd[p + "conv.0.weight"], sd[p + "conv.0.bias"], stride=2))
    b, c, t, f = h.shape
    h = F.linear(h.transpose(1, 2).contiguous().view(b, t, c * f), sd[p + "out.0.weight"], sd[p + "out.0.bias"])
    d = h.shape[-1]
    return h * math.sqrt(d), rel_pos_table(t, d), mask[:, :, :-2:2]


def rel_mha(sd: State, p: str, x: Tensor, mask: Tensor, pos_emb: Tensor, heads: int) -> Tensor:
    """RelPositionMultiHeadedAttention.forward (attention.py:172-222; rel_shift removed upstream)."""
    b, t, d = x.shape
    dk = d // heads
    lin = lambda n, v: F.linear(v, sd[p + n + ".weight"], sd.get(p + n + ".bias"))   # noqa: E731
    q = lin("linear_q", x).view(b, t, heads, dk)
    k = lin("linear_k", x).view(b, t, heads, dk).transpose(1, 2)
    v = lin("linear_v", x).view(b, t, heads, dk).transpose(1, 2)
    pp = lin("linear_pos", pos_emb).view(pos_emb.shape[0], -1, heads, dk).transpose(1, 2)
    qu = (q + sd[p + "pos_bias_u"]).transpose(1, 2)
    qv = (q + sd[p + "pos_bias_v"]).transpose(1, 2)
    scores = (qu @ k.transpose(-2, -1) + qv @ pp.transpose(-2, -1)) / math.sqrt(dk)
    m = mask.unsqueeze(1).eq(0)
    attn = torch.softmax(scores.masked_fill(m, -float("inf")), dim=-1).masked_fill(m, 0.0)
    ctx = (attn @ v).transpose(1, 2).contiguous().view(b, t, d)
    return lin("linear_out", ctx)


def conv_module(sd: State, p: str, x: Tensor, mask_pad: Tensor) -> Tensor:
    """ConvolutionModule.forward, batch-norm variant, non-causal (convolution.py:84-133)."""
    h = x.transpose(1, 2).masked_fill(~mask_pad, 0.0)
    h = F.glu(F.conv1d(h, sd[p + "pointwise_conv1.weight"], sd[p + "pointwise_conv1.bias"]), dim=1)
    w = sd[p + "depthwise_conv.weight"]
    h = F.conv1d(h, w, sd[p + "depthwise_conv.bias"], padding=(w.shape[-1] - 1) // 2, groups=w.shape[0])
    h = F.batch_norm(h, sd[p + "norm.running_mean"], sd[p + "norm.running_var"], sd[p + "norm.weight"], sd[p + "norm.bias"],
                     training=False, eps=1e-5)
    h = F.conv1d(F.silu(h), sd[p + "pointwise_conv2.weight"], sd[p + "pointwise_conv2.bias"])
    return h.masked_fill(~mask_pad, 0.0).transpose(1, 2)


def conformer_layer(sd: State, p: str, x: Tensor, mask: Tensor, pos_emb: Tensor, mask_pad: Tensor, heads: int) -> Tensor:
    """ConformerEncoderLayer.forward, normalize_before, macaron, cnn module (encoder_layer.py:199-268)."""
    ln = lambda n, v: F.layer_norm(v, (v.shape[-1],), sd[p + n + ".weight"], sd[p + n + ".bias"], eps=1e-5)   # noqa: E731
    ff = lambda n, v: F.linear(F.silu(F.linear(v, sd[p + n + ".w_1.weight"], sd[p + n + ".w_1.bias"])),           # noqa: E731
                               sd[p + n + ".w_2.weight"], sd[p + n + ".w_2.bias"])
    x = x + 0.5 * ff("feed_forward_macaron", ln("norm_ff_macaron", x))
    x = x + rel_mha(sd, p + "self_attn.", ln("norm_mha", x), mask, pos_emb, heads)
    x = x + conv_module(sd, p + "conv_module.", ln("norm_conv", x), mask_pad)
    x = x + 0.5 * ff("feed_forward", ln("norm_ff", x))
    return ln("norm_final", x)


def encoder_depth(sd: State) -> int:
    n = 0
    while f"encoder.encoders.{n}.norm_mha.weight" in sd:
        n += 1
    return n


def conformer_encoder(sd: State, feats: Tensor, lens: Tensor, heads: int = 4) -> Tuple[Tensor, Tensor]:
    """BaseEncoder.forward with decoding_chunk_size = -1, static chunk 0 (encoder.py:141-209) -> (xs [B, T', D], masks)."""
    t = feats.shape[1]
    masks = (torch.arange(t)[None, :] < lens[:, None]).unsqueeze(1)
    x = feats
    if "encoder.global_cmvn.mean" in sd:
        x = (x - sd["encoder.global_cmvn.mean"]) * sd["encoder.global_cmvn.istd"]
    x, pos_emb, masks = conv2d_subsampling2(sd, "encoder.embed.", x, masks)
    for i in range(encoder_depth(sd)):
        x = conformer_layer(sd, f"encoder.encoders.{i}.", x, masks, pos_emb, masks, heads)
    return F.layer_norm(x, (x.shape[-1],), sd["encoder.after_norm.weight"], sd["encoder.after_norm.bias"], eps=1e-5), masks


def asr_extract(sd: State, feats: Tensor, lens: Tensor, heads: int = 4) -> Tuple[Tensor, Tensor]:
    """ASRModel.extract(stream=False) (asr_model.py:221-244) -> (ppg [B, T', D], logits [B*T', vocab+1])."""
    enc, _ = conformer_encoder(sd, feats, lens, heads)
    out = F.linear(enc, sd["linear.weight"], sd["linear.bias"])
    return out, F.linear(out.flatten(0, 1), sd["ce.fc.weight"], sd["ce.fc.bias"])


def mel_to_ppg(sd: State, mel: Tensor, mel_lens: Tensor, heads: int = 4, ppg_frame_length: int = 20,
               mel_f_shift: int = 10) -> Tuple[Tensor, Tensor]:
    """PPGModelWapper.mel_to_ppg, output_type "ppg" (ppg_model.py:112-140): zero the frames past true_len."""
    ppg, _ = asr_extract(sd, mel, mel_lens, heads)
    true_len = (mel_lens / (ppg_frame_length / mel_f_shift)).long().clamp(max=ppg.shape[1])
    keep = (torch.arange(int(true_len.max()))[None, :] < true_len[:, None])[:, :, None]
    return ppg * F.pad(keep, (0, 0, 0, ppg.shape[1] - keep.shape[1])), true_len
