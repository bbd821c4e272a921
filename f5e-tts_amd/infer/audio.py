"""Host-side audio glue either side of the hot path: sample-rate conversion and silence clipping.

The reference delegates these to third-party packages that are absent here and NOT vendored in /root/reference:
  * torchaudio.transforms.Resample (default: sinc_interp_hann, lowpass_filter_width 6, rolloff 0.99) at
    infer/utils_infer.py:445-447 and ppg/ppg_model.py:156-158;
  * pydub.AudioSegment / pydub.silence (detect_leading_silence, split_on_silence, dBFS) at
    infer/utils_infer.py:273-334 (preprocess_ref_audio_text, remove_silence_edges) and :567-575.
Both are restated from their published algorithms; PARITY UNPINNED (no reference fixture exists for either — SURVEY
8c).  Everything here is once-per-reference-clip CPU work on 16-bit PCM, as it is in the reference (the audio is
still on the host at those call sites); nothing here is on the measured path.
"""
from __future__ import annotations

import math
from typing import List, Sequence, Tuple

import numpy as np
import torch
import torch.nn.functional as F

# ----------------------------------------------------------------------------- resampling


def sinc_resample_kernel(orig_freq: int, new_freq: int, lowpass_filter_width: int = 6, rolloff: float = 0.99):
    """Polyphase windowed-sinc bank [new, 1, 2*width + orig] (hann window), torchaudio's default resampling method."""
    g = math.gcd(int(orig_freq), int(new_freq))
    orig, new = int(orig_freq) // g, int(new_freq) // g
    base_freq = min(orig, new) * rolloff
    width = math.ceil(lowpass_filter_width * orig / base_freq)
    idx = torch.arange(-width, width + orig, dtype=torch.float64)[None, None] / orig
    t = torch.arange(0, -new, -1, dtype=torch.float64)[:, None, None] / new + idx
    t = (t * base_freq).clamp_(-lowpass_filter_width, lowpass_filter_width)
    window = torch.cos(t * math.pi / lowpass_filter_width / 2) ** 2
    t = t * math.pi
    scale = base_freq / orig
    kernels = torch.where(t == 0, torch.tensor(1.0, dtype=torch.float64), t.sin() / t) * window * scale
    return kernels.to(torch.float32), width, orig, new


def resample(waveform: torch.Tensor, orig_freq: int, new_freq: int) -> torch.Tensor:
    """[..., n] at orig_freq -> [..., ceil(n * new / orig)] at new_freq."""
    if int(orig_freq) == int(new_freq):
        return waveform
    kernels, width, orig, new = sinc_resample_kernel(orig_freq, new_freq)
    shape = waveform.shape
    x = waveform.reshape(-1, shape[-1]).to(torch.float32)
    length = x.shape[-1]
    x = F.pad(x, (width, width + orig))
    y = F.conv1d(x[:, None], kernels.to(x.device), stride=orig)          # [b, new, frames]
    y = y.transpose(1, 2).reshape(x.shape[0], -1)
    target = int(math.ceil(new * length / orig))
    return y[..., :target].reshape(shape[:-1] + (target,))


# ----------------------------------------------------------------------------- 16-bit segments with millisecond slicing


class Segment:
    """Mono/stereo int16 PCM with pydub's millisecond indexing (frames = int(ms * rate / 1000))."""

    MAX_AMP = 32768.0

    def __init__(self, pcm: np.ndarray, rate: int):
        pcm = np.asarray(pcm)
        if pcm.ndim == 1:
            pcm = pcm[:, None]
        self.pcm = pcm.astype(np.int16, copy=False)    # [frames, channels]
        self.rate = int(rate)

    @classmethod
    def from_float(cls, wav: np.ndarray, rate: int) -> "Segment":
        """float [channels, n] or [n] in [-1, 1] -> 16-bit."""
        w = np.asarray(wav, dtype=np.float64)
        if w.ndim == 2:
            w = w.T
        return cls(np.clip(np.round(w * 32767.0), -32768, 32767).astype(np.int16), rate)

    @classmethod
    def silent(cls, duration_ms: int, rate: int, channels: int = 1) -> "Segment":
        return cls(np.zeros((int(rate * duration_ms / 1000.0), channels), np.int16), rate)

    def to_float(self) -> np.ndarray:
        return (self.pcm.astype(np.float32) / 32768.0).T                   # [channels, n]

    def __len__(self) -> int:                                               # milliseconds, rounded like pydub
        return int(round(1000.0 * self.pcm.shape[0] / self.rate))

    @property
    def duration_seconds(self) -> float:
        return self.pcm.shape[0] / self.rate if self.rate else 0.0

    def _frame(self, ms: float) -> int:
        return int(ms * (self.rate / 1000.0))

    def slice_ms(self, start, end) -> "Segment":
        n = len(self)
        start = 0 if start is None else (max(n + start, 0) if start < 0 else min(start, n))
        end = n if end is None else (max(n + end, 0) if end < 0 else min(end, n))
        return Segment(self.pcm[self._frame(start):self._frame(end)], self.rate)

    def __getitem__(self, s) -> "Segment":
        if isinstance(s, slice):
            return self.slice_ms(s.start, s.stop)
        return self.slice_ms(s, s + 1)

    def __add__(self, other: "Segment") -> "Segment":
        if other.pcm.shape[0] == 0:
            return Segment(self.pcm, self.rate)
        if self.pcm.shape[0] == 0:
            return Segment(other.pcm, other.rate)
        assert self.rate == other.rate and self.pcm.shape[1] == other.pcm.shape[1]
        return Segment(np.concatenate([self.pcm, other.pcm]), self.rate)

    @property
    def rms(self) -> int:
        """audioop.rms: integer sqrt of the mean square over all samples of all channels."""
        if self.pcm.size == 0:
            return 0
        x = self.pcm.astype(np.int64).reshape(-1)
        return int(math.sqrt(float(np.dot(x, x)) / x.size))

    @property
    def dBFS(self) -> float:
        r = self.rms
        return -float("inf") if r == 0 else 20.0 * math.log10(r / self.MAX_AMP)


def detect_silence(seg: Segment, min_silence_len=1000, silence_thresh=-16, seek_step=1) -> List[List[int]]:
    seg_len = len(seg)
    if seg_len < min_silence_len:
        return []
    thresh = (10 ** (silence_thresh / 20.0)) * Segment.MAX_AMP
    last = seg_len - min_silence_len
    starts = list(range(0, last + 1, seek_step))
    if last % seek_step:
        starts.append(last)
    # rms of every window from prefix sums of the squared samples (pydub slices and calls audioop.rms each time)
    sq = seg.pcm.astype(np.int64).reshape(seg.pcm.shape[0], -1)
    csum = np.concatenate([[0], np.cumsum((sq * sq).sum(axis=1))])
    ch = seg.pcm.shape[1]
    silence_starts = []
    for i in starts:
        f0, f1 = seg._frame(i), seg._frame(min(i + min_silence_len, seg_len))
        n = (f1 - f0) * ch
        r = int(math.sqrt(float(csum[f1] - csum[f0]) / n)) if n > 0 else 0
        if r <= thresh:
            silence_starts.append(i)
    if not silence_starts:
        return []
    ranges = []
    prev = silence_starts.pop(0)
    cur = prev
    for s in silence_starts:
        continuous = s == prev + seek_step
        has_gap = s > prev + min_silence_len
        if not continuous and has_gap:
            ranges.append([cur, prev + min_silence_len])
            cur = s
        prev = s
    ranges.append([cur, prev + min_silence_len])
    return ranges


def detect_nonsilent(seg: Segment, min_silence_len=1000, silence_thresh=-16, seek_step=1) -> List[List[int]]:
    silent = detect_silence(seg, min_silence_len, silence_thresh, seek_step)
    n = len(seg)
    if not silent:
        return [[0, n]]
    if silent[0][0] == 0 and silent[0][1] == n:
        return []
    prev_end, out = 0, []
    end_i = 0
    for start_i, end_i in silent:
        out.append([prev_end, start_i])
        prev_end = end_i
    if end_i != n:
        out.append([prev_end, n])
    if out[0] == [0, 0]:
        out.pop(0)
    return out


def split_on_silence(seg: Segment, min_silence_len=1000, silence_thresh=-16, keep_silence=100, seek_step=1) -> List[Segment]:
    if isinstance(keep_silence, bool):
        keep_silence = len(seg) if keep_silence else 0
    ranges = [[s - keep_silence, e + keep_silence]
              for s, e in detect_nonsilent(seg, min_silence_len, silence_thresh, seek_step)]
    for a, b in zip(ranges, ranges[1:]):
        if b[0] < a[1]:
            a[1] = (a[1] + b[0]) // 2
            b[0] = a[1]
    return [seg[max(s, 0):min(e, len(seg))] for s, e in ranges]


def detect_leading_silence(seg: Segment, silence_threshold=-50.0, chunk_size=10) -> int:
    trim, n = 0, len(seg)
    while trim < n and seg[trim:trim + chunk_size].dBFS < silence_threshold:
        trim += chunk_size
    return min(trim, n)


def remove_silence_edges(seg: Segment, silence_threshold=-42) -> Segment:
    """Reference infer/utils_infer.py:273-287: leading silence in 10 ms chunks, trailing silence per millisecond."""
    seg = seg[detect_leading_silence(seg, silence_threshold):]
    end = seg.duration_seconds
    for ms in range(len(seg) - 1, -1, -1):
        if seg[ms].dBFS > silence_threshold:
            break
        end -= 0.001
    return seg[:int(end * 1000)]


def _join_until_12s(segs: Sequence[Segment], rate: int, channels: int, note, tag: str) -> Segment:
    out = Segment.silent(0, rate, channels)
    for s in segs:
        if len(out) > 6000 and len(out + s) > 12000:
            note(f"Audio is over 12s, clipping short. ({tag})")
            break
        out = out + s
    return out


def clip_reference(seg: Segment, clip_short: bool = True, note=lambda m: None) -> Segment:
    """The audio half of preprocess_ref_audio_text (reference infer/utils_infer.py:295-330): clip to <= 12 s at a long
    silence, else at a short one, else hard; strip silent edges; append 50 ms of silence."""
    ch = seg.pcm.shape[1]
    if clip_short:
        out = _join_until_12s(split_on_silence(seg, 1000, -50, 1000, 10), seg.rate, ch, note, "1")
        if len(out) > 12000:
            out = _join_until_12s(split_on_silence(seg, 100, -40, 1000, 10), seg.rate, ch, note, "2")
        seg = out
        if len(seg) > 12000:
            seg = seg[:12000]
            note("Audio is over 12s, clipping short. (3)")
    return remove_silence_edges(seg) + Segment.silent(50, seg.rate, ch)


def strip_generated_silence(seg: Segment) -> Segment:
    """remove_silence_for_generated_wav (reference infer/utils_infer.py:567-575)."""
    out = Segment.silent(0, seg.rate, seg.pcm.shape[1])
    for s in split_on_silence(seg, min_silence_len=1000, silence_thresh=-50, keep_silence=500, seek_step=10):
        out = out + s
    return out
