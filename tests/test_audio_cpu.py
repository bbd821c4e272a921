"""Host audio glue: windowed-sinc resampler and the pydub-style silence logic.  Third-party arithmetic
restated from the published packages (parity unpinned): checked here through properties on synthetic signals."""
import math
import os

import numpy as np
import pytest
import torch

from f5e_tts_amd.infer import audio as A
from f5e_tts_amd.infer import utils_infer as U


def _tone(freq, secs, sr, amp=0.5):
    t = np.arange(int(secs * sr)) / sr
    return (amp * np.sin(2 * np.pi * freq * t)).astype(np.float32)


@pytest.mark.parametrize("orig,new", [(16000, 24000), (48000, 24000), (44100, 24000), (22050, 24000)])
def test_resample_tone_matches_analytic(orig, new):
    x = torch.from_numpy(_tone(440.0, 0.5, orig))[None]
    y = A.resample(x, orig, new)
    assert y.shape == (1, math.ceil(x.shape[1] * new / orig))
    ref = _tone(440.0, 0.5, new)[: y.shape[1]]
    core = slice(200, y.shape[1] - 200)                      # the zero-padded edges ring
    assert np.abs(y[0].numpy()[core] - ref[core]).max() < 2e-3
    assert A.resample(x, orig, orig) is x


def test_resample_removes_content_above_new_nyquist():
    x = torch.from_numpy(_tone(15000.0, 0.25, 48000))[None]  # above 12 kHz: must vanish at 24 kHz
    y = A.resample(x, 48000, 24000)
    assert float(y[0, 200:-200].abs().max()) < 5e-3
    k, width, o, n = A.sinc_resample_kernel(48000, 24000)
    assert (o, n) == (2, 1) and k.shape == (1, 1, 2 * width + 2)
    assert abs(float(k.sum()) - 1.0) < 1e-2                 # unit DC gain


def _speech_like(sr=24000):
    sil = lambda s: np.zeros(int(s * sr), np.float32)        # noqa: E731
    return np.concatenate([sil(0.5), _tone(300, 2.0, sr), sil(1.5), _tone(500, 1.0, sr), sil(0.3)]), sr


def test_segment_ms_indexing_and_dbfs():
    w, sr = _speech_like()
    seg = A.Segment.from_float(w, sr)
    assert len(seg) == 5300 and abs(seg.duration_seconds - 5.3) < 1e-9
    assert len(seg[500:2500]) == 2000 and len(seg[-300:]) == 300 and len(seg[10]) == 1
    assert seg[0:400].dBFS == -float("inf")
    assert abs(seg[600:2400].dBFS - 20 * math.log10(0.5 / math.sqrt(2))) < 0.05   # sine of amplitude 0.5
    assert len(seg[0:100] + seg[100:300]) == 300


def test_silence_detection_and_split():
    w, sr = _speech_like()
    seg = A.Segment.from_float(w, sr)
    assert A.detect_leading_silence(seg, -50.0) == 500
    sil = A.detect_silence(seg, min_silence_len=1000, silence_thresh=-50, seek_step=10)
    assert len(sil) == 1 and abs(sil[0][0] - 2500) <= 10 and abs(sil[0][1] - 4000) <= 10
    non = A.detect_nonsilent(seg, min_silence_len=1000, silence_thresh=-50, seek_step=10)
    assert len(non) == 2 and non[0][0] == 0 and abs(non[1][0] - 4000) <= 10 and non[1][1] == 5300
    parts = A.split_on_silence(seg, min_silence_len=1000, silence_thresh=-50, keep_silence=500, seek_step=10)
    assert len(parts) == 2
    assert abs(len(parts[0]) - 3000) <= 10 and abs(len(parts[1]) - 1800) <= 10   # ranges clipped to the segment
    short = A.detect_silence(seg, min_silence_len=100, silence_thresh=-40, seek_step=10)
    assert len(short) == 3                                                        # head, pause, tail
    edges = A.remove_silence_edges(seg)
    assert abs(len(edges) - 4500) <= 12
    stripped = A.strip_generated_silence(seg)
    assert abs(len(stripped) - 4800) <= 20


def test_clip_reference_caps_at_twelve_seconds():
    sr = 16000
    sil = lambda s: np.zeros(int(s * sr), np.float32)        # noqa: E731
    w = np.concatenate([_tone(300, 7.0, sr), sil(1.2), _tone(400, 6.0, sr), sil(1.2), _tone(500, 6.0, sr)])
    notes = []
    out = A.clip_reference(A.Segment.from_float(w, sr), note=notes.append)
    assert any("clipping short. (1)" in n for n in notes)
    assert 6000 < len(out) <= 12000 + 50
    assert out[len(out) - 50:].dBFS == -float("inf")          # the appended 50 ms of silence
    hard = A.clip_reference(A.Segment.from_float(_tone(300, 15.0, sr), sr), note=notes.append)
    assert any("(3)" in n for n in notes) and abs(len(hard) - 12050) <= 2
    keep = A.clip_reference(A.Segment.from_float(w, sr), clip_short=False)
    assert abs(len(keep) - (21400 + 50)) <= 2


def test_preprocess_ref_audio_text_and_resampled_reference(tmp_path):
    sr = 16000
    w = np.concatenate([np.zeros(3200, np.float32), _tone(300, 2.0, sr), np.zeros(1600, np.float32)])
    src = str(tmp_path / "ref16k.wav")
    U.save_wav(src, w, sr)
    path, text = U.preprocess_ref_audio_text(src, "Some call me nature", show_info=lambda m: None)
    assert text == "Some call me nature. "
    assert U.preprocess_ref_audio_text(src, "Ends with dot.", show_info=lambda m: None)[1] == "Ends with dot. "
    wav, sr2 = U.load_wav(path)
    os.unlink(path)
    assert sr2 == sr and abs(wav.shape[1] / sr - 2.05) < 0.02   # edges stripped, 50 ms appended
    with pytest.raises(ValueError, match="ref_text is empty"):
        U.preprocess_ref_audio_text(src, "  ", show_info=lambda m: None)
    out = str(tmp_path / "gen.wav")
    U.save_wav(out, np.concatenate([_tone(300, 1.0, 24000), np.zeros(48000, np.float32), _tone(300, 1.0, 24000)]), 24000)
    U.remove_silence_for_generated_wav(out)
    g, _ = U.load_wav(out)
    assert abs(g.shape[1] / 24000 - 3.0) < 0.03                  # 2 s pause -> 2 x 500 ms kept
