#!/usr/bin/env python3
"""Random-length sweep of the front / back ends (HIP path) against the oracle: log-mel, Vocos decode, kaldi fbank, the PPG
Conformer.  GPU box only.

    python tools/fuzz_frontend.py [seconds] [seed]"""
import os
import random
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch  # noqa: E402

import test_ppg_gpu as TP  # noqa: E402
from oracle import f5e_oracle as O  # noqa: E402
from oracle import f5e_ppg_oracle as P  # noqa: E402
from tools import synth as SY  # noqa: E402

budget = float(sys.argv[1]) if len(sys.argv) > 1 else 90.0
rng = random.Random(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
torch.set_num_threads(16)

from f5e_tts_amd.model import MelSpec  # noqa: E402
from f5e_tts_amd.ppg import kaldiFbank  # noqa: E402
from f5e_tts_amd.vocoder import Vocos  # noqa: E402

vs = SY.init_vocos_state()
voc = Vocos()
voc.load_state_dict(vs, strict=False)
voc = voc.cuda().eval()
melspec, fbank = MelSpec(), kaldiFbank()


def mel_vocos():
    B, frames = rng.randint(1, 4), rng.randint(4, 400)   # reflect padding needs more than n_fft / 2 samples, as torch.stft does
    wav = SY.synthetic_ref_wave(frames, batch=B) * rng.choice([0.1, 1.0, 3.0])
    extra = rng.randint(0, 255)                       # a wave that is not a whole number of hops
    if extra:
        wav = torch.cat([wav, 0.01 * torch.randn(B, extra)], 1)
    mel = melspec(wav.cuda())
    ref = O.log_mel_spectrogram(wav)
    assert mel.shape == ref.shape, (mel.shape, ref.shape)
    torch.testing.assert_close(mel.cpu(), ref, rtol=1e-4, atol=3e-4)
    out = voc.decode(mel)
    ref_w = O.vocos_decode(vs, ref)
    assert out.shape == ref_w.shape, (out.shape, ref_w.shape)
    assert float((out.cpu() - ref_w).abs().max()) < 1e-3 * float(ref_w.abs().max()) + 1e-6
    return (B, frames, extra)


def kaldi():
    B, n = rng.randint(1, 3), rng.randint(400, 16000 * 3)
    g = torch.Generator().manual_seed(rng.randint(0, 10 ** 6))
    wav = rng.choice([0.01, 0.05, 0.5]) * torch.randn(B, n, generator=g) + rng.choice([0.0, 0.02])
    feats, nf = fbank(wav.cuda())
    ref = torch.stack([P.kaldi_fbank(wav[i]) for i in range(B)])
    assert feats.shape == ref.shape and int(nf) == ref.shape[1], (feats.shape, ref.shape)
    # log of a mel-bin energy: the lowest bins of a pre-emphasised noise frame are sums of two or three small FFT bins whose
    # fp32 rounding is relative to the frame's TOTAL energy, so a handful of elements in a million sit a few 1e-3 further
    # out than the rest; the criterion is all elements within 5e-2 and at most 1 in 10 000 outside the tight band
    err = (feats.cpu() - ref).abs()
    tight = err > 3e-3 + 2e-4 * ref.abs()
    assert float(err.max()) < 5e-2 and int(tight.sum()) <= max(1, ref.numel() // 10000), (float(err.max()), int(tight.sum()))
    return (B, n)


def conformer():
    B = rng.randint(1, 2)
    T = rng.randint(70, 700)
    TP.test_default_size_conformer_vs_oracle(B, T)
    return (B, T)


t0, n, bad = time.time(), 0, []
while time.time() - t0 < budget:
    fn = rng.choice([mel_vocos, mel_vocos, kaldi, conformer])
    n += 1
    try:
        fn()
    except Exception as e:  # noqa: BLE001
        bad.append(fn.__name__)
        print("FAIL", fn.__name__, repr(e).splitlines()[0][:400], flush=True)
print(f"{n} front/back-end cases in {time.time() - t0:.0f} s, {len(bad)} failed")
sys.exit(1 if bad else 0)
