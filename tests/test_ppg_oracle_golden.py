"""Pins oracle/f5e_ppg_oracle.py against tests/golden/ppg_conformer.npz, produced by the reference's own
``ppg/asr_model.py`` + ``ppg/ppg_model.py`` classes (tests/golden/make_golden.py ppg)."""
import os

import numpy as np
import torch

from oracle import f5e_ppg_oracle as P

GOLD = os.path.join(os.path.dirname(__file__), "golden")
TOL = dict(rtol=1e-4, atol=5e-5)   # fp32 vs fp32; the conformer stacks ~20 reductions per layer


def fixture():
    z = np.load(os.path.join(GOLD, "ppg_conformer.npz"), allow_pickle=False)
    sd = {k[2:]: torch.from_numpy(z[k]) for k in z.files if k.startswith("w/")}
    g = {k: torch.from_numpy(z[k]) for k in z.files if not k.startswith("w/")}
    return sd, g


def test_conformer_extract_matches_reference():
    sd, g = fixture()
    assert P.encoder_depth(sd) == 2 and "encoder.global_cmvn.mean" in sd
    ppg, logits = P.asr_extract(sd, g["feats"], g["lens"], heads=4)
    assert ppg.shape == g["ppg"].shape == (2, 50, 64)
    valid = torch.arange(50)[None, :] < torch.tensor([50, 38])[:, None]     # frames past a sequence's end are don't-care
    torch.testing.assert_close(ppg[valid], g["ppg"][valid], **TOL)
    torch.testing.assert_close(logits.view(2, 50, -1)[valid], g["logits"].view(2, 50, -1)[valid], **TOL)


def test_mel_to_ppg_glue_matches_reference():
    sd, g = fixture()
    tgt, true_len = P.mel_to_ppg(sd, g["feats"], g["lens"], heads=4)
    assert torch.equal(true_len, g["true_len"]) and true_len.tolist() == [50, 38]      # integer work: bit exact
    assert tgt.shape == g["target"].shape
    torch.testing.assert_close(tgt, g["target"], **TOL)
    assert float(tgt[1, 38:].abs().max()) == 0.0


def test_kaldi_fbank_shape_and_tables():
    """torchaudio.compliance.kaldi.fbank is not in the tree (PARITY UNPINNED); structural checks only: frame count of
    snip_edges, triangular mel bank that covers 20 Hz .. Nyquist with unit peaks, povey window end points."""
    wav = 0.1 * torch.randn(16000, generator=torch.Generator().manual_seed(1))
    fb = P.kaldi_fbank(wav)
    assert fb.shape == (1 + (16000 - 400) // 160, 80) and torch.isfinite(fb).all()
    banks = P.kaldi_mel_banks()
    assert banks.shape == (80, 257) and float(banks[:, -1].abs().max()) == 0.0 and float(banks.min()) >= 0.0
    assert float(banks.max()) <= 1.0 and (banks.sum(1) > 0).all()
    w = P.povey_window(400)
    assert float(w[0]) == 0.0 and abs(float(w[199]) - 1.0) < 1e-3
    # a pure DC offset is removed before the FFT: the features of x and x + c coincide
    torch.testing.assert_close(P.kaldi_fbank(wav + 0.05), fb, rtol=1e-3, atol=1e-3)
