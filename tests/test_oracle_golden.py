"""Pins oracle/f5e_oracle.py against fixtures produced by the reference itself (tests/golden/make_golden.py)."""
import ast
import os

import numpy as np
import pytest
import torch

from oracle import f5e_oracle as O
from tools import synth as SY

GOLD = os.path.join(os.path.dirname(__file__), "golden")
CASES = ["b1", "b2_mask", "b1_midpoint", "b2_ppg_tts", "b1_ppg_vc"]
TOL = dict(rtol=1e-4, atol=2e-5)  # fp32 vs fp32, same op order up to reassociation inside ATen


def load_case(tag):
    z = np.load(os.path.join(GOLD, f"dit_{tag}.npz"), allow_pickle=False)
    meta = ast.literal_eval(str(z["meta"]))
    sd = {k[2:]: torch.from_numpy(z[k]) for k in z.files if k.startswith("w/")}
    g = {k: torch.from_numpy(z[k]) for k in z.files if not k.startswith("w/") and k != "meta"}
    cfg = O.DiTConfig(dim=meta["dim"], depth=meta["depth"], heads=meta["heads"], dim_head=meta["dim_head"],
                      ff_mult=meta["ff_mult"], mel_dim=meta["mel_dim"], text_num_embeds=meta["text_num_embeds"],
                      text_dim=meta["text_dim"], text_mask_padding=meta.get("text_mask_padding", True),
                      qk_norm=meta.get("qk_norm"), conv_layers=meta["conv_layers"],
                      pe_attn_head=meta.get("pe_attn_head"),
                      long_skip_connection=meta.get("long_skip_connection", False),
                      use_ppg=meta["n_ppg"] > 0, ppg_dim=32)
    return meta, sd, g, cfg


@pytest.mark.parametrize("tag", CASES)
def test_dit_forward_matches_reference(tag):
    meta, sd, g, cfg = load_case(tag)
    mask = g.get("fwd/mask")
    ppg = g.get("fwd/ppg")
    for drop, key in ((False, "fwd/pred_cond"), (True, "fwd/pred_uncond")):
        pred = O.dit_sample(sd, cfg, g["fwd/x"], g["fwd/cond"], g["fwd/text"], ppg, g["fwd/time"],
                            drop, drop, drop, mask)
        torch.testing.assert_close(pred, g[key], **TOL)
        assert float(g[key].abs().max()) > 1e-3  # not the vacuous all-zero network


@pytest.mark.parametrize("tag", CASES)
def test_dit_intermediates_match_reference(tag):
    meta, sd, g, cfg = load_case(tag)
    b, n = g["fwd/x"].shape[:2]
    t = O.time_embedding(sd, g["fwd/time"].repeat(b))
    torch.testing.assert_close(t, g["fwd/c_time_embed_out"], **TOL)
    for drop, pre in ((False, "c_"), (True, "u_")):
        te = O.text_embedding(sd, g["fwd/text"], b, n, drop, mask_padding=cfg.text_mask_padding)
        torch.testing.assert_close(te, g[f"fwd/{pre}text_embed_out"], **TOL)
        pe = None
        if cfg.use_ppg:
            pe = O.ppg_embedding(sd, g["fwd/ppg"], b, n, drop)
            torch.testing.assert_close(pe, g[f"fwd/{pre}ppg_embed_out"], **TOL)
        h = O.input_embedding(sd, g["fwd/x"], g["fwd/cond"], te, pe, drop)
        torch.testing.assert_close(h, g[f"fwd/{pre}input_embed_out"], **TOL)
        freqs = O.rope_freqs(n, cfg.dim_head, sd["rotary_embed.inv_freq"])
        blk = O.dit_block(sd, "transformer_blocks.0.", h, t, cfg.heads, g.get("fwd/mask"), freqs, cfg.pe_attn_head)
        torch.testing.assert_close(blk, g[f"fwd/{pre}block0_out"], **TOL)


@pytest.mark.parametrize("tag", CASES)
def test_sampler_matches_reference(tag):
    meta, sd, g, cfg = load_case(tag)
    kw = dict(duration=g["smp/duration"], lens=g["smp/lens"], steps=int(g["smp/steps"]), sway_sampling_coef=-1.0,
              seed=int(g["smp/seed"]), method=meta["method"])
    if meta["mode"] == "cfg":
        out, traj = O.cfm_sample(sd, cfg, g["smp/cond"], g["fwd/text"], g.get("fwd/ppg"),
                                 cfg_strength=float(g["smp/cfg"]), **kw)
    elif meta["mode"] == "tts":
        out, traj = O.cfm_sample(sd, cfg, g["smp/cond"], g["fwd/text"], None, mode="tts", alpha_a=2.5, alpha_b=3.0, **kw)
    else:
        out, traj = O.cfm_sample(sd, cfg, g["smp/cond"], None, g["fwd/ppg"], mode="vc", alpha_a=2.5, alpha_b=3.0, **kw)
    torch.testing.assert_close(traj[0], g["smp/traj"][0], rtol=0, atol=0)  # seeded noise: bit exact
    torch.testing.assert_close(traj, g["smp/traj"], rtol=2e-4, atol=1e-4)
    torch.testing.assert_close(out, g["smp/out"], rtol=2e-4, atol=1e-4)


def test_sample_prep_matches_reference():
    z = np.load(os.path.join(GOLD, "cfm_prep.npz"))
    g = {k: torch.from_numpy(z[k]) for k in z.files}
    mel = O.log_mel_spectrogram(g["wav"])
    torch.testing.assert_close(mel, g["mel"], rtol=1e-5, atol=1e-5)
    prep = O.sample_prep(mel.permute(0, 2, 1), g["text"], torch.tensor([20, 30]), torch.tensor([12, 9]), seed=11,
                         max_duration=28)
    torch.testing.assert_close(prep["step_cond"], g["step_cond"], rtol=1e-5, atol=1e-5)
    assert torch.equal(prep["mask"], g["mask"])
    assert torch.equal(prep["y0"], g["y0"])
    t = O.sway_time_grid(4, -1.0)
    torch.testing.assert_close(t[:-1], g["t"], rtol=0, atol=0)
    # closed form of the sway grid
    i = torch.arange(5, dtype=torch.float64)
    torch.testing.assert_close(t.double(), 1 - torch.cos(torch.pi * i / 8), rtol=0, atol=1e-6)


def test_mel_and_istft_third_party_crosschecks():
    """torchaudio / vocos arithmetic is 'parity unpinned'; cross-check structure with torch.stft/istft only."""
    w = SY.synthetic_ref_wave(20)
    assert w.shape[-1] // 256 + 1 == 20
    mel = O.log_mel_spectrogram(w)
    assert mel.shape == (1, 100, 20)
    fb = O.mel_filterbank_htk()
    assert fb.shape == (513, 100) and float(fb.min()) >= 0 and float(fb.sum(0).min()) > 0
    vs = SY.init_vocos_state()
    wav = O.vocos_decode(vs, mel)
    assert wav.shape == (1, 256 * 19)
    assert torch.isfinite(wav).all()


def _vq_case(tag):
    z = np.load(os.path.join(GOLD, "vq_eval.npz"))
    g = {k[len(tag) + 1:]: torch.from_numpy(z[k]) for k in z.files if k.startswith(tag + "/")}
    w = {k[2:]: v for k, v in g.items() if k.startswith("w/")}
    if "weight_proj.weight" in w:
        layers = [(w["weight_proj.weight"], w["weight_proj.bias"], False)]
    else:
        n = max(int(k.split(".")[1]) for k in w if k.startswith("weight_proj."))
        layers = [(w[f"weight_proj.{i}.0.weight"], w[f"weight_proj.{i}.0.bias"], True) for i in range(n)]
        layers.append((w[f"weight_proj.{n}.weight"], w[f"weight_proj.{n}.bias"], False))
    groups = {"plain": 2, "combine": 2, "deep": 4}[tag]
    return g, w, layers, groups


@pytest.mark.parametrize("tag", ["plain", "combine", "deep"])
def test_gumbel_vq_eval_matches_reference(tag):
    g, w, layers, groups = _vq_case(tag)
    r = O.gumbel_vq_eval(g["x"], layers, w["vars"], groups, 10, combine_groups=(tag == "combine"))
    assert torch.equal(r["targets"], g["targets"])
    torch.testing.assert_close(r["x"], g["q"], rtol=0, atol=0)
    torch.testing.assert_close(r["code_perplexity"], g["code_perplexity"], rtol=1e-6, atol=1e-6)
    torch.testing.assert_close(r["prob_perplexity"], g["prob_perplexity"], rtol=1e-6, atol=1e-6)


@pytest.mark.parametrize("tag,skip", [("concat_b1", "concat"), ("add_b2", "add")])
def test_unett_forward_matches_reference(tag, skip):
    z = np.load(os.path.join(GOLD, f"unett_{tag}.npz"))
    g = {k: torch.from_numpy(z[k]) for k in z.files}
    sd = {k[2:]: v for k, v in g.items() if k.startswith("w/")}
    for drop in (False, True):
        out = O.unett_forward(sd, 2, g["x"], g["cond"], g["text"], g["time"], drop, drop, g.get("mask"), skip)
        torch.testing.assert_close(out, g[f"pred_drop{int(drop)}"], **TOL)


# ---- transform part of the mel front-end (a14) and of the Vocos iSTFT head (a17), pinned by reference-held code:
# runtime/triton_trtllm/scripts/conv_stft.py (STFT.transform / .inverse) and export_vocoder_to_onnx.py (ISTFTHead)

def _stft_fixture():
    z = np.load(os.path.join(GOLD, "stft_head.npz"), allow_pickle=False)
    return {k: torch.from_numpy(z[k]) for k in z.files}


@pytest.mark.parametrize("tag", ["a", "b"])
def test_oracle_stft_matches_reference_conv_stft(tag):
    """The oracle's front-end STFT (torch.stft: periodic hann 1024, hop 256, centred, reflect padding) against the
    reference's convolutional STFT: same frame count (1 + nw // 256), real / imaginary parts and magnitude.  The conv
    form multiplies by an explicit 1024 x 1026 DFT matrix in fp32, so agreement is to fp32 summation error of a 1024-term
    dot product (1e-4 of the largest magnitude), not bit level."""
    g = _stft_fixture()
    wav = g[f"fwd_{tag}/wav"]
    window = torch.hann_window(1024, periodic=True)
    spec = torch.stft(wav, 1024, hop_length=256, win_length=1024, window=window, center=True, pad_mode="reflect",
                      normalized=False, onesided=True, return_complex=True)
    assert spec.shape == g[f"fwd_{tag}/real"].shape == (2, 513, 1 + wav.shape[1] // 256)
    scale = float(g[f"fwd_{tag}/mag"].max())
    for got, key in ((spec.real, "real"), (spec.imag, "imag"), (spec.abs(), "mag")):
        assert float((got - g[f"fwd_{tag}/{key}"]).abs().max()) < 1e-4 * scale, key
    # and the oracle's log-mel is exactly log(clamp(fb^T |S|, 1e-5)) of that magnitude (HTK filterbank values: the one
    # part of a14 that stays unpinned -- they come from torchaudio, absent here and from the reference tree)
    mel = O.log_mel_spectrogram(wav)
    fb = O.mel_filterbank_htk(513, 100, 24000)
    want = torch.matmul(g[f"fwd_{tag}/mag"].transpose(-1, -2), fb).transpose(-1, -2).clamp(min=1e-5).log()
    torch.testing.assert_close(mel, want, rtol=1e-3, atol=1e-3)


@pytest.mark.parametrize("tag", ["a", "b"])
def test_oracle_istft_head_matches_reference_istft_head(tag):
    """O.istft_head (exp, clip at 100, cos / sin, torch.istft centred) against the reference's ISTFTHead.  The reference's
    conv form keeps hop * T samples (conv_stft.py:225-231), torch.istft -- which is what vocoder.decode runs at
    infer/utils_infer.py:489 -- keeps hop * (T - 1): the common hop * (T - 1) samples are compared.  fp32 both sides; the
    reference inverts through a pseudo-inverse of the DFT matrix, so 2e-4 of the peak."""
    g = _stft_fixture()
    z = g[f"inv_{tag}/z"]
    vs = {"head.out.weight": torch.eye(1026), "head.out.bias": torch.zeros(1026)}
    audio = O.istft_head(vs, z)
    ref = g[f"inv_{tag}/audio"]
    T = z.shape[1]
    assert audio.shape == (z.shape[0], 256 * (T - 1)) and ref.shape == (z.shape[0], 256 * T)
    common = ref[:, : 256 * (T - 1)]
    assert float((audio[0] - common[0]).abs().max()) < 2e-4 * float(common[0].abs().max())
    # Batch items > 0: the reference's conv inverse divides by the window envelope through `th.where(coff > 1e-8)` on a
    # [1, 1, L] tensor (conv_stft.py:230-233), which addresses batch item 0 only -- later items come out UN-normalised.
    # That is a quirk of the ONNX-export restatement, not of vocoder.decode (torch.istft normalises every item), so the
    # oracle is held to it as "oracle x envelope == reference" there: the transform itself is pinned for those items too.
    if z.shape[0] > 1:
        w2 = torch.hann_window(1024, periodic=True) ** 2
        env = torch.zeros(256 * T + 1024)
        for f in range(T):
            env[f * 256: f * 256 + 1024] += w2
        env = env[512: 512 + 256 * (T - 1)]
        for b in range(1, z.shape[0]):
            assert float((audio[b] * env - common[b]).abs().max()) < 2e-4 * float(common[b].abs().max())


def test_ppg_embedding_transformer_variant_matches_reference():
    """PPGEmbedding(use_transformer=True) (backbones/dit.py:105-119) against the reference class's own outputs."""
    z = np.load(os.path.join(GOLD, "ppg_embed_transformer.npz"), allow_pickle=False)
    sd = {"ppg_embed." + k[2:]: torch.from_numpy(z[k]) for k in z.files if k.startswith("w/")}
    ppg = torch.from_numpy(z["ppg"])
    for drop, key in ((False, "out"), (True, "out_drop")):
        got = O.ppg_embedding(sd, ppg, 2, 14, drop, heads=4)
        torch.testing.assert_close(got, torch.from_numpy(z[key]), **TOL)
    torch.testing.assert_close(O.ppg_embedding(sd, None, 2, 14, False, heads=4), torch.from_numpy(z["out_none"]), **TOL)
