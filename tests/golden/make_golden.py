#!/usr/bin/env python3
"""Generate tests/golden/*.npz by running the REFERENCE's own Python for the hot path.

Usage (build container only; the reference never travels to the GPU box):

    python tests/golden/make_golden.py /root/reference

What it does
  * loads the reference's ``model/modules.py``, ``model/backbones/dit.py``, ``model/cfm.py``,
    ``model/utils.py`` and ``durpred/*`` from ``<ref>/src`` *where they lie* (nothing is copied),
    bypassing ``f5_tts/model/__init__.py`` (which drags in the trainer) by registering stub
    parent packages with ``__path__`` set;
  * injects ``sys.modules`` shims for third-party packages absent from this image
    (x_transformers, torchdiffeq, torchaudio, librosa, numba, jieba, pypinyin).  The shims are
    this repo's own restatements of those packages' *published* algorithms;
    they are NOT reference code, and the arithmetic inside them stays "parity unpinned";
  * runs seeded small configurations through the reference modules and stores inputs, weights
    and outputs as ``.npz`` fixtures (data only) next to this script.

The fixtures pin ``oracle/f5e_oracle.py`` (tests/test_oracle_golden.py).
"""
from __future__ import annotations

import importlib
import importlib.util
import math
import os
import sys
import types

import numpy as np
import torch
import torch.nn.functional as F
from torch import nn

HERE = os.path.dirname(os.path.abspath(__file__))


# ---------------------------------------------------------------- shims ----

def _install_shims():
    # x_transformers -------------------------------------------------------
    class RotaryEmbedding(nn.Module):
        def __init__(self, dim, theta=10000.0):
            super().__init__()
            inv_freq = 1.0 / (theta ** (torch.arange(0, dim, 2).float() / dim))
            self.register_buffer("inv_freq", inv_freq)

        def forward_from_seq_len(self, seq_len):
            t = torch.arange(seq_len, device=self.inv_freq.device)
            return self.forward(t)

        def forward(self, t):
            if t.ndim == 1:
                t = t[None]
            freqs = torch.einsum("b i , j -> b i j", t.type_as(self.inv_freq), self.inv_freq)
            freqs = torch.stack((freqs, freqs), dim=-1).flatten(-2)
            return freqs, 1.0

    def rotate_half(x):
        x = x.reshape(*x.shape[:-1], x.shape[-1] // 2, 2)
        x1, x2 = x.unbind(dim=-1)
        return torch.stack((-x2, x1), dim=-1).flatten(-2)

    def apply_rotary_pos_emb(t, freqs, scale=1):
        rot_dim, seq_len, orig_dtype = freqs.shape[-1], t.shape[-2], t.dtype
        freqs = freqs[:, -seq_len:, :]
        if t.ndim == 4 and freqs.ndim == 3:
            freqs = freqs[:, None]
        t, t_unrot = t[..., :rot_dim], t[..., rot_dim:]
        t = (t * freqs.cos() * scale) + (rotate_half(t) * freqs.sin() * scale)
        return torch.cat((t, t_unrot), dim=-1).type(orig_dtype)

    class XRMSNorm(nn.Module):
        def __init__(self, dim):
            super().__init__()
            self.scale = dim ** 0.5
            self.g = nn.Parameter(torch.ones(dim))

        def forward(self, x):
            return F.normalize(x, dim=-1) * self.scale * self.g

    xt = types.ModuleType("x_transformers")
    xtx = types.ModuleType("x_transformers.x_transformers")
    xtx.RotaryEmbedding = RotaryEmbedding
    xtx.apply_rotary_pos_emb = apply_rotary_pos_emb
    xt.RMSNorm = XRMSNorm
    xt.x_transformers = xtx
    sys.modules["x_transformers"] = xt
    sys.modules["x_transformers.x_transformers"] = xtx

    # torchdiffeq (fixed-grid euler / midpoint on the given grid) ----------
    def odeint(fn, y0, t, method="euler", **kw):
        ys, y = [y0], y0
        for i in range(len(t) - 1):
            t0, t1 = t[i], t[i + 1]
            dt = t1 - t0
            if method == "euler":
                y = y + dt * fn(t0, y)
            elif method == "midpoint":
                half = 0.5 * dt
                y = y + dt * fn(t0 + half, y + fn(t0, y) * half)
            else:
                raise NotImplementedError(method)
            ys.append(y)
        return torch.stack(ys)

    td = types.ModuleType("torchdiffeq")
    td.odeint = odeint
    sys.modules["torchdiffeq"] = td

    # torchaudio.transforms.MelSpectrogram (HTK, norm=None) ----------------
    class MelSpectrogram(nn.Module):
        def __init__(self, sample_rate, n_fft, win_length, hop_length, n_mels, power, center, normalized, norm):
            super().__init__()
            self.n_fft, self.win, self.hop, self.power, self.center = n_fft, win_length, hop_length, power, center
            nf = n_fft // 2 + 1
            all_freqs = torch.linspace(0, sample_rate // 2, nf)
            m_max = 2595.0 * math.log10(1.0 + (sample_rate // 2) / 700.0)
            m_pts = torch.linspace(0.0, m_max, n_mels + 2)
            f_pts = 700.0 * (10.0 ** (m_pts / 2595.0) - 1.0)
            f_diff = f_pts[1:] - f_pts[:-1]
            slopes = f_pts.unsqueeze(0) - all_freqs.unsqueeze(1)
            fb = torch.clamp(torch.min(-slopes[:, :-2] / f_diff[:-1], slopes[:, 2:] / f_diff[1:]), min=0.0)
            self.register_buffer("fb", fb)
            self.register_buffer("window", torch.hann_window(win_length))

        def forward(self, wav):
            s = torch.stft(wav, self.n_fft, self.hop, self.win, self.window, center=self.center,
                           pad_mode="reflect", normalized=False, onesided=True, return_complex=True).abs()
            if self.power != 1:
                s = s.pow(self.power)
            return torch.matmul(s.transpose(-1, -2), self.fb).transpose(-1, -2)

    ta = types.ModuleType("torchaudio")
    tat = types.ModuleType("torchaudio.transforms")
    tat.MelSpectrogram = MelSpectrogram
    ta.transforms = tat
    sys.modules["torchaudio"] = ta
    sys.modules["torchaudio.transforms"] = tat

    # librosa.filters.mel (bigvgan path, never called here) ----------------
    lb = types.ModuleType("librosa")
    lbf = types.ModuleType("librosa.filters")
    lbf.mel = lambda **kw: (_ for _ in ()).throw(NotImplementedError("librosa shim"))
    lb.filters = lbf
    sys.modules["librosa"] = lb
    sys.modules["librosa.filters"] = lbf

    # numba (training-only MAS jit): pass-through decorator ----------------
    class _T:
        def __getitem__(self, k):
            return self

        def __call__(self, *a, **k):
            return self

    nb = types.ModuleType("numba")
    nb.jit = lambda *a, **k: (lambda f: f)
    nb.void = nb.int32 = nb.float32 = _T()
    sys.modules["numba"] = nb

    # tokeniser deps (not exercised) ---------------------------------------
    jb = types.ModuleType("jieba")
    import re as _re
    # jieba on single-byte text yields alphanumeric runs and single other characters (restated; unpinned third party)
    jb.cut = lambda s: _re.findall(r"[A-Za-z0-9]+|.", s, _re.S)
    jb.dt = types.SimpleNamespace(initialized=True)
    sys.modules["jieba"] = jb
    pp = types.ModuleType("pypinyin")
    pp.Style = types.SimpleNamespace(TONE3=0)
    pp.lazy_pinyin = lambda *a, **k: []
    sys.modules["pypinyin"] = pp


def load_reference(ref_root: str):
    """Returns (modules_mod, dit_mod, cfm_mod, utils_mod) loaded from <ref_root>/src/f5_tts."""
    src = os.path.join(ref_root, "src", "f5_tts")
    if not os.path.isdir(src):
        raise FileNotFoundError(src)
    _install_shims()
    for name, sub in (("f5_tts", ""), ("f5_tts.model", "model"), ("f5_tts.model.backbones", "model/backbones")):
        pkg = types.ModuleType(name)
        pkg.__path__ = [os.path.join(src, sub)]
        sys.modules[name] = pkg
    # durpred's __init__ is importable as-is (einops present); let the normal machinery find it
    mods = {}
    for name in ("f5_tts.model.utils", "f5_tts.model.modules", "f5_tts.model.backbones.dit", "f5_tts.model.cfm"):
        mods[name] = importlib.import_module(name)
    return (mods["f5_tts.model.modules"], mods["f5_tts.model.backbones.dit"], mods["f5_tts.model.cfm"],
            mods["f5_tts.model.utils"])


# ------------------------------------------------------------- fixtures ----

def _unzero(model: nn.Module, seed: int):
    """a default-init DiT outputs exactly 0; re-randomise the zeroed tensors N(0, 0.02)."""
    g = torch.Generator().manual_seed(seed)
    for name, p in model.named_parameters():
        if float(p.detach().abs().max()) == 0.0:
            p.data.copy_(torch.randn(p.shape, generator=g) * 0.02)
        if name.endswith("grn.gamma") or name.endswith("grn.beta"):
            p.data.copy_(torch.randn(p.shape, generator=g) * 0.1)
    for name, b in model.named_buffers():
        if name.endswith("running_mean"):
            b.copy_(torch.randn(b.shape, generator=g) * 0.1)
        if name.endswith("running_var"):
            b.copy_(1.0 + 0.2 * torch.rand(b.shape, generator=g))


def _np(d):
    out = {}
    for k, v in d.items():
        if v is None:
            continue
        if isinstance(v, torch.Tensor):
            v = v.detach().cpu().numpy()
        out[k] = np.asarray(v)
    return out


def _sd(model, prefix="w/"):
    return {prefix + k: v for k, v in model.state_dict().items()}


def make_dit_case(dit_mod, cfm_mod, tag, arch, b, n, nc, nt, steps, cfg_strength, seed, n_ppg=0,
                  method="euler", mode="cfg"):
    torch.manual_seed(seed)
    ppg_config = dict(use_ppg=False)
    if n_ppg:
        ppg_config = dict(use_ppg=True, ppg_dim=32, use_transformer=False, transformer_config=dict(),
                          combined_cond_drop_prob=[0.3, 0.1, 0.5, 0.1])
    model = dit_mod.DiT(**arch, ppg_config=ppg_config)
    _unzero(model, seed + 1)
    model.eval()
    g = torch.Generator().manual_seed(seed + 2)
    x = torch.randn(b, n, arch["mel_dim"], generator=g)
    cond = torch.randn(b, n, arch["mel_dim"], generator=g)
    text = torch.randint(0, arch["text_num_embeds"], (b, nt), generator=g)
    if b > 1:
        text[1, nt - 3:] = -1
    ppg = torch.randn(b, n_ppg, 32, generator=g) if n_ppg else None
    tm = torch.tensor(0.37)
    mask = None
    if b > 1:
        lens = torch.tensor([n] + [n - 7] * (b - 1))
        mask = torch.arange(n)[None] < lens[:, None]
    out = {}
    # per-op intermediates through forward hooks on the first block
    inter = {}

    def hook(name):
        def f(mod, inp, outp):
            inter[name] = outp[0] if isinstance(outp, tuple) else outp
        return f

    hs = [model.transformer_blocks[0].register_forward_hook(hook("block0_out")),
          model.input_embed.register_forward_hook(hook("input_embed_out")),
          model.text_embed.register_forward_hook(hook("text_embed_out")),
          model.time_embed.register_forward_hook(hook("time_embed_out")),
          model.transformer_blocks[0].attn.register_forward_hook(hook("block0_attn_out"))]
    if n_ppg:
        hs.append(model.ppg_embed.register_forward_hook(hook("ppg_embed_out")))
    with torch.no_grad():
        pred_c = model.sample(x=x, cond=cond, text=text, ppg=ppg, time=tm, mask=mask,
                              drop_audio_cond=False, drop_text=False, drop_ppg=False)
        inter_c = dict(inter)
        model.clear_cache()
        pred_u = model.sample(x=x, cond=cond, text=text, ppg=ppg, time=tm, mask=mask,
                              drop_audio_cond=True, drop_text=True, drop_ppg=True)
        inter_u = dict(inter)
        model.clear_cache()
    for h in hs:
        h.remove()
    out.update({"fwd/x": x, "fwd/cond": cond, "fwd/text": text, "fwd/ppg": ppg, "fwd/time": tm, "fwd/mask": mask,
                "fwd/pred_cond": pred_c, "fwd/pred_uncond": pred_u})
    out.update({"fwd/c_" + k: v for k, v in inter_c.items()})
    out.update({"fwd/u_" + k: v for k, v in inter_u.items()})

    # full sampler through the reference CFM
    cfm = cfm_mod.CFM(transformer=model, odeint_kwargs=dict(method=method), ppg_config=ppg_config,
                      mel_spec_kwargs=dict(n_mel_channels=arch["mel_dim"]))
    cfm.eval()
    cond_mel = torch.randn(b, nc, arch["mel_dim"], generator=g)
    lens_s = torch.tensor([nc] + [nc - 5] * (b - 1))
    dur = torch.tensor([n] + [n - 7] * (b - 1))
    with torch.no_grad():
        if mode == "cfg":
            o, traj = cfm.sample(cond=cond_mel, text=text, ppg=ppg, duration=dur, lens=lens_s, steps=steps,
                                 cfg_strength=cfg_strength, sway_sampling_coef=-1.0, seed=seed + 3)
        elif mode == "tts":
            o, traj = cfm.sample_tts(cond=cond_mel, text=text, duration=dur, lens=lens_s, steps=steps,
                                     alpha_spk=2.5, alpha_txt=3.0, sway_sampling_coef=-1.0, seed=seed + 3)
        else:
            o, traj = cfm.sample_vc(cond=cond_mel, ppg=ppg, duration=dur, lens=lens_s, steps=steps,
                                    alpha_spk=2.5, alpha_ppg=3.0, sway_sampling_coef=-1.0, seed=seed + 3)
    out.update({"smp/cond": cond_mel, "smp/lens": lens_s, "smp/duration": dur, "smp/out": o, "smp/traj": traj,
                "smp/seed": seed + 3, "smp/steps": steps, "smp/cfg": cfg_strength})
    out.update(_sd(model))
    meta = dict(arch)
    meta.update(b=b, n=n, nc=nc, nt=nt, n_ppg=n_ppg, mode=mode, method=method)
    out["meta"] = np.array(repr(meta))
    path = os.path.join(HERE, f"dit_{tag}.npz")
    np.savez_compressed(path, **_np(out))
    print("wrote", path, os.path.getsize(path) // 1024, "KiB")


def make_prep_case(cfm_mod, dit_mod):
    """CFM.sample prep (masks/padding/duration clamp/y0/t grid) captured through a recording transformer."""
    class Rec(nn.Module):
        def __init__(self):
            super().__init__()
            self.dim = 8
            self.p = nn.Parameter(torch.zeros(1))
            self.calls = []

        def sample(self, **kw):
            self.calls.append({k: (v.clone() if isinstance(v, torch.Tensor) else v) for k, v in kw.items()})
            return torch.zeros_like(kw["x"])

        def clear_cache(self):
            pass

    rec = Rec()
    cfm = cfm_mod.CFM(transformer=rec, mel_spec_kwargs=dict(n_mel_channels=100))
    g = torch.Generator().manual_seed(99)
    wav = 0.1 * torch.randn(2, 256 * 11 + 17, generator=g)
    text = torch.tensor([[5, 9, 2, -1, -1], [3, 3, 7, 7, 1]])
    with torch.no_grad():
        o, traj = cfm.sample(cond=wav, text=text, duration=torch.tensor([20, 30]), lens=torch.tensor([12, 9]),
                             steps=4, cfg_strength=2.0, sway_sampling_coef=-1.0, seed=11, max_duration=28)
    c0 = rec.calls[0]
    out = {"wav": wav, "text": text, "step_cond": c0["cond"], "mask": c0["mask"], "y0": traj[0],
           "t": torch.stack([c["time"] for c in rec.calls[::2]]), "out": o,
           "mel": cfm.mel_spec(wav)}
    path = os.path.join(HERE, "cfm_prep.npz")
    np.savez_compressed(path, **_np(out))
    print("wrote", path)


def load_reference_infer(ref_root, cfm_mod):
    """reference infer/utils_infer.py with its heavyweight imports shimmed (nothing in them is exercised)."""
    def mod(name, **attrs):
        m = types.ModuleType(name)
        for k, v in attrs.items():
            setattr(m, k, v)
        sys.modules[name] = m
        return m

    mpl = mod("matplotlib", use=lambda *a, **k: None)
    mpl.pylab = mod("matplotlib.pylab")
    mod("pydub", AudioSegment=object, silence=types.SimpleNamespace())
    mod("vocos", Vocos=object)
    mod("transformers", pipeline=lambda *a, **k: None)
    mod("huggingface_hub", hf_hub_download=lambda *a, **k: None)
    sys.modules["torchaudio"].load = lambda *a, **k: (_ for _ in ()).throw(NotImplementedError("shim"))
    sys.modules["f5_tts.model"].CFM = cfm_mod.CFM
    pkg = types.ModuleType("f5_tts.infer")
    pkg.__path__ = [os.path.join(ref_root, "src", "f5_tts", "infer")]
    sys.modules["f5_tts.infer"] = pkg
    return importlib.import_module("f5_tts.infer.utils_infer")


def make_callers_case(ref_root, cfm_mod):
    """Callers of the path: chunk_text, process_batch's duration / slice / RMS / cross-fade
    rules and load_checkpoint's key handling, captured from the reference with recording stubs."""
    import json
    U = load_reference_infer(ref_root, cfm_mod)
    out = {"chunk_text": [], "batch": []}
    for text, mc in (("Hello there. This is a test, of chunking; really! Yes? Ok: fine.", 24),
                     ("One sentence only", 135), ("A. B. C. D. E. F. G.", 5),
                     ("No punctuation here at all just words going on and on", 20)):
        out["chunk_text"].append({"text": text, "max_chars": mc, "chunks": U.chunk_text(text, max_chars=mc)})

    class Model:
        def __init__(self):
            self.calls = []

        def sample(self, **kw):
            self.calls.append({k: (v if not isinstance(v, torch.Tensor) else list(v.shape)) for k, v in kw.items()})
            return torch.zeros(1, kw["duration"], 100), None

    class Voc:
        def decode(self, mel):
            n = 256 * (mel.shape[-1] - 1)
            return (torch.arange(n, dtype=torch.float32)[None] % 97) / 97.0 - 0.5

    g = torch.Generator().manual_seed(21)
    for nw, amp, ref_text, gens, speed, fixd, xf in (
            (24000 * 2 + 100, 0.02, "Reference text here.", ["Short.", "A somewhat longer piece of generated text."], 1.0, None, 0.15),
            (24000 * 3, 0.5, "Another reference", ["Tiny", "Medium sized chunk of words", "And a third chunk to fade."], 1.3, None, 0.05),
            (24000 * 2, 0.05, "Fixed duration case.", ["Whatever text."], 1.0, 6.5, 0.0)):
        audio = amp * torch.randn(2, nw, generator=g)
        m = Model()
        wave_, sr, spec = next(U.infer_batch_process((audio, 24000), ref_text, gens, m, Voc(), progress=None,
                                                     cross_fade_duration=xf, speed=speed, fix_duration=fixd,
                                                     device="cpu"))
        out["batch"].append({"nw": nw, "amp": amp, "seed_note": "audio = amp * randn(2, nw, generator seeded 21, in order)",
                             "ref_text": ref_text, "gens": gens, "speed": speed, "fix_duration": fixd, "cross_fade": xf,
                             "durations": [c["duration"] for c in m.calls],
                             "cond_shapes": [c["cond"] for c in m.calls],
                             "wave_len": int(len(wave_)), "wave_sum": float(np.sum(wave_)),
                             "wave_abs_sum": float(np.abs(wave_).sum()), "wave_head": [float(x) for x in wave_[:8]],
                             "spec_shape": list(spec.shape)})
    # load_checkpoint key handling
    import tempfile
    lin = nn.Linear(3, 2)
    sd = {"ema_model." + k: v.clone() + 1 for k, v in lin.state_dict().items()}
    sd.update({"initted": torch.tensor(True), "step": torch.tensor(5),
               "ema_model.mel_spec.mel_stft.mel_scale.fb": torch.zeros(2),
               "ema_model.mel_spec.mel_stft.spectrogram.window": torch.zeros(2)})
    with tempfile.TemporaryDirectory() as d:
        pth = os.path.join(d, "m.pt")
        torch.save({"ema_model_state_dict": sd}, pth)
        loaded = U.load_checkpoint(nn.Linear(3, 2), pth, "cpu", dtype=torch.float32, use_ema=True)
    out["load_checkpoint"] = {"in_keys": sorted(sd.keys()), "loaded_keys": sorted(loaded.state_dict().keys()),
                              "weight_delta": float((loaded.weight - lin.weight).mean())}
    path = os.path.join(HERE, "callers.json")
    with open(path, "w") as f:
        json.dump(out, f, indent=1)
    print("wrote", path)


def make_eval_case(ref_root, cfm_mod):
    """Integer / index callers of the eval driver, captured from the reference's own eval/utils_eval.py:77-219
    (get_inference_prompt: trailing-space rule, total_mel_len, bucket index, flush order, residual buckets, the seed-666
    shuffle) and model/utils.py:270-311 (convert_char_to_pinyin) on ASCII input.  Audio loading is stubbed (the prompt
    "file name" encodes length and seed), the mel is the reference's own MelSpec over the torchaudio shim; what is captured
    is data: utterance order per batch, frame counts, token lists.  jieba's segmentation of single-byte text is the shim's
    restatement (third party, unpinned); everything the REFERENCE does around it is pinned here."""
    import json
    load_reference_infer(ref_root, cfm_mod)          # installs the remaining import shims (pydub, vocos, ...)
    tr = sys.modules["transformers"]
    tr.WhisperProcessor = tr.WhisperForConditionalGeneration = object
    ev = types.ModuleType("f5_tts.eval")
    ev.__path__ = [os.path.join(ref_root, "src", "f5_tts", "eval")]
    sys.modules["f5_tts.eval"] = ev
    ec = types.ModuleType("f5_tts.eval.ecapa_tdnn")
    ec.ECAPA_TDNN_SMALL = object
    sys.modules["f5_tts.eval.ecapa_tdnn"] = ec
    if "tqdm" not in sys.modules:
        import tqdm  # noqa: F401
    ta = sys.modules["torchaudio"]

    def fake_load(path):      # "<anything>/n<samples>_s<seed>_a<amp in 1/1000>.wav"
        name = os.path.basename(path)[:-4]
        n, sd_, amp = (int(x[1:]) for x in name.split("_"))
        g = torch.Generator().manual_seed(sd_)
        return amp / 1000.0 * torch.randn(1, n, generator=g), 24000

    ta.load = fake_load
    UE = importlib.import_module("f5_tts.eval.utils_eval")
    UM = sys.modules["f5_tts.model.utils"]
    rows = [ln.strip().split(",") for ln in open(os.path.join(HERE, "c4_durations.csv")) if ln[0] != "#" and ln.strip()]
    rng = np.random.default_rng(4242)

    def ascii_text(nbytes, punct):
        words = []
        while sum(len(w) + 1 for w in words) < nbytes + 8:
            words.append("".join(rng.choice(list("abcdefghijklmnopqrstuvwxyz"), size=int(rng.integers(1, 9)))))
        t = " ".join(words)[:max(1, nbytes - 1)].rstrip()
        return (t[0].upper() + t[1:] + punct)[:nbytes] if nbytes > 1 else "a"

    meta = []
    for i, (rs, rb, gs, gb) in enumerate(rows[:72]):
        n = int(float(rs) * 24000)
        ptxt = ascii_text(int(rb), ".;'" [i % 3] if i % 5 else "")      # some prompts end in a letter, ';' is translated
        gtxt = ascii_text(int(gb), ".")
        if i % 7 == 0:
            gtxt = gtxt.replace(" ", "; ", 1).replace("a", "it's ", 1)
        meta.append((f"utt{i:03d}", ptxt, f"/nowhere/n{n}_s{1000 + i}_a{30 + 40 * (i % 4)}.wav", " " + gtxt, ""))
    out = {"meta": [[u, p, w, g] for u, p, w, g, _ in meta], "cases": []}
    for bs, nb in ((1, 200), (3000, 200), (2500, 16)):
        pr = UE.get_inference_prompt(meta, infer_batch_size=bs, num_buckets=nb)
        out["cases"].append({"infer_batch_size": bs, "num_buckets": nb,
                             "batches": [{"utts": list(b[0]), "ref_mel_lens": [int(x) for x in b[3]],
                                          "total_mel_lens": [int(x) for x in b[4]],
                                          "mel_shape": list(b[2].shape), "ref_rms": [float(x) for x in b[1]],
                                          "tokens_first": b[5][0]} for b in pr]})
    out["pinyin_ascii"] = []
    for t in ("end.Next one;ok", "it's 2 fast", "Hello, World!  Two  spaces", "a;b;c", "x", "Numbers 123 and-dashes_under",
              'Quote "this" and that: ok', meta[0][1] + " " + meta[0][3]):
        out["pinyin_ascii"].append({"text": t, "tokens": UM.convert_char_to_pinyin([t])[0]})
    path = os.path.join(HERE, "eval_prompts.json")
    with open(path, "w") as f:
        json.dump(out, f, indent=0)
    print("wrote", path, [len(c["batches"]) for c in out["cases"]])


def make_unett_case():
    """Reference UNetT.forward (backbones/unett.py) on two small seeded configs."""
    unett_mod = importlib.import_module("f5_tts.model.backbones.unett")
    for tag, skip, b in (("concat_b1", "concat", 1), ("add_b2", "add", 2)):
        torch.manual_seed(900)
        m = unett_mod.UNetT(dim=128, depth=4, heads=2, dim_head=64, ff_mult=2, mel_dim=20, text_num_embeds=50,
                            text_dim=32, conv_layers=2, skip_connect_type=skip).eval()
        g = torch.Generator().manual_seed(901)
        for name, p in m.named_parameters():
            if name.endswith(".g") or "grn" in name:
                p.data.add_(0.1 * torch.randn(p.shape, generator=g))
        n = 40
        x, cond = torch.randn(b, n, 20, generator=g), torch.randn(b, n, 20, generator=g)
        text = torch.randint(0, 50, (b, 9), generator=g)
        mask = None
        if b > 1:
            text[1, 6:] = -1
            mask = torch.arange(n)[None] < torch.tensor([n, n - 6])[:, None]
        out = {"x": x, "cond": cond, "text": text, "mask": mask, "time": torch.tensor(0.41)}
        with torch.no_grad():
            for drop in (False, True):
                out["pred_drop%d" % int(drop)] = m(x, cond, text, torch.tensor(0.41), drop_audio_cond=drop,
                                                   drop_text=drop, mask=mask)
        out.update({"w/" + k: v for k, v in m.state_dict().items()})
        path = os.path.join(HERE, f"unett_{tag}.npz")
        np.savez_compressed(path, **_np(out))
        print("wrote", path, os.path.getsize(path) // 1024, "KiB")


def make_vq_case(modules_mod):
    """Eval forward of the reference GumbelVectorQuantizer (three constructor variants)."""
    out = {}
    for tag, kw in (("plain", dict(groups=2, combine_groups=False, weight_proj_depth=1)),
                    ("combine", dict(groups=2, combine_groups=True, weight_proj_depth=1)),
                    ("deep", dict(groups=4, combine_groups=False, weight_proj_depth=2, weight_proj_factor=2))):
        torch.manual_seed(77)
        m = modules_mod.GumbelVectorQuantizer(dim=32, num_vars=10, temp=(2, 0.5, 0.999995), vq_dim=32, time_first=True,
                                              **kw).eval()
        x = torch.randn(2, 13, 32)
        with torch.no_grad():
            r = m(x, produce_targets=True)
        out[f"{tag}/x"] = x
        out[f"{tag}/q"] = r["x"]
        out[f"{tag}/targets"] = r["targets"]
        out[f"{tag}/code_perplexity"] = r["code_perplexity"]
        out[f"{tag}/prob_perplexity"] = r["prob_perplexity"]
        for k, v in m.state_dict().items():
            out[f"{tag}/w/{k}"] = v
    path = os.path.join(HERE, "vq_eval.npz")
    np.savez_compressed(path, **_np(out))
    print("wrote", path)


def make_ppg_embed_transformer_case(dit_mod):
    """The ``use_transformer=True`` PPGEmbedding of the reference (backbones/dit.py:105-119) on seeded weights."""
    torch.manual_seed(909)
    m = dit_mod.PPGEmbedding(ppg_dim=32, text_dim=48, use_transformer=True,
                             transformer_config=dict(nhead=4, dim_feedforward=64, dropout=0.1, num_layers=2)).eval()
    g = torch.Generator().manual_seed(910)
    with torch.no_grad():
        for p_ in m.parameters():
            if p_.ndim == 1:
                p_.add_(0.1 * torch.randn(p_.shape, generator=g))
        ppg = torch.randn(2, 9, 32, generator=g)
        out = {"ppg": ppg, "out": m(ppg, 14), "out_drop": m(ppg, 14, drop_ppg=True), "out_none": m(None, 14, batch=2)}
    out.update(_sd(m))
    path = os.path.join(HERE, "ppg_embed_transformer.npz")
    np.savez_compressed(path, **_np(out))
    print("wrote", path, tuple(out["out"].shape))


def make_stft_case(ref_root):
    """Pins for the transform part of a14 (log-mel front-end) and a17 (Vocos iSTFT head) from REFERENCE-HELD code:
    the reference restates both as convolutions for its ONNX export -- ``runtime/triton_trtllm/scripts/conv_stft.py``
    (``STFT.transform`` :156-191, ``STFT.inverse`` :193-234) and ``export_vocoder_to_onnx.py:45-59`` (``ISTFTHead``).
    Both import only torch + scipy (+ a ``vocos`` name that is never touched on this path: an empty stand-in module
    object is registered for the import statement, no arithmetic comes from it).  Fixture: seeded waves -> (real, imag,
    magnitude) of the 1024 / 256 hann STFT; seeded head pre-activations z -> audio of ISTFTHead.forward."""
    scripts = os.path.join(ref_root, "src", "f5_tts", "runtime", "triton_trtllm", "scripts")

    def load(name):
        spec = importlib.util.spec_from_file_location(name, os.path.join(scripts, name + ".py"))
        mod = importlib.util.module_from_spec(spec)
        sys.modules[name] = mod
        spec.loader.exec_module(mod)
        return mod

    conv_stft = load("conv_stft")
    if "vocos" not in sys.modules:
        v = types.ModuleType("vocos")
        v.Vocos = type("Vocos", (), {})
        sys.modules["vocos"] = v
    export = load("export_vocoder_to_onnx")
    g = torch.Generator().manual_seed(1414)
    out = {}
    stft = conv_stft.STFT(win_len=1024, win_hop=256, fft_len=1024)
    for tag, nw in (("a", 256 * 11 + 128), ("b", 256 * 40 + 7)):
        wav = 0.1 * torch.randn(2, nw, generator=g)
        wav = F.avg_pool1d(wav[:, None], 5, stride=1, padding=2)[:, 0]
        real, imag = stft.transform(wav, return_type="realimag")
        mag, _ = stft.transform(wav, return_type="magphase")
        out.update({f"fwd_{tag}/wav": wav, f"fwd_{tag}/real": real, f"fwd_{tag}/imag": imag, f"fwd_{tag}/mag": mag})
    for tag, b, t in (("a", 1, 9), ("b", 2, 33)):
        head = export.ISTFTHead(1024, 256)
        head.out = nn.Identity()     # the head's Linear is F.linear (pinned with every other linear); what is pinned here is
        with torch.no_grad():        # exp / clip / cos / sin + the inverse transform on 513 log-magnitudes | 513 phases
            z = torch.cat((1.5 * torch.randn(b, t, 513, generator=g), 3.0 * torch.randn(b, t, 513, generator=g)), -1)
            z[:, 0, :7] = 6.0        # exp(6) > 100: exercises the clip
            audio = head(z)
        out.update({f"inv_{tag}/z": z, f"inv_{tag}/audio": audio})
    path = os.path.join(HERE, "stft_head.npz")
    np.savez_compressed(path, **_np(out))
    print("wrote", path, {k: tuple(v.shape) for k, v in out.items() if k.endswith(("mag", "audio"))})


def make_ppg_case(ref_root):
    """Pins for SURVEY row f3 from the reference's own PPG classes: ``ppg/asr_model.py`` (``init_asr_model`` ->
    ``ASRModel.extract``: wenet ConformerEncoder + the 256-d ``linear`` head + ``ce.fc`` logits, with ``GlobalCMVN``) and
    ``ppg/ppg_model.py`` (``PPGModelWapper.mel_to_ppg`` / ``ppg_to_target``).  A reduced encoder (64-d, 4 heads, 2 blocks,
    same module classes / code paths as the 256-d default) keeps the fixture small; every parameter and BatchNorm buffer
    is seeded-random.  kaldi fbank itself lives in torchaudio (absent): features here are seeded noise of fbank shape, and
    ``torchaudio.compliance.kaldi`` is registered as an EMPTY stand-in only so that ``ppg_model.py`` imports."""
    src = os.path.join(ref_root, "src", "f5_tts")
    if "f5_tts" not in sys.modules or not hasattr(sys.modules["f5_tts"], "__path__"):
        pkg = types.ModuleType("f5_tts")
        pkg.__path__ = [src]
        sys.modules["f5_tts"] = pkg
    pk = types.ModuleType("f5_tts.ppg")
    pk.__path__ = [os.path.join(src, "ppg")]
    sys.modules["f5_tts.ppg"] = pk
    for name in ("torchaudio", "torchaudio.transforms", "torchaudio.compliance", "torchaudio.compliance.kaldi"):
        if name not in sys.modules:
            sys.modules[name] = types.ModuleType(name)
    sys.modules["torchaudio"].transforms = sys.modules["torchaudio.transforms"]
    sys.modules["torchaudio"].compliance = sys.modules["torchaudio.compliance"]
    sys.modules["torchaudio.compliance"].kaldi = sys.modules["torchaudio.compliance.kaldi"]
    asr = importlib.import_module("f5_tts.ppg.asr_model")
    ppg_model = importlib.import_module("f5_tts.ppg.ppg_model")
    cmvn_mod = importlib.import_module("f5_tts.ppg.wenet.transformer.cmvn")
    cfg = dict(cmvn_file=None, is_json_cmvn=True, input_dim=80, output_dim=40, encoder="conformer", decoder="transformer",
               encoder_conf=dict(output_size=64, attention_heads=4, linear_units=128, num_blocks=2),
               decoder_conf=dict(attention_heads=4, linear_units=64, num_blocks=1),
               model_conf=dict(ctc_weight=0.3, lsm_weight=0.1, length_normalized_loss=False, sv_conf=dict(use_sv=False)))
    torch.manual_seed(4242)
    model = asr.init_asr_model(cfg)
    g = torch.Generator().manual_seed(4243)
    model.encoder.global_cmvn = cmvn_mod.GlobalCMVN(torch.randn(80, generator=g), 0.5 + torch.rand(80, generator=g))
    with torch.no_grad():
        for name, p in model.named_parameters():
            if p.ndim == 1:          # LayerNorm / BatchNorm affine, biases: away from their 1 / 0 defaults
                p.add_(0.1 * torch.randn(p.shape, generator=g))
        for name, b in model.named_buffers():
            if name.endswith("running_mean"):
                b.copy_(0.1 * torch.randn(b.shape, generator=g))
            if name.endswith("running_var"):
                b.copy_(1.0 + 0.2 * torch.rand(b.shape, generator=g))
    model.eval()
    feats = 4.0 * torch.randn(2, 101, 80, generator=g) + 8.0
    lens = torch.tensor([101, 77])
    feats[1, 77:] = 0.0                      # padded frames of a batch are zeros (pad_sequence), before CMVN
    with torch.no_grad():
        ppg, logits = model.extract(feats, lens, stream=False)
        wrap = object.__new__(ppg_model.PPGModelWapper)       # __init__ loads files from absolute paths: skipped
        wrap.ppg_model, wrap.output_type, wrap.map_mix_ratio = model, "ppg", 1.0
        wrap.ppg_frame_length, wrap.mel_f_shift, wrap.device = 20, 10, "cpu"
        tgt, true_len = wrap.mel_to_ppg(feats, lens)
    keep = ("encoder.embed.", "encoder.encoders.", "encoder.after_norm.", "encoder.global_cmvn.", "linear.", "ce.fc.")
    out = {"w/" + k: v for k, v in model.state_dict().items() if k.startswith(keep) and "concat_linear" not in k}
    out.update({"feats": feats, "lens": lens, "ppg": ppg, "logits": logits, "target": tgt, "true_len": true_len})
    path = os.path.join(HERE, "ppg_conformer.npz")
    np.savez_compressed(path, **_np(out))
    print("wrote", path, tuple(ppg.shape), tuple(logits.shape), true_len.tolist())


def make_layouts(dit_mod):
    """state_dict key -> shape of the full-size models as the REFERENCE constructs them (no weights: names and shapes
    are what `load_checkpoint(strict=True)` needs): F5TTS_v1_Base, and BASELINE config 5 (Small + PPG + codebook)."""
    import json
    base = dict(dim=1024, depth=22, heads=16, ff_mult=2, text_dim=512, conv_layers=4, text_num_embeds=2545, mel_dim=100)
    small = dict(dim=768, depth=18, heads=12, ff_mult=2, text_dim=512, text_mask_padding=False, conv_layers=4,
                 pe_attn_head=1, text_num_embeds=2545, mel_dim=100,
                 ppg_config=dict(use_ppg=True, ppg_dim=256, use_cross_mask=False, cross_mask_config={},
                                 use_transformer=False, transformer_config={}),
                 cb_config=dict(use_codebook=True, num_vars=100, temp_start=2, temp_stop=0.5, temp_decay=0.999995,
                                groups=2, combine_groups=False, weight_proj_depth=1, weight_proj_factor=1,
                                use_perplex_loss=False, perplex_loss_config={}, use_align_loss=False,
                                align_loss_config={}))
    out = {}
    for tag, kw in (("v1_base", base), ("small_ppg_codebook", small)):
        out[tag] = {k: list(v.shape) for k, v in dit_mod.DiT(**kw).state_dict().items()}
    with open(os.path.join(HERE, "layouts.json"), "w") as f:
        json.dump(out, f, indent=0, sort_keys=True)
    print("layouts.json", {k: len(v) for k, v in out.items()})


def main():
    ref = sys.argv[1] if len(sys.argv) > 1 else "/root/reference"
    modules_mod, dit_mod, cfm_mod, utils_mod = load_reference(ref)
    if len(sys.argv) > 2 and sys.argv[2] == "layouts":
        make_layouts(dit_mod)
        return
    if len(sys.argv) > 2 and sys.argv[2] == "callers":
        make_callers_case(ref, cfm_mod)
        return
    if len(sys.argv) > 2 and sys.argv[2] == "eval":
        make_eval_case(ref, cfm_mod)
        return
    if len(sys.argv) > 2 and sys.argv[2] == "vq":
        make_vq_case(modules_mod)
        return
    if len(sys.argv) > 2 and sys.argv[2] == "unett":
        make_unett_case()
        return
    if len(sys.argv) > 2 and sys.argv[2] == "ppgtr":
        make_ppg_embed_transformer_case(dit_mod)
        return
    if len(sys.argv) > 2 and sys.argv[2] == "stft":
        make_stft_case(ref)
        return
    if len(sys.argv) > 2 and sys.argv[2] == "ppg":
        make_ppg_case(ref)
        return
    small = dict(dim=128, depth=2, heads=2, dim_head=64, ff_mult=2, mel_dim=20, text_num_embeds=50, text_dim=32,
                 conv_layers=2)
    make_dit_case(dit_mod, cfm_mod, "b1", small, b=1, n=48, nc=17, nt=9, steps=4, cfg_strength=2.0, seed=100)
    make_dit_case(dit_mod, cfm_mod, "b2_mask", small, b=2, n=64, nc=21, nt=12, steps=3, cfg_strength=2.0, seed=200)
    make_dit_case(dit_mod, cfm_mod, "b1_midpoint", small, b=1, n=40, nc=13, nt=7, steps=2, cfg_strength=1.5,
                  seed=300, method="midpoint")
    pe = dict(small, pe_attn_head=1, text_mask_padding=False, qk_norm="rms_norm", long_skip_connection=True)
    make_dit_case(dit_mod, cfm_mod, "b2_ppg_tts", pe, b=2, n=48, nc=15, nt=10, steps=2, cfg_strength=2.0, seed=400,
                  n_ppg=25, mode="tts")
    make_dit_case(dit_mod, cfm_mod, "b1_ppg_vc", pe, b=1, n=40, nc=15, nt=10, steps=2, cfg_strength=2.0, seed=500,
                  n_ppg=21, mode="vc")
    make_prep_case(cfm_mod, dit_mod)
    make_callers_case(ref, cfm_mod)
    make_eval_case(ref, cfm_mod)
    make_vq_case(modules_mod)
    make_unett_case()
    make_stft_case(ref)
    make_ppg_embed_transformer_case(dit_mod)
    make_ppg_case(ref)
    make_layouts(dit_mod)


if __name__ == "__main__":
    main()
