"""Callers of the hot path (mirror of reference infer/utils_infer.py): checkpoint/model/vocoder loading and
``infer_process`` / ``infer_batch_process`` with the reference's argument names, defaults and duration / RMS /
cross-fade rules.  Everything between ``model_obj.sample`` and ``vocoder.decode`` runs on libf5e_hip.so; what is left
here is host glue (text chunking, wav I/O, numpy cross-fade) exactly as in the reference.

Resampling and the pydub-style silence clipping live in ``infer/audio.py``.  Not rebuilt (out of scope,
SURVEY section 8): the Whisper ASR fallback of ``preprocess_ref_audio_text`` (an empty ``ref_text`` is an error here),
the bigvgan vocoder, HF-hub downloads (no network).  Chinese g2p needs the optional ``jieba`` + ``pypinyin`` packages;
the ASCII path is self-contained.
"""
from __future__ import annotations

import hashlib
import os
import re
import tempfile
import wave
from concurrent.futures import ThreadPoolExecutor
from typing import List, Optional, Tuple

import numpy as np
import torch

from ..model import CFM
from ..model.utils import get_tokenizer
from ..vocoder import Vocos, load_vocos
from . import audio as A

# ----------------------------------------- defaults (reference infer/utils_infer.py:49-62)
device = "cuda" if torch.cuda.is_available() else "cpu"
target_sample_rate = 24000
n_mel_channels = 100
hop_length = 256
win_length = 1024
n_fft = 1024
mel_spec_type = "vocos"
target_rms = 0.1
cross_fade_duration = 0.15
ode_method = "euler"
nfe_step = 32
cfg_strength = 2.0
sway_sampling_coef = -1.0
speed = 1.0
fix_duration = None

_DEFAULT_VOCAB = os.path.join(os.path.dirname(os.path.abspath(__file__)), "examples", "vocab.txt")


# ----------------------------------------- text front-end

def chunk_text(text: str, max_chars: int = 135) -> List[str]:
    """Sentence-wise greedy packing into chunks of at most ``max_chars`` UTF-8 bytes (reference utils_infer.py:70-97):
    split after ``;:,.!?`` followed by whitespace, or after a full-width ``；：，。！？``; a sentence ending in a
    single-byte character gets a trailing space when appended."""
    pieces = re.split(r"(?<=[;:,.!?])\s+|(?<=[；：，。！？])", text)
    chunks: List[str] = []
    cur = ""
    for sent in pieces:
        add = sent + " " if sent and len(sent[-1].encode("utf-8")) == 1 else sent
        if len(cur.encode("utf-8")) + len(sent.encode("utf-8")) <= max_chars:
            cur += add
        else:
            if cur:
                chunks.append(cur.strip())
            cur = add
    if cur:
        chunks.append(cur.strip())
    return chunks


_ASCII_TOKEN = re.compile(r"[A-Za-z0-9]+|.", re.DOTALL)
_OOV_TRANS = str.maketrans({";": ",", "“": '"', "”": '"', "‘": "'", "’": "'"})


def convert_char_to_pinyin(text_list: List[str], polyphone: bool = True) -> List[List[str]]:
    """Character/pinyin tokeniser (reference model/utils.py:270-311).  For single-byte text jieba yields alphanumeric
    runs and single other characters; that segmentation is restated here, so ASCII needs no third-party package.
    Text with multi-byte characters defers to jieba + pypinyin when they are installed."""
    out = []
    for text in text_list:
        text = text.translate(_OOV_TRANS)
        if all(ord(c) < 128 for c in text):
            chars: List[str] = []
            for seg in _ASCII_TOKEN.findall(text):
                if chars and len(seg) > 1 and chars[-1] not in " :'\"":
                    chars.append(" ")
                chars.extend(seg)
            out.append(chars)
            continue
        try:
            import jieba
            from pypinyin import Style, lazy_pinyin
        except ImportError as e:  # pragma: no cover - optional dependency
            raise RuntimeError("non-ASCII text needs the optional jieba + pypinyin packages (SURVEY f1)") from e
        chars = []
        for seg in jieba.cut(text):
            nb = len(seg.encode("utf-8"))
            if nb == len(seg):
                if chars and nb > 1 and chars[-1] not in " :'\"":
                    chars.append(" ")
                chars.extend(seg)
            elif polyphone and nb == 3 * len(seg):
                py = lazy_pinyin(seg, style=Style.TONE3, tone_sandhi=True)
                for i, c in enumerate(seg):
                    if "㄀" <= c <= "鿿":
                        chars.append(" ")
                    chars.append(py[i])
            else:
                for c in seg:
                    if ord(c) < 256:
                        chars.extend(c)
                    elif "㄀" <= c <= "鿿":
                        chars.append(" ")
                        chars.extend(lazy_pinyin(c, style=Style.TONE3, tone_sandhi=True))
                    else:
                        chars.append(c)
        out.append(chars)
    return out


# ----------------------------------------- loading (reference utils_infer.py:101-271)

def load_vocoder(vocoder_name="vocos", is_local=False, local_path="", device=device, hf_cache_dir=None):
    if vocoder_name != "vocos":
        raise NotImplementedError("only the vocos vocoder is built for MI355X")
    if not is_local:
        raise RuntimeError("no network access: pass is_local=True and local_path=<vocos-mel-24khz directory>")
    return load_vocos(local_path, device)


def load_checkpoint(model, ckpt_path: str, device: str, dtype=None, use_ema=True):
    """EMA prefix strip, bookkeeping/legacy key drops and strict load as in reference utils_infer.py:185-227.
    The module keeps fp32 master weights whatever ``dtype`` says: the HIP engine makes its own bf16 repack (the
    reference's fp16 cast would degrade the RoPE / time tables)."""
    ckpt_type = ckpt_path.split(".")[-1]
    if ckpt_type == "safetensors":
        from safetensors.torch import load_file
        checkpoint = load_file(ckpt_path, device="cpu")
    else:
        checkpoint = torch.load(ckpt_path, map_location="cpu", weights_only=True)
    if use_ema:
        if ckpt_type == "safetensors":
            checkpoint = {"ema_model_state_dict": checkpoint}
        state = {k.replace("ema_model.", ""): v for k, v in checkpoint["ema_model_state_dict"].items()
                 if k not in ("initted", "step")}
        for legacy in ("mel_spec.mel_stft.mel_scale.fb", "mel_spec.mel_stft.spectrogram.window"):
            state.pop(legacy, None)
    else:
        if ckpt_type == "safetensors":
            checkpoint = {"model_state_dict": checkpoint}
        state = checkpoint["model_state_dict"]
    model.load_state_dict({k: v.float() if v.is_floating_point() else v for k, v in state.items()})
    return model.to(device)


def load_model(model_cls, model_cfg, ckpt_path, mel_spec_type=mel_spec_type, vocab_file="", ode_method=ode_method,
               use_ema=True, device=device):
    if vocab_file == "":
        vocab_file = _DEFAULT_VOCAB
    vocab_char_map, vocab_size = get_tokenizer(vocab_file)
    model = CFM(
        transformer=model_cls(**model_cfg, text_num_embeds=vocab_size, mel_dim=n_mel_channels),
        mel_spec_kwargs=dict(n_fft=n_fft, hop_length=hop_length, win_length=win_length,
                             n_mel_channels=n_mel_channels, target_sample_rate=target_sample_rate,
                             mel_spec_type=mel_spec_type),
        odeint_kwargs=dict(method=ode_method), vocab_char_map=vocab_char_map).to(device)
    if ckpt_path:
        model = load_checkpoint(model, ckpt_path, device, use_ema=use_ema)
    return model


# ----------------------------------------- audio I/O (stdlib wave instead of torchaudio/soundfile)

def load_wav(path: str) -> Tuple[torch.Tensor, int]:
    """PCM16 / PCM32 / float32 RIFF -> (float32 [channels, n], sample_rate)."""
    with wave.open(path, "rb") as w:
        nch, sw, sr, n = w.getnchannels(), w.getsampwidth(), w.getframerate(), w.getnframes()
        raw = w.readframes(n)
    if sw == 2:
        data = np.frombuffer(raw, dtype="<i2").astype(np.float32) / 32768.0
    elif sw == 4:
        data = np.frombuffer(raw, dtype="<i4").astype(np.float32) / 2147483648.0
    else:
        raise ValueError(f"unsupported sample width {sw}")
    return torch.from_numpy(data.reshape(-1, nch).T.copy()), sr


def save_wav(path: str, audio: np.ndarray, sr: int) -> None:
    pcm = np.clip(np.asarray(audio, dtype=np.float64), -1.0, 1.0)
    pcm = (pcm * 32767.0).round().astype("<i2")
    with wave.open(path, "wb") as w:
        w.setnchannels(1)
        w.setsampwidth(2)
        w.setframerate(sr)
        w.writeframes(pcm.tobytes())


def _read_segment(path: str) -> A.Segment:
    wav, sr = load_wav(path)
    return A.Segment.from_float(wav.numpy(), sr)


def _write_segment(path: str, seg: A.Segment) -> None:
    with wave.open(path, "wb") as w:
        w.setnchannels(seg.pcm.shape[1])
        w.setsampwidth(2)
        w.setframerate(seg.rate)
        w.writeframes(np.ascontiguousarray(seg.pcm, dtype="<i2").tobytes())


_ref_audio_cache = {}


def preprocess_ref_audio_text(ref_audio_orig, ref_text, clip_short=True, show_info=print):
    """Reference infer/utils_infer.py:293-352: clip the clip to <= 12 s at silences, strip silent edges, add 50 ms of
    silence, write a temporary wav; make ref_text end in ". ".  Returns (wav path, ref_text)."""
    show_info("Converting audio...")
    seg = A.clip_reference(_read_segment(ref_audio_orig), clip_short=clip_short, note=show_info)
    with tempfile.NamedTemporaryFile(delete=False, suffix=".wav") as f:
        ref_audio = f.name
    _write_segment(ref_audio, seg)
    with open(ref_audio, "rb") as fh:
        audio_hash = hashlib.md5(fh.read()).hexdigest()
    if not ref_text.strip():
        if audio_hash in _ref_audio_cache:
            show_info("Using cached reference text...")
            ref_text = _ref_audio_cache[audio_hash]
        else:
            raise ValueError("ref_text is empty: the reference falls back to Whisper ASR here (utils_infer.py:341), "
                             "which is outside this build; pass the transcript of the reference audio")
    else:
        show_info("Using custom reference text...")
    if not ref_text.endswith(". ") and not ref_text.endswith("\u3002"):
        ref_text += " " if ref_text.endswith(".") else ". "
    return ref_audio, ref_text


def remove_silence_for_generated_wav(filename):
    """Reference infer/utils_infer.py:567-575."""
    _write_segment(filename, A.strip_generated_silence(_read_segment(filename)))


def cross_fade_concat(waves: List[np.ndarray], cross_fade_duration: float, sr: int = target_sample_rate) -> np.ndarray:
    """Linear cross-fade between consecutive chunk waves (reference utils_infer.py:520-556)."""
    if cross_fade_duration <= 0:
        return np.concatenate(waves)
    final = waves[0]
    for nxt in waves[1:]:
        n = min(int(cross_fade_duration * sr), len(final), len(nxt))
        if n <= 0:
            final = np.concatenate([final, nxt])
            continue
        mix = final[-n:] * np.linspace(1, 0, n) + nxt[:n] * np.linspace(0, 1, n)
        final = np.concatenate([final[:-n], mix, nxt[n:]])
    return final


# ----------------------------------------- inference drivers (reference utils_infer.py:367-565)

def plan_batch(ref_audio_len: int, ref_text: str, gen_text: str, speed_: float, fix_duration_) -> Tuple[int, float]:
    """Duration heuristic of process_batch (reference utils_infer.py:455-471) -> (duration in frames, speed used)."""
    local_speed = 0.3 if len(gen_text.encode("utf-8")) < 10 else speed_
    if fix_duration_ is not None:
        return int(fix_duration_ * target_sample_rate / hop_length), local_speed
    ref_text_len = len(ref_text.encode("utf-8"))
    gen_text_len = len(gen_text.encode("utf-8"))
    return ref_audio_len + int(ref_audio_len / ref_text_len * gen_text_len / local_speed), local_speed


def infer_batch_process(ref_audio, ref_text, gen_text_batches, model_obj, vocoder, mel_spec_type="vocos", progress=None,
                        target_rms=0.1, cross_fade_duration=0.15, nfe_step=32, cfg_strength=2.0,
                        sway_sampling_coef=-1, speed=1, fix_duration=None, device=None, streaming=False,
                        chunk_size=2048):
    """Generator like the reference's: yields (final_wave, sample_rate, combined_spectrogram) or, when streaming,
    (chunk, sample_rate) pieces."""
    audio, sr = ref_audio
    if audio.shape[0] > 1:
        audio = torch.mean(audio, dim=0, keepdim=True)
    rms = torch.sqrt(torch.mean(torch.square(audio)))
    if rms < target_rms:
        audio = audio * target_rms / rms
    if sr != target_sample_rate:
        audio = A.resample(audio, sr, target_sample_rate)
    audio = audio.to(device)
    if len(ref_text[-1].encode("utf-8")) == 1:
        ref_text = ref_text + " "

    def process_batch(gen_text, seed=None):
        """Queues one chunk on the GPU and returns DEVICE tensors: nothing here waits for the GPU, so the host prep of
        the next chunk overlaps this chunk's ODE loop; the caller copies to the host afterwards."""
        text_list = convert_char_to_pinyin([ref_text + gen_text])
        ref_audio_len = audio.shape[-1] // hop_length
        duration, _ = plan_batch(ref_audio_len, ref_text, gen_text, speed, fix_duration)
        with torch.inference_mode():
            generated, _traj = model_obj.sample(cond=audio, text=text_list, duration=duration, steps=nfe_step,
                                                cfg_strength=cfg_strength, sway_sampling_coef=sway_sampling_coef,
                                                seed=seed)
            del _traj
            generated = generated.to(torch.float32)[:, ref_audio_len:, :].permute(0, 2, 1)
            wave_ = vocoder.decode(generated)
            if rms < target_rms:
                wave_ = wave_ * rms / target_rms
            return wave_, generated

    def to_host(res):
        wave_, generated = res
        return wave_.squeeze().cpu().numpy(), generated[0].cpu().numpy()

    if streaming:
        for gen_text in gen_text_batches:
            w, _ = to_host(process_batch(gen_text))
            for j in range(0, len(w), chunk_size):
                yield w[j: j + chunk_size], target_sample_rate
        return
    # The reference submits process_batch to a ThreadPoolExecutor (utils_infer.py:511), but process_batch is a generator
    # function there: submit() only creates the generator, and the chunks then run ONE AT A TIME, in order, in the consumer
    # loop (`next(result)`, :514-518).  Default here = the same order on the calling thread, so with seed=None the noise
    # of chunk i is the i-th draw from the global CPU generator exactly as in the reference (reproducible under
    # torch.manual_seed); the GPU still overlaps chunks with the host because nothing above waits for it.
    # F5E_INFER_WORKERS=k (2..4) opts into k host threads, each on its own stream (the engine keeps
    # per-call buffers private per thread): more throughput on one GPU, and a per-chunk seed drawn in submission order on
    # this thread keeps the result independent of thread timing (a different noise stream than the serial default).
    workers = max(1, min(4, int(os.environ.get("F5E_INFER_WORKERS", "1"))))
    if workers == 1 or len(gen_text_batches) < 2:
        results = [to_host(r) for r in [process_batch(g) for g in gen_text_batches]]
    else:
        seeds = [int(torch.randint(0, 2 ** 31 - 1, (1,)).item()) for _ in gen_text_batches]
        import threading
        tls = threading.local()
        main_stream = torch.cuda.current_stream()

        def on_own_stream(a):
            # the samplers run on the caller's current stream: every worker brings its own, ordered behind this thread's
            if not hasattr(tls, "stream"):
                tls.stream = torch.cuda.Stream()
                tls.stream.wait_stream(main_stream)
            with torch.cuda.stream(tls.stream):
                return to_host(process_batch(*a))

        with ThreadPoolExecutor(max_workers=workers) as ex:
            results = list(ex.map(on_own_stream, zip(gen_text_batches, seeds)))
    waves = [r[0] for r in results]
    specs = [r[1] for r in results]
    if waves:
        yield cross_fade_concat(waves, cross_fade_duration), target_sample_rate, np.concatenate(specs, axis=1)
    else:
        yield None, target_sample_rate, None


def infer_process(ref_audio, ref_text, gen_text, model_obj, vocoder, mel_spec_type=mel_spec_type, show_info=print,
                  progress=None, target_rms=target_rms, cross_fade_duration=cross_fade_duration, nfe_step=nfe_step,
                  cfg_strength=cfg_strength, sway_sampling_coef=sway_sampling_coef, speed=speed,
                  fix_duration=fix_duration, device=device):
    audio, sr = load_wav(ref_audio)
    secs = audio.shape[-1] / sr
    max_chars = int(len(ref_text.encode("utf-8")) / secs * (22 - secs))
    gen_text_batches = chunk_text(gen_text, max_chars=max_chars)
    show_info(f"Generating audio in {len(gen_text_batches)} batches...")
    return next(infer_batch_process((audio, sr), ref_text, gen_text_batches, model_obj, vocoder,
                                    mel_spec_type=mel_spec_type, progress=progress, target_rms=target_rms,
                                    cross_fade_duration=cross_fade_duration, nfe_step=nfe_step,
                                    cfg_strength=cfg_strength, sway_sampling_coef=sway_sampling_coef, speed=speed,
                                    fix_duration=fix_duration, device=device))
