"""Command-line entry with the argument surface of reference infer/infer_cli.py:34-171 and its three-layer setting
resolution (flag > TOML > module default, including the ``x or default`` quirk that lets falsy flag values such as
``--cfg_strength 0`` fall through, reference :181-211).  Model/vocoder weights are local files only (no network)."""
from __future__ import annotations

import argparse
import os
import re
from datetime import datetime

import numpy as np

from . import utils_infer as U

_PKG = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def build_parser() -> argparse.ArgumentParser:
    p = argparse.ArgumentParser(prog="python3 infer-cli.py",
                                description="Commandline interface for E2/F5 TTS with Advanced Batch Processing.",
                                epilog="Specify options above to override one or more settings from config.")
    p.add_argument("-c", "--config", type=str, default="", help="TOML configuration file")
    p.add_argument("-m", "--model", type=str, help="The model name: F5TTS_v1_Base | ...")
    p.add_argument("-mc", "--model_cfg", type=str, help="The path to the model config file .yaml")
    p.add_argument("-p", "--ckpt_file", type=str, help="The path to model checkpoint .pt/.safetensors")
    p.add_argument("-v", "--vocab_file", type=str, help="The path to vocab file .txt")
    p.add_argument("-r", "--ref_audio", type=str, help="The reference audio file.")
    p.add_argument("-s", "--ref_text", type=str, help="The transcript/subtitle for the reference audio")
    p.add_argument("-t", "--gen_text", type=str, help="The text to make model synthesize a speech")
    p.add_argument("-f", "--gen_file", type=str, help="The file with text to generate, will ignore --gen_text")
    p.add_argument("-o", "--output_dir", type=str, help="The path to output folder")
    p.add_argument("-w", "--output_file", type=str, help="The name of output file")
    p.add_argument("--save_chunk", action="store_true", help="To save each audio chunks during inference")
    p.add_argument("--remove_silence", action="store_true", help="To remove long silence found in output")
    p.add_argument("--load_vocoder_from_local", action="store_true", help="To load vocoder from local dir")
    p.add_argument("--vocoder_name", type=str, choices=["vocos", "bigvgan"], help="vocoder")
    p.add_argument("--target_rms", type=float, help="Target output speech loudness normalization value")
    p.add_argument("--cross_fade_duration", type=float, help="Duration of cross-fade between audio segments in seconds")
    p.add_argument("--nfe_step", type=int, help="The number of function evaluation (denoising steps)")
    p.add_argument("--cfg_strength", type=float, help="Classifier-free guidance strength")
    p.add_argument("--sway_sampling_coef", type=float, help="Sway Sampling coefficient")
    p.add_argument("--speed", type=float, help="The speed of the generated audio")
    p.add_argument("--fix_duration", type=float, help="Fix the total duration (ref and gen audios) in seconds")
    p.add_argument("--device", type=str, help="Specify the device to run on")
    return p


def resolve_settings(args: argparse.Namespace, config: dict) -> dict:
    """flag > toml > default, with the reference's falsy-``or`` semantics (only ref_text uses ``is not None``)."""
    g = config.get
    s = dict(
        model=args.model or g("model", "F5TTS_v1_Base"),
        model_cfg=args.model_cfg or g("model_cfg", ""),
        ckpt_file=args.ckpt_file or g("ckpt_file", ""),
        vocab_file=args.vocab_file or g("vocab_file", ""),
        ref_audio=args.ref_audio or g("ref_audio", "infer/examples/basic/basic_ref_en.wav"),
        ref_text=args.ref_text if args.ref_text is not None
        else g("ref_text", "Some call me nature, others call me mother nature."),
        gen_text=args.gen_text or g("gen_text", "Here we generate something just for test."),
        gen_file=args.gen_file or g("gen_file", ""),
        output_dir=args.output_dir or g("output_dir", "tests"),
        output_file=args.output_file or g("output_file", f"infer_cli_{datetime.now().strftime(r'%Y%m%d_%H%M%S')}.wav"),
        save_chunk=args.save_chunk or g("save_chunk", False),
        remove_silence=args.remove_silence or g("remove_silence", False),
        load_vocoder_from_local=args.load_vocoder_from_local or g("load_vocoder_from_local", False),
        vocoder_name=args.vocoder_name or g("vocoder_name", U.mel_spec_type),
        target_rms=args.target_rms or g("target_rms", U.target_rms),
        cross_fade_duration=args.cross_fade_duration or g("cross_fade_duration", U.cross_fade_duration),
        nfe_step=args.nfe_step or g("nfe_step", U.nfe_step),
        cfg_strength=args.cfg_strength or g("cfg_strength", U.cfg_strength),
        sway_sampling_coef=args.sway_sampling_coef or g("sway_sampling_coef", U.sway_sampling_coef),
        speed=args.speed or g("speed", U.speed),
        fix_duration=args.fix_duration or g("fix_duration", U.fix_duration),
        device=args.device or g("device", U.device),
    )
    return s


def load_arch(model: str, model_cfg: str) -> dict:
    import yaml
    path = model_cfg or os.path.join(_PKG, "configs", f"{model}.yaml")
    with open(path, "r") as f:
        cfg = yaml.safe_load(f)
    arch = dict(cfg["model"]["arch"])
    arch.pop("checkpoint_activations", None)
    return arch


def split_voices(gen_text: str):
    """``[voice]`` tags split the text into (voice, text) chunks (reference infer_cli.py:306-321)."""
    out = []
    for chunk in re.split(r"(?=\[\w+\])", gen_text):
        if not chunk.strip():
            continue
        m = re.match(r"\[(\w+)\]", chunk)
        out.append((m[1] if m else "main", re.sub(r"\[(\w+)\]", "", chunk).strip()))
    return out


def main(argv=None):
    args = build_parser().parse_args(argv)
    config = {}
    if args.config:
        import tomli
        with open(args.config, "rb") as f:
            config = tomli.load(f)
    s = resolve_settings(args, config)
    if s["gen_file"]:
        with open(s["gen_file"], "r", encoding="utf-8") as f:
            s["gen_text"] = f.read()
    from ..model import DiT
    vocoder = U.load_vocoder(s["vocoder_name"], is_local=True,
                             local_path=config.get("vocoder_local_path", "pretrained_models/vocos-mel-24khz"),
                             device=s["device"])
    model = U.load_model(DiT, load_arch(s["model"], s["model_cfg"]), s["ckpt_file"], mel_spec_type=s["vocoder_name"],
                         vocab_file=s["vocab_file"], device=s["device"])
    voices = dict(config.get("voices", {}))
    voices["main"] = {"ref_audio": s["ref_audio"], "ref_text": s["ref_text"]}
    for v in voices.values():   # reference infer_cli.py:297-303
        v["ref_audio"], v["ref_text"] = U.preprocess_ref_audio_text(v["ref_audio"], v["ref_text"])
    segments = []
    for voice, text in split_voices(s["gen_text"]):
        v = voices.get(voice, voices["main"])
        seg, sr, _ = U.infer_process(v["ref_audio"], v["ref_text"], text, model, vocoder,
                                     mel_spec_type=s["vocoder_name"], target_rms=s["target_rms"],
                                     cross_fade_duration=s["cross_fade_duration"], nfe_step=s["nfe_step"],
                                     cfg_strength=s["cfg_strength"], sway_sampling_coef=s["sway_sampling_coef"],
                                     speed=s["speed"], fix_duration=s["fix_duration"], device=s["device"])
        segments.append(seg)
        if s["save_chunk"]:
            os.makedirs(os.path.join(s["output_dir"], "chunks"), exist_ok=True)
            U.save_wav(os.path.join(s["output_dir"], "chunks", f"{len(segments) - 1}.wav"), seg, sr)
    if segments:
        os.makedirs(s["output_dir"], exist_ok=True)
        path = os.path.join(s["output_dir"], s["output_file"])
        U.save_wav(path, np.concatenate(segments), U.target_sample_rate)
        if s["remove_silence"]:
            U.remove_silence_for_generated_wav(path)
        print(path)


if __name__ == "__main__":
    main()
