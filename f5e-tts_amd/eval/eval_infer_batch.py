"""Multi-GPU batch inference driver (mirror of reference eval/eval_infer_batch.py:184-223 + the bucketing of
eval/utils_eval.py:77-219), one process per GPU under ``torch.distributed`` (backend "nccl" = RCCL on ROCm).

The path shards by utterance with NO data-path collective: RCCL carries only the start/stop barriers and
one 16-byte all-reduce of (frames, seconds) -- weights are replicated, every rank writes its own wavs.  Differences
from the reference that do not change any output file: each rank prepares only ITS share of the prompts (the
reference prepares all prompts on every rank), and the default partition is longest-processing-time on the analytic
FLOP cost instead of contiguous slices (``--partition contiguous`` restores the reference's
``split_between_processes``).
"""
from __future__ import annotations

import argparse
import math
import os
import random
import time
from typing import List, Sequence, Tuple

import torch


# ----------------------------------------------------------------------------- pure partition / bucketing logic

def flop_fwd(n: int) -> float:
    """Analytic FLOPs of one F5TTS_v1_Base DiT forward at N frames (SURVEY 8d / BASELINE.md section 3)."""
    return 2.0 * (n * (189.44e6 + 45056.0 * n) + 141.8e6)


def split_between_processes(items: Sequence, rank: int, world: int) -> List:
    """HF accelerate semantics used at reference eval_infer_batch.py:187: contiguous slices, the first
    ``len % world`` ranks get one extra item."""
    n = len(items)
    base, extra = divmod(n, world)
    start = rank * base + min(rank, extra)
    end = start + base + (1 if rank < extra else 0)
    return list(items[start:end])


def lpt_partition(costs: Sequence[float], world: int) -> List[List[int]]:
    """Greedy longest-processing-time assignment of item indices to ranks (ties broken by index: deterministic)."""
    order = sorted(range(len(costs)), key=lambda i: (-costs[i], i))
    loads = [0.0] * world
    parts: List[List[int]] = [[] for _ in range(world)]
    for i in order:
        r = min(range(world), key=lambda k: (loads[k], k))
        parts[r].append(i)
        loads[r] += costs[i]
    return [sorted(p) for p in parts]


def bucket_batches(total_mel_lens: Sequence[int], infer_batch_size: int = 1, num_buckets: int = 200, min_secs: int = 3,
                   max_secs: int = 40, target_sample_rate: int = 24000, hop_length: int = 256,
                   shuffle_seed: int = 666) -> List[List[int]]:
    """Index-level restatement of get_inference_prompt's batching (reference eval/utils_eval.py:96-219):
    bucket = floor((len - min) / (max - min + 1) * num_buckets); a bucket is flushed once its accumulated frames
    reach ``infer_batch_size``; residual buckets are flushed in bucket order; batches shuffled with seed 666."""
    min_tokens = min_secs * target_sample_rate // hop_length
    max_tokens = max_secs * target_sample_rate // hop_length
    accum = [0] * num_buckets
    pending: List[List[int]] = [[] for _ in range(num_buckets)]
    batches: List[List[int]] = []
    for i, n in enumerate(total_mel_lens):
        if not (min_tokens <= n <= max_tokens):
            raise ValueError(f"item {i}: {n} frames outside [{min_tokens}, {max_tokens}]")
        b = math.floor((n - min_tokens) / (max_tokens - min_tokens + 1) * num_buckets)
        pending[b].append(i)
        accum[b] += n
        if accum[b] >= infer_batch_size:
            batches.append(pending[b])
            pending[b], accum[b] = [], 0
    for b in range(num_buckets):
        if accum[b] > 0:
            batches.append(pending[b])
    rng = random.Random(shuffle_seed)  # same stream as random.seed(666); random.shuffle(...)
    rng.shuffle(batches)
    return batches


def total_mel_len(ref_mel_len: int, prompt_text: str, gt_text: str, speed: float = 1.0) -> int:
    """reference eval/utils_eval.py:147-161 (prompt_text already carries its trailing space rule)."""
    return ref_mel_len + int(ref_mel_len / len(prompt_text.encode("utf-8")) * len(gt_text.encode("utf-8")) / speed)


def c4_work_list(csv_path: str, need: int) -> List[Tuple[int, int]]:
    """(ref frames, total frames) of `need` utterances, taken cyclically from a `ref_secs,ref_bytes,gen_secs,gen_bytes`
    table (tests/golden/c4_durations.csv: the duration / byte-count columns of the reference's LibriSpeech-PC
    cross-sentence list) with the reference's length rule (eval/utils_eval.py:147-161; the prompt text gains a trailing
    space, :130-131, and the LibriSpeech list reader hands the target text over with a leading one, :57)."""
    rows = []
    with open(csv_path) as f:
        for line in f:
            if line.startswith("#") or not line.strip():
                continue
            rs, rb, _gs, gb = line.split(",")
            ref_len = int(float(rs) * 24000) // 256
            rows.append((ref_len, ref_len + int(ref_len / (int(rb) + 1) * (int(gb) + 1))))
    return [rows[i % len(rows)] for i in range(need)]


def reduce_job_totals(dist, frames: float, gen_frames: float, seconds: float, device="cpu") -> dict:
    """Whole-job totals of a sharded run: SUM of the frame counts, MAX of the wall time, plus what every rank saw
    (so that a scaling record can show that `world` ranks really took part).  The only collectives of the path: a few
    dozen bytes after the timed region.  dist=None -> single process."""
    if dist is None:
        return dict(frames=frames, gen_frames=gen_frames, seconds=seconds, per_rank_frames=[frames],
                    per_rank_seconds=[seconds], world_size=1, backend=None)
    world = dist.get_world_size()
    mine = torch.tensor([float(frames), float(gen_frames), float(seconds)], dtype=torch.float64, device=device)
    every = [torch.zeros_like(mine) for _ in range(world)]
    dist.all_gather(every, mine)
    tot = mine.clone()
    dist.all_reduce(tot, op=dist.ReduceOp.SUM)
    mx = mine.clone()
    dist.all_reduce(mx, op=dist.ReduceOp.MAX)
    return dict(frames=float(tot[0]), gen_frames=float(tot[1]), seconds=float(mx[2]),
                per_rank_frames=[float(x[0]) for x in every], per_rank_seconds=[float(x[2]) for x in every],
                world_size=world, backend=dist.get_backend())


# ----------------------------------------------------------------------------- in-rank concurrency

class RankWorkers:
    """K host threads of ONE rank, each with its own HIP stream, pulling utterances off a shared cursor.

    Why: at batch 1 the block kernels are latency-bound (M = 2 N rows fill ~16 % of the GEMM roofline), so
    independent utterances in flight on separate streams overlap each other's launch gaps -- the MI355X counterpart of the
    reference's thread pool over chunks (infer/utils_infer.py:511) and of its per-rank batches (eval_infer_batch.py:187-216).
    Every utterance is still one `sample` call with its own seed, so each output is bit-identical to the sequential run
    whichever worker takes it (tests/test_e2e_gpu.py).  The threads persist for the object's life: the samplers' loop
    states (buffers, captured graphs) are keyed by (thread, stream), and a warm-up through the same object is what warms
    the timed run.  `workers` <= 1: plain loop on the caller's thread and current stream."""

    def __init__(self, workers: int = 1, device=None):
        import threading
        self.workers = max(1, int(workers))
        self.device = device
        self._tls = threading.local()
        self._pool = None
        if self.workers > 1:
            from concurrent.futures import ThreadPoolExecutor
            self._pool = ThreadPoolExecutor(max_workers=self.workers, thread_name_prefix="f5e-rank-worker")

    def _stream_ctx(self):
        import contextlib
        if not torch.cuda.is_available():
            return contextlib.nullcontext()
        st = getattr(self._tls, "stream", None)
        if st is None:
            st = self._tls.stream = torch.cuda.Stream(device=self.device)
        return torch.cuda.stream(st)

    def map(self, fn, items: Sequence) -> List:
        """[fn(x) for x in items], in input order; with K > 1 the calls are spread over the worker threads (dynamic: a
        free worker takes the next item).  Returns after every worker's stream has drained.  The first exception stops
        the hand-out and is re-raised."""
        items = list(items)
        if self._pool is None:
            return [fn(x) for x in items]
        import threading
        results, errors = [None] * len(items), []
        cursor, lock = [0], threading.Lock()

        def drain(_):
            with self._stream_ctx():
                while True:
                    with lock:
                        i = cursor[0]
                        cursor[0] += 1
                    if i >= len(items) or errors:
                        break
                    try:
                        results[i] = fn(items[i])
                    except BaseException as e:  # noqa: BLE001  (re-raised on the caller's thread)
                        errors.append(e)
                        break
                if torch.cuda.is_available():
                    torch.cuda.current_stream(self.device).synchronize()

        list(self._pool.map(drain, range(self.workers)))
        if errors:
            raise errors[0]
        return results

    def warm(self, fn, items: Sequence, rounds: int = 1) -> None:
        """Every worker thread runs fn over ALL of `items`, `rounds` times, behind a common barrier: deterministic per-thread
        warm-up (stream, allocator pool, loop states) -- a plain map() would leave it to chance which thread saw what."""
        items = list(items)
        if self._pool is None:
            for _ in range(rounds):
                for x in items:
                    fn(x)
            return
        import threading
        gate = threading.Barrier(self.workers)

        def each(_):
            gate.wait()
            with self._stream_ctx():
                for _r in range(rounds):
                    for x in items:
                        fn(x)
                if torch.cuda.is_available():
                    torch.cuda.current_stream(self.device).synchronize()

        list(self._pool.map(each, range(self.workers)))

    def close(self):
        if self._pool is not None:
            self._pool.shutdown(wait=True)
            self._pool = None

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()


# ----------------------------------------------------------------------------- distributed driver

def shard(items: Sequence, costs: Sequence[float], rank: int, world: int, mode: str = "lpt") -> List:
    if mode == "contiguous":
        return split_between_processes(items, rank, world)
    return [items[i] for i in lpt_partition(costs, world)[rank]]


def run_sharded(work: Sequence[Tuple[str, int, int]], process_one, rank: int, world: int, dist=None,
                mode: str = "lpt", workers: int = 1) -> dict:
    """``work``: (utt id, ref frames, total frames).  ``process_one(item)`` generates + writes one utterance.
    ``workers`` > 1: that many host threads of this rank, one HIP stream each, take the rank's utterances off a shared
    cursor (RankWorkers; process_one must be thread-safe -- the samplers are).
    Returns the whole-job totals after the closing all-reduce (works with any backend, incl. gloo on CPU)."""
    mine = shard(list(work), [flop_fwd(w[2]) for w in work], rank, world, mode)
    if dist is not None:
        dist.barrier()
    t0 = time.perf_counter()
    with RankWorkers(workers) as pool:
        pool.map(process_one, mine)          # returns after every worker's stream has drained
    frames = sum(item[2] for item in mine)
    elapsed = time.perf_counter() - t0
    done = [w[0] for w in mine]
    if dist is not None:
        dist.barrier()
    red = reduce_job_totals(dist, frames, sum(w[2] - w[1] for w in mine), elapsed,
                            "cuda" if dist is not None and dist.get_backend() == "nccl" else "cpu")
    return dict(red, utts=done)


def main(argv=None):
    p = argparse.ArgumentParser(description="batch inference")
    p.add_argument("-s", "--seed", default=None, type=int)
    p.add_argument("-n", "--expname", required=True)
    p.add_argument("-c", "--ckptstep", default=1250000, type=int)
    p.add_argument("-nfe", "--nfestep", default=32, type=int)
    p.add_argument("-o", "--odemethod", default="euler")
    p.add_argument("-ss", "--swaysampling", default=-1, type=float)
    p.add_argument("-t", "--testset", required=True, help="a .lst: ref_utt \\t ref_dur \\t ref_txt \\t gen_utt \\t gen_dur \\t gen_txt")
    p.add_argument("--ckpt", default="", help="checkpoint file (default ckpts/<expname>/model_<ckptstep>.pt)")
    p.add_argument("--audio_root", default="", help="directory with <ref_utt>.wav prompt audio (24 kHz)")
    p.add_argument("--vocoder_path", default="pretrained_models/vocos-mel-24khz")
    p.add_argument("--output_dir", default="")
    p.add_argument("--partition", default="lpt", choices=["lpt", "contiguous"])
    p.add_argument("--workers", default=4, type=int,
                   help="host threads per rank, one HIP stream each, sampling independent utterances concurrently "
                        "(outputs are bit-identical to --workers 1; 1 = the reference's one-utterance-at-a-time loop)")
    # the three drivers of the reference in one: eval_infer_batch.py (cfg), eval_infer_batch_tts.py (-as/-at) and
    # eval_infer_batch_vc.py (-as/-ap; PPG read from --ppg_dir/<gen_utt>.npy, the wenet extractor is SURVEY row f3)
    p.add_argument("--mode", default="cfg", choices=["cfg", "tts", "vc"])
    p.add_argument("-as", "--alpha_spk", default=2.5, type=float)
    p.add_argument("-at", "--alpha_txt", default=3, type=float)
    p.add_argument("-ap", "--alpha_ppg", default=3, type=float)
    p.add_argument("-mc", "--model_cfg", default="", help="model yaml (default configs/<expname>.yaml of this package)")
    p.add_argument("--ppg_dir", default="", help="vc mode: directory of precomputed <gen_utt>.npy PPG arrays [frames, ppg_dim]; "
                                                 "empty = extract them on the GPU with the PPG model of the model yaml")
    p.add_argument("--source_root", default="", help="vc mode without --ppg_dir: directory with the <gen_utt>.wav source "
                                                     "utterances whose content is converted (default: --audio_root)")
    p.add_argument("--ppg_model", default="", help="PPG checkpoint (default: ppg_config.model_path of the model yaml)")
    p.add_argument("--ppg_config", default="", help="PPG train.yaml (default: ppg_config.config of the model yaml)")
    args = p.parse_args(argv)

    import torch.distributed as dist

    from ..infer import utils_infer as U
    from ..model import CFM, DiT
    from ..model.utils import get_tokenizer
    from ..train.parse_cfg import parse_model_yaml

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    torch.cuda.set_device(local)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl")
    device = f"cuda:{local}"
    ckpt = args.ckpt or f"ckpts/{args.expname}/model_{args.ckptstep}.pt"
    import yaml
    cfg_path = args.model_cfg or os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "configs",
                                              f"{args.expname}.yaml")
    with open(cfg_path, "r") as f:
        mc = parse_model_yaml(yaml.safe_load(f))
    if args.mode == "vc" and not mc["transformer_ppg_config"]["use_ppg"]:
        raise SystemExit("--mode vc needs a model with use_ppg: True")
    vocab_char_map, vocab_size = get_tokenizer(U._DEFAULT_VOCAB)
    model = CFM(transformer=DiT(**mc["arch"], text_num_embeds=vocab_size, mel_dim=U.n_mel_channels,
                                ppg_config=mc["transformer_ppg_config"], cb_config=mc["transformer_codebook_config"]),
                mel_spec_kwargs=dict(n_fft=U.n_fft, hop_length=U.hop_length, win_length=U.win_length,
                                     n_mel_channels=U.n_mel_channels, target_sample_rate=U.target_sample_rate,
                                     mel_spec_type="vocos"),
                odeint_kwargs=dict(method=args.odemethod), vocab_char_map=vocab_char_map,
                ppg_config=mc["cfm_ppg_config"], cb_config=mc["cfm_codebook_config"]).to(device)
    model = U.load_checkpoint(model, ckpt, device, use_ema=True)
    vocoder = U.load_vocoder("vocos", is_local=True, local_path=args.vocoder_path, device=device)
    out_dir = args.output_dir or (f"results/{args.expname}_{args.ckptstep}/{os.path.basename(args.testset)}/"
                                  f"seed{args.seed}_{args.odemethod}_nfe{args.nfestep}_vocos_ss{args.swaysampling}"
                                  + {"cfg": "_cfg2.0_speed1.0",
                                     "tts": f"_alpha_spk{args.alpha_spk}_txt{args.alpha_txt}_speed1.0",
                                     "vc": f"_alpha_spk{args.alpha_spk}_ppg{args.alpha_ppg}_speed1.0"}[args.mode])
    if rank == 0:
        os.makedirs(out_dir, exist_ok=True)
    rows = []
    with open(args.testset, "r", encoding="utf-8") as f:
        for line in f:
            ref_utt, _rd, ref_txt, gen_utt, _gd, gen_txt = line.rstrip("\n").split("\t")
            # the reference's LibriSpeech-PC list reader prepends a space to the target text (eval/utils_eval.py:57); it
            # counts in the duration rule and shows up as a token between prompt and target
            rows.append((ref_utt, ref_txt, gen_utt, " " + gen_txt))
    # cheap metadata pass (wav headers only) so that every rank can build the same partition without decoding audio
    import wave
    work, meta = [], {}
    for ref_utt, ref_txt, gen_utt, gen_txt in rows:
        path = os.path.join(args.audio_root, ref_utt + ".wav")
        with wave.open(path, "rb") as w:
            ref_len = w.getnframes() // U.hop_length
        if len(ref_txt[-1].encode("utf-8")) == 1:
            ref_txt = ref_txt + " "
        tot = total_mel_len(ref_len, ref_txt, gen_txt)
        if args.mode == "vc" and not args.ppg_dir:
            # voice conversion keeps the source utterance's duration (reference eval/utils_eval.py:332-336, use_truth_duration)
            with wave.open(os.path.join(args.source_root or args.audio_root, gen_utt + ".wav"), "rb") as w:
                tot = ref_len + int(w.getnframes() * U.target_sample_rate / w.getframerate() / U.hop_length)
        work.append((gen_utt, ref_len, tot))
        meta[gen_utt] = (path, ref_txt, gen_txt)

    ppg_front = None
    if args.mode == "vc" and not args.ppg_dir:
        # reference eval_infer_batch_vc.py:131-141 + :281-328: PPG of [prompt ; source] at 16 kHz from the wenet extractor
        from ..ppg import PPGModelWapper
        fc = mc["frontend_ppg_config"]
        ppg_front = PPGModelWapper(args.ppg_model or fc["model_path"], args.ppg_config or fc["config"], device,
                                   output_type=fc["output_type"], ppg_frame_length=fc["frame_length"],
                                   mel_f_shift=fc["mel_frame_shift"], map_mix_ratio=fc["map_mix_ratio"],
                                   global_phn_center_path=fc["global_phn_center_path"],
                                   para_softmax_path=fc["para_softmax_path"])

    def extract_ppg(utt, ref_path):
        from ..infer import audio as A

        def mono16k(path):
            a, sr = U.load_wav(path)
            a = a.mean(0, keepdim=True)
            return A.resample(a, sr, 16000) if sr != 16000 else a

        full = torch.cat([mono16k(ref_path), mono16k(os.path.join(args.source_root or args.audio_root, utt + ".wav"))], dim=1)
        ppg, _len = ppg_front.audio_to_ppg(full.to(device), 16000)
        return ppg

    def process_one(item):
        utt, ref_len, tot = item
        path, ref_txt, gen_txt = meta[utt]
        audio, sr = U.load_wav(path)
        audio = audio.mean(0, keepdim=True)
        rms = torch.sqrt(torch.mean(torch.square(audio)))
        if rms < U.target_rms:
            audio = audio * U.target_rms / rms
        text = U.convert_char_to_pinyin([ref_txt + gen_txt])
        with torch.inference_mode():
            ref_mel = model.mel_spec(audio.to(device)).permute(0, 2, 1)[:, :ref_len]
            kw = dict(duration=torch.tensor([tot]), steps=args.nfestep, sway_sampling_coef=args.swaysampling,
                      seed=args.seed)
            if args.mode == "cfg":      # reference eval_infer_batch.py:196-206
                gen, _ = model.sample(cond=ref_mel, text=text, lens=torch.tensor([ref_len]), cfg_strength=2.0, **kw)
            elif args.mode == "tts":    # reference eval_infer_batch_tts.py:203-214
                gen, _ = model.sample_tts(cond=ref_mel, text=text, lens=torch.tensor([ref_len]),
                                          alpha_spk=args.alpha_spk, alpha_txt=args.alpha_txt, **kw)
            else:                       # reference eval_infer_batch_vc.py:214-224
                if ppg_front is not None:
                    ppg = extract_ppg(utt, path)
                else:
                    import numpy as np
                    ppg = torch.from_numpy(np.load(os.path.join(args.ppg_dir, utt + ".npy")).astype("float32"))[None]
                gen, _ = model.sample_vc(cond=ref_mel, ppg=ppg.to(device), alpha_spk=args.alpha_spk,
                                         alpha_ppg=args.alpha_ppg, **kw)
            wav = vocoder.decode(gen[:, ref_len:tot].permute(0, 2, 1).float())
        if rms < U.target_rms:
            wav = wav * rms / U.target_rms
        U.save_wav(os.path.join(out_dir, f"{utt}.wav"), wav[0].cpu().numpy(), U.target_sample_rate)

    res = run_sharded(work, process_one, rank, world, dist if world > 1 else None, args.partition, workers=args.workers)
    if rank == 0:
        print(f"Done batch inference in {res['seconds'] / 60:.2f} minutes: {res['frames'] / res['seconds']:.1f} "
              f"mel-frames/s on {world} GPU(s).")
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
