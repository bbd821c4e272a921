#!/usr/bin/env python3
"""bench.py -- mel-frames/sec of the F5TTS_v1_Base flow-matching hot path on MI355X (BASELINE.json metric).

  python bench.py --gpus N --steps K --warmup W
  (N > 1: launched by torch.distributed.run, one rank per GPU; RCCL is used for barriers + one MAX all-reduce only --
   utterances are independent)

A "step" is one full pass of the hot path over one batch: log-mel front-end (HIP STFT) -> 32 Euler steps of the
CFG-batched DiT (hipGraph replay) -> stitch -> Vocos decode on the GPU, for BASELINE config C2:
F5TTS_v1_Base, bf16 MFMA contractions, batch 1, N_ref = 188 frames (2 s), N = 469 frames (5 s), NFE 32, CFG 2.0,
sway -1, seeded random-init weights (AdaLN-zero tensors re-randomised), synthetic audio / token ids.
Inputs are resident in HBM when the timed region starts; outputs stay on the device.

Prints ONE JSON line (rank 0) with the contract fields plus
  roofline     -- the dominant kernel (QKV/FF bf16 MFMA GEMM class chosen by total time) timed per launch with HIP
                  events on the launch stream during an eager, in-situ pass of the same workload;
  cpu_baseline -- the CPU oracle ("port") on a bounded sample of the same workload on this box's host cores.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

N_REF, N_TOTAL, NFE, CFG, SWAY = 188, 469, 32, 2.0, -1.0
BATCH = 1
PEAK_BF16_TFLOPS = 2500.0  # dense bf16 MFMA peak, MI355X_MICROARCH.md chip table


def gemm_flops(op, M, D=1024, FF=2048):
    return {"QKV": 2.0 * M * D * 3 * D, "OUT": 2.0 * M * D * D, "FF1": 2.0 * M * D * FF, "FF2": 2.0 * M * FF * D}[op]


def log(msg):
    print(f"[bench {time.strftime('%H:%M:%S')}] {msg}", file=sys.stderr, flush=True)


def host_threads():
    """CPU share of this process: affinity mask capped at 16 (the GPU box gives 16 cores per GPU; os.cpu_count()
    reports the whole host and oversubscribes OpenMP)."""
    try:
        n = len(os.sched_getaffinity(0))
    except AttributeError:
        n = os.cpu_count() or 1
    return max(1, min(n, 16))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-roofline", action="store_true")
    ap.add_argument("--streams", type=int, default=4,
                    help="extra leg (1 GPU only, reported as `concurrent`, never as `value`): this many host threads sample "
                         "batch-1 utterances concurrently on one model, as the reference's infer_batch_process does "
                         "with its ThreadPoolExecutor; 0 skips it")
    ap.add_argument("--workload", default="C2", choices=["C2", "C3", "C4", "C5"],
                    help="C2 (default, the graded line): batch 1, 2 s ref / 5 s total; C3: batch 32, 4 s / 10 s; "
                         "C4: eval_infer_batch-style stream of single utterances (LibriSpeech-PC length mix, NFE 16, "
                         "LPT-sharded over the ranks, a step = one utterance); C5: Small + PPG config, sample_vc, NFE 32")
    args = ap.parse_args()
    global N_REF, N_TOTAL, BATCH, NFE
    if args.workload == "C3":
        N_REF, N_TOTAL, BATCH = 375, 938, 32
    if args.workload == "C4":
        NFE = 16
    if args.workload in ("C4", "C5"):
        args.no_roofline = True   # the in-situ GEMM timing is set up for the v1_Base shapes of C2 / C3

    import torch
    import torch.distributed as dist

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    distributed = world > 1 or "TORCHELASTIC_RUN_ID" in os.environ  # under torch.distributed.run: always RCCL
    torch.cuda.set_device(local_rank)
    # host-side prep (seeded noise, masks, token ids) is a few tiny CPU ops per pass: keep every rank on a small, fixed
    # number of threads so 8 ranks on one node do not oversubscribe the host with OpenMP teams
    torch.set_num_threads(max(1, min(4, host_threads() // max(1, int(os.environ.get("LOCAL_WORLD_SIZE", world))))))
    if distributed:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        # RCCL writes its banner / warnings to stdout: send them to a file so stdout carries exactly one JSON line
        os.environ["NCCL_DEBUG"] = os.environ.get("F5E_NCCL_DEBUG", "WARN")
        os.environ.setdefault("NCCL_DEBUG_FILE", "/tmp/f5e_rccl_%h_%p.log")
        dist.init_process_group("nccl", device_id=torch.device(f"cuda:{local_rank}"))  # RCCL on ROCm

    from f5e_tts_amd import ops
    from f5e_tts_amd._C import OP_FF1, OP_FF2, OP_OUT, OP_QKV
    from f5e_tts_amd.engine import KernelTimer
    from f5e_tts_amd.model import CFM, DiT
    from f5e_tts_amd.vocoder import Vocos
    from oracle import f5e_oracle as O  # only for seeded synthetic inputs/weights and the cpu_baseline leg

    ops.require_device()
    if args.workload == "C5":   # BASELINE config 5: configs/F5TTS_Small_PPG.yaml (dim 768, 18 blocks, PPG input)
        cfg = O.DiTConfig(dim=768, depth=18, heads=12, ff_mult=2, text_dim=512, conv_layers=4, text_num_embeds=2545,
                          text_mask_padding=False, pe_attn_head=1, use_ppg=True, ppg_dim=256)
        ppg_config = dict(use_ppg=True, ppg_dim=256, use_transformer=False)
        dit = DiT(dim=768, depth=18, heads=12, ff_mult=2, text_dim=512, conv_layers=4, text_num_embeds=2545,
                  text_mask_padding=False, pe_attn_head=1, ppg_config=ppg_config)
        dit.load_state_dict(SY.init_dit_state(cfg, 1234), strict=True)
        cfm = CFM(transformer=dit, ppg_config=ppg_config).cuda().eval()
    else:
        cfg = O.DiTConfig()
        sd = SY.init_dit_state(cfg, 1234)
        dit = DiT(dim=1024, depth=22, heads=16, ff_mult=2, text_dim=512, conv_layers=4, text_num_embeds=2545)
        dit.load_state_dict(sd, strict=True)
        cfm = CFM(transformer=dit).cuda().eval()
    vs = SY.init_vocos_state()
    voc = Vocos()
    voc.load_state_dict(vs, strict=False)
    voc = voc.cuda().eval()
    wav = SY.synthetic_ref_wave(N_REF, batch=BATCH).cuda()
    text = SY.synthetic_text_ids(N_TOTAL, batch=BATCH)   # token ids stay on the host, as the reference's callers hand them

    def one_pass():
        mel, _ = cfm.sample(wav, text, duration=N_TOTAL, steps=NFE, cfg_strength=CFG, sway_sampling_coef=SWAY, seed=0)
        return voc.decode(mel[:, N_REF:, :].permute(0, 2, 1))

    step_frames = None       # per-step (total, generated) frame counts when the steps differ (C4)
    if args.workload == "C5":
        g5 = torch.Generator().manual_seed(55)
        ppg = torch.randn(1, round(0.533 * N_TOTAL), 256, generator=g5).cuda()

        def one_pass():  # noqa: F811  reference eval_infer_batch_vc.py:214-224 (alpha_spk 2.5, alpha_ppg 3)
            mel, _ = cfm.sample_vc(wav, ppg, duration=N_TOTAL, steps=NFE, alpha_spk=2.5, alpha_ppg=3.0,
                                   sway_sampling_coef=SWAY, seed=0)
            return voc.decode(mel[:, N_REF:, :].permute(0, 2, 1))
    if args.workload == "C4":
        from f5e_tts_amd.eval.eval_infer_batch import flop_fwd, lpt_partition
        utts = []
        with open(os.path.join(ROOT, "tests", "golden", "c4_durations.csv")) as f:
            for line in f:
                if line.startswith("#"):
                    continue
                rs, rb, _gs, gb = line.split(",")
                ref_len = int(float(rs) * 24000) // 256
                utts.append((ref_len, ref_len + int(ref_len / (int(rb) + 1) * int(gb))))   # utils_eval.py:337-339
        need = world * (args.steps + args.warmup)
        utts = [utts[i % len(utts)] for i in range(need)]
        mine = [utts[i] for i in lpt_partition([flop_fwd(t) for _, t in utts], world)[rank]]
        inputs = [(SY.synthetic_ref_wave(r).cuda(), SY.synthetic_text_ids(t), r, t) for r, t in mine]
        step_frames = [(t, t - r) for _, _, r, t in inputs[args.warmup:]]
        cursor = [0]

        def one_pass():  # noqa: F811  one utterance per call, as reference eval_infer_batch.py:196-216
            w, ids, r, t = inputs[cursor[0] % len(inputs)]
            cursor[0] += 1
            mel, _ = cfm.sample(w, ids, duration=t, steps=NFE, cfg_strength=CFG, sway_sampling_coef=SWAY, seed=0)
            return voc.decode(mel[:, r:t, :].permute(0, 2, 1))

    log(f"model ready on cuda:{local_rank}, world {world}")
    latency_ms = None
    isolated_out = None
    for i in range(args.warmup):
        torch.cuda.synchronize()
        tw = time.perf_counter()
        isolated_out = one_pass()
        torch.cuda.synchronize()
        latency_ms = (time.perf_counter() - tw) * 1e3   # one isolated pass, host prep not overlapped
        log(f"warmup {i} done ({latency_ms:.1f} ms isolated)")
    torch.cuda.synchronize()
    if distributed:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    last_out = None
    for _ in range(args.steps):
        last_out = one_pass()
    torch.cuda.synchronize()
    if distributed:
        dist.barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    # the timed passes are queued back to back (host prep of pass i+1 overlaps the ODE loop of pass i): same inputs and
    # seed, so the last of them must reproduce the isolated, synchronised warm-up pass bit for bit
    pipelined_ok = None
    if isolated_out is not None and step_frames is None:
        pipelined_ok = bool(torch.equal(last_out, isolated_out))
    if pipelined_ok is False:
        raise SystemExit("bench: pipelined pass differs from the isolated pass")
    if distributed:
        tt = torch.tensor([elapsed], device="cuda", dtype=torch.float64)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        elapsed = float(tt.item())

    log(f"timed region: {args.steps} steps in {elapsed:.3f} s")

    # Concurrency leg: the batch-1 chain is latency-bound, so independent utterances in flight on separate
    # capture streams overlap each other's launch gaps.  Throughput only -- each pass takes longer -- and kept out of `value`.
    concurrent = None
    if world == 1 and args.streams > 1 and step_frames is None:
        from concurrent.futures import ThreadPoolExecutor
        n_conc = -(-max(args.steps, 8 * args.streams) // args.streams) * args.streams   # a whole number of rounds
        with ThreadPoolExecutor(max_workers=args.streams) as ex:
            list(ex.map(lambda _: one_pass(), range(args.streams)))      # per-thread stream / graph warm-up
            torch.cuda.synchronize()
            tc = time.perf_counter()
            outs = list(ex.map(lambda _: one_pass(), range(n_conc)))
            torch.cuda.synchronize()
            dtc = time.perf_counter() - tc
        same = all(bool(torch.equal(o, last_out)) for o in outs)
        if not same:
            raise SystemExit("bench: concurrent passes differ from the sequential ones")
        concurrent = {"streams": args.streams, "passes": n_conc, "value": round(n_conc * N_TOTAL * BATCH / dtc, 2),
                      "unit": "mel-frames/s", "ms_per_pass_amortised": round(dtc / n_conc * 1e3, 3),
                      "bit_identical_to_sequential": same}
        log(f"concurrent leg: {args.streams} threads, {n_conc} passes in {dtc:.3f} s")
        del outs
    frames = world * args.steps * N_TOTAL * BATCH
    gen_frames = world * args.steps * BATCH * (N_TOTAL - N_REF)
    if step_frames is not None:   # C4: every rank ran its own utterances
        tt = torch.tensor([sum(a for a, _ in step_frames[:args.steps]), sum(b for _, b in step_frames[:args.steps])],
                          device="cuda", dtype=torch.float64)
        if distributed:
            dist.all_reduce(tt, op=dist.ReduceOp.SUM)
        frames, gen_frames = int(tt[0].item()), int(tt[1].item())
    gen_audio_s = gen_frames * 256 / 24000.0
    value = frames / elapsed

    roofline = None
    if rank == 0 and not args.no_roofline:
        # in-situ per-launch timing: eager launches of the SAME kernels in the SAME order, one op class at a time
        n_chains = getattr(dit.engine(), "last_n_chains", 1)
        M = 2 * N_TOTAL * BATCH // n_chains          # rows per launch (the CFG branches may run as parallel chains)
        cfm.use_graph = False
        per_op = {}
        for name, op in (("QKV", OP_QKV), ("OUT", OP_OUT), ("FF1", OP_FF1), ("FF2", OP_FF2)):
            tm = KernelTimer(op, capacity=NFE * 22 * n_chains + 8)
            cfm.kernel_timer = tm
            one_pass()
            torch.cuda.synchronize()
            ms = tm.read_ms()
            ms_sorted = sorted(ms)
            med = ms_sorted[len(ms) // 2]   # median: an eager pass has host-launch hiccups that inflate the mean
            per_op[name] = (med, len(ms), sum(ms) / len(ms))
            log(f"roofline pass {name}: {len(ms)} launches, median {med * 1e3:.1f} us, mean {per_op[name][2] * 1e3:.1f} us")
        cfm.kernel_timer = None
        cfm.use_graph = True
        dom = max(per_op, key=lambda k: per_op[k][0] * per_op[k][1])
        avg_ms = per_op[dom][0]
        ach = gemm_flops(dom, M) / (avg_ms * 1e-3) / 1e12
        # HBM traffic per launch of that kernel: rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of this same command,
        # corrected per the microarch guide (tools/pmc_traffic.py); the committed summary is read back here
        traffic = None
        tpath = os.path.join(ROOT, "profiles", f"pmc_traffic_{args.workload.lower()}.json")
        if os.path.exists(tpath):
            import re
            epi = {"QKV": 3, "OUT": 2, "FF1": 1, "FF2": 2}[dom]
            cands = [(v["launches"], v["hbm_bytes_per_launch"]) for k, v in json.load(open(tpath))["kernels"].items()
                     if re.match(rf"gemm_bf16_kernel<\d+, \d+, {epi},", k)]
            if cands:
                traffic = max(cands)[1]
        roofline = {"bound": "mfma", "achieved": round(ach, 2), "peak": PEAK_BF16_TFLOPS, "unit": "TFLOP/s",
                    "frac": round(ach / PEAK_BF16_TFLOPS, 4), "traffic": traffic,
                    "kernel": {"QKV": "gemm_bf16_kernel<*,*,EPI_QKV_ROPE>", "OUT": "gemm_bf16_kernel<*,*,EPI_GATE_RES>",
                               "FF1": "gemm_bf16_kernel<*,*,EPI_BF16_GELU>", "FF2": "gemm_bf16_kernel<*,*,EPI_GATE_RES>"}[dom],
                    "op": dom, "avg_launch_us": round(avg_ms * 1e3, 2), "avg_is": "median of per-launch HIP-event intervals",
                    "mean_launch_us": round(per_op[dom][2] * 1e3, 2), "launches_timed": per_op[dom][1],
                    "flop_per_launch": gemm_flops(dom, M), "rows_per_launch": M, "parallel_chains": n_chains,
                    "all_gemm_avg_us": {k: round(v[0] * 1e3, 2) for k, v in per_op.items()}}

    cpu_baseline = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline and args.workload == "C2":
        sample_steps = NFE  # the full workload once (~15-25 s on 16 host threads)
        torch.set_num_threads(host_threads())
        log(f"cpu baseline on {torch.get_num_threads()} threads")
        wav_c, text_c = wav.cpu(), text.cpu()
        with torch.inference_mode():
            c0 = time.perf_counter()
            out_c, _ = O.cfm_sample(sd, cfg, wav_c, text_c, None, N_TOTAL, steps=sample_steps, cfg_strength=CFG,
                                    sway_sampling_coef=SWAY, seed=0)
            c_loop = time.perf_counter() - c0
            c1 = time.perf_counter()
            O.vocos_decode(vs, out_c[:, N_REF:].permute(0, 2, 1))
            c_voc = time.perf_counter() - c1
        log(f"cpu baseline: loop sample {c_loop:.2f} s, vocoder {c_voc:.2f} s")
        est = c_loop * (NFE / sample_steps) + c_voc  # one-off parts (mel, text embed) are < 1% of c_loop
        cpu_baseline = {"value": round(N_TOTAL / est, 3), "unit": "mel-frames/s", "cores": torch.get_num_threads(),
                        "kind": "port",
                        "sample": f"oracle fp32, the same B=1 N={N_TOTAL} workload once: mel + {sample_steps} Euler steps "
                                  f"({2 * sample_steps} DiT forwards, {c_loop:.2f} s) + Vocos decode ({c_voc:.2f} s), "
                                  "no warm-up"}

    workload_desc = (f"{args.workload}: F5TTS_v1_Base random-init, batch {BATCH} per GPU, N_ref={N_REF} N={N_TOTAL} frames, "
                     f"euler NFE={NFE}, CFG={CFG} (cond+uncond batched), sway={SWAY}, hipGraph ODE step, "
                     "HIP log-mel front-end + Vocos decode on GPU")
    if args.workload == "C4":
        workload_desc = ("C4: F5TTS_v1_Base random-init, one utterance per call, lengths from the reference's LibriSpeech-PC "
                         f"cross-sentence list (750-1875 frames), euler NFE={NFE}, CFG={CFG}, sway={SWAY}, LPT-sharded over "
                         f"{world} rank(s), hipGraph ODE step, HIP log-mel + Vocos on GPU")
    if args.workload == "C5":
        workload_desc = (f"C5: F5TTS_Small + PPG (dim 768, 18 blocks) random-init, batch 1, N_ref={N_REF} N={N_TOTAL} frames, "
                         f"sample_vc (3 branches batched, alpha_spk 2.5, alpha_ppg 3), euler NFE={NFE}, sway={SWAY}, "
                         "hipGraph ODE step, HIP log-mel + Vocos on GPU")
    if rank == 0:
        line = {
            "metric": "mel_frames_per_sec", "value": round(value, 2), "unit": "mel-frames/s", "n_gpus": world,
            "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(elapsed / args.steps * 1e3, 3),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "bf16", "data": "synthetic",
            "config": {"workload": workload_desc,
                       "frames_per_step": (N_TOTAL * BATCH) if step_frames is None else round(frames / (world * args.steps), 1), "parallelism": f"replica x{world} (utterance sharding, no collective "
                                                                  "on the data path)"},
            "rtf": round(elapsed / gen_audio_s, 5),
            "isolated_pass_ms": None if latency_ms is None else round(latency_ms, 2),
            "pipelined_equals_isolated": pipelined_ok,
            "generated_mel_frames_per_sec": round(gen_frames / elapsed, 2),
        }
        if concurrent is not None:
            line["concurrent"] = concurrent
        if roofline is not None:
            line["roofline"] = roofline
        if cpu_baseline is not None:
            line["cpu_baseline"] = cpu_baseline
        print(json.dumps(line), flush=True)
    if distributed:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
