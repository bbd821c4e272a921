"""Parity of the HIP path at the FULL sizes of BASELINE.json's configs (C2..C5), each through its real caller:

  C2  F5TTS_v1_Base, batch 1, N_ref 188 / N 469, euler NFE 32, CFG 2, hipGraph loop + Vocos      -> vs the fp32 oracle
  C3  F5TTS_v1_Base, batch 32 x (N_ref 375 / N 938), CFG 2: (a) 4-step grid -> every item vs its own batch-1 run
      (size-independent property) and one item vs the fp32 oracle; (b) the config's own NFE 32 -> four items vs batch 1 and
      one item vs the fp32 oracle with the per-step trajectory check
  C4  eval_infer_batch.main() on three utterances of the C4 length mix (shortest / median / longest of the table: ~610 / 910 / 1390 frames), NFE 16 -> the written
      wavs vs a direct sample + decode (bit exact up to int16) and the shortest one vs the oracle's mel -> Vocos
  C5  configs/F5TTS_Small_PPG.yaml (dim 768, 18 blocks, PPG input, codebook keys), sample_vc, NFE 32 -> vs the fp32 oracle

Tolerances (stated, bf16 MFMA contractions against an fp32 oracle through 22 x 32 or 18 x 96 network evaluations):
relative L2 of the generated mel frames <= 3e-3 at a config's own step count (measured 0.9-1.4e-3: a 2.5 x regression fails;
round 3 stated 1e-2) and <= 5e-3 on the coarse 4-step grid of C3 (a) (its four large steps carry 2.1-2.7e-3), maximum
absolute error <= 5 % of the mel's dynamic range, and the
per-step error along the trajectory may not blow up (each step <= 4 x the previous one + 1e-4: on the sway grid the
step sizes grow, so early ratios of 2-3 are the plain accumulation of per-step rounding, an instability shows as 10 x).
The oracle legs cost ~15 s (C2), ~5 s (C3), ~15 s (C4), ~15 s (C5) of host CPU."""
import os

import numpy as np
import pytest
import torch

from oracle import f5e_oracle as O
from tools import synth as SY

pytestmark = pytest.mark.gpu

TOL_REL_L2 = 3e-3         # at the config's own NFE (16 / 32 steps)
TOL_COARSE_GRID = 5e-3    # the 4-step grid of test_c3_batch32_items_equal_their_batch1_runs_and_the_oracle
TOL_MAXABS_OF_RANGE = 0.05


def rel_l2(a, b):
    a, b = a.float().cpu(), b.float().cpu()
    return float((a - b).norm() / b.norm())


def full_model(seed=1234):
    from f5e_tts_amd.model import CFM, DiT
    cfg = O.DiTConfig()
    sd = SY.init_dit_state(cfg, seed)
    dit = DiT(dim=1024, depth=22, heads=16, ff_mult=2, text_dim=512, conv_layers=4, text_num_embeds=2545)
    dit.load_state_dict(sd, strict=True)
    return cfg, sd, dit, CFM(transformer=dit).cuda().eval()


def vocoder():
    from f5e_tts_amd.vocoder import Vocos
    vs = SY.init_vocos_state()
    voc = Vocos()
    voc.load_state_dict(vs, strict=False)
    return vs, voc.cuda().eval()


def check_trajectory(traj, ref_traj, n_ref, what, tol=TOL_REL_L2):
    steps = ref_traj.shape[0] - 1
    errs = [rel_l2(traj[i][:, n_ref:], ref_traj[i][:, n_ref:]) for i in range(1, steps + 1)]
    print(f"{what}: per-step rel L2 (generated frames):", " ".join("%.1e" % e for e in errs))
    assert errs[-1] < tol, (what, errs[-1])
    assert all(b < 4.0 * a + 1e-4 for a, b in zip(errs, errs[1:])), (what, errs)
    return errs


def test_c2_full_nfe32_graph_and_vocos_vs_oracle():
    """BASELINE C2 exactly as bench.py runs it."""
    cfg, sd, dit, cfm = full_model()
    vs, voc = vocoder()
    n_ref, n = 188, 469
    wav, text = SY.synthetic_ref_wave(n_ref), SY.synthetic_text_ids(n)
    kw = dict(duration=n, steps=32, cfg_strength=2.0, sway_sampling_coef=-1.0, seed=0)
    ref_out, ref_traj = O.cfm_sample(sd, cfg, wav, text, None, **kw)
    out, traj = cfm.sample(wav.cuda(), text, **kw)          # first call of a shape: eager launches
    out2, traj2 = cfm.sample(wav.cuda(), text, **kw)        # second call: one-step graph replayed 32 times
    out3, traj3 = cfm.sample(wav.cuda(), text, **kw)        # third call: the whole loop as ONE graph launch
    assert torch.equal(out, out2) and torch.equal(traj, traj2) and torch.equal(out, out3) and torch.equal(traj, traj3)
    assert out.shape == ref_out.shape == (1, n, 100) and traj.shape == ref_traj.shape == (33, 1, n, 100)
    assert torch.equal(traj[0].cpu(), ref_traj[0])          # seeded CPU noise is bit-identical
    torch.testing.assert_close(out[:, :n_ref].cpu(), ref_out[:, :n_ref], rtol=1e-4, atol=2e-4)   # stitched reference mel
    check_trajectory(traj, ref_traj, 0, "C2")
    gen, rgen = out[:, n_ref:].cpu(), ref_out[:, n_ref:]
    rng = float(rgen.max() - rgen.min())
    print("C2: final rel L2 %.3e, max abs %.3e of range %.3e" % (rel_l2(gen, rgen), float((gen - rgen).abs().max()), rng))
    assert rel_l2(gen, rgen) < TOL_REL_L2
    assert float((gen - rgen).abs().max()) < TOL_MAXABS_OF_RANGE * rng
    # Vocos on the GPU mel vs the oracle's Vocos on the same mel (fp32 both sides): 1e-3 of the peak
    wave = voc.decode(out[:, n_ref:].permute(0, 2, 1))
    ref_wave = O.vocos_decode(vs, out[:, n_ref:].permute(0, 2, 1).cpu())
    assert wave.shape == ref_wave.shape == (1, 256 * (n - n_ref - 1))
    assert float((wave.cpu() - ref_wave).abs().max()) < 1e-3 * float(ref_wave.abs().max())


def test_c3_batch32_items_equal_their_batch1_runs_and_the_oracle():
    """BASELINE C3 sizes (batch 32 x 10 s, 60 032 rows per launch: the 256 x 256 ping-pong GEMMs, the LDS-shared attention
    kernel, separate LayerNorms) against the batch-1 path (64 x 64 GEMMs, fused AdaLN) of the same items, on the first 4
    steps of the NFE-32 sway grid... the grid of a 4-step call differs, so the 4-step grid is what both sides integrate."""
    cfg, sd, dit, cfm = full_model()
    B, n_ref, n = 32, 375, 938
    wav, text = SY.synthetic_ref_wave(n_ref, batch=B), SY.synthetic_text_ids(n, batch=B)
    kw = dict(duration=n, steps=4, cfg_strength=2.0, sway_sampling_coef=-1.0, seed=4)
    full, ftraj = cfm.sample(wav.cuda(), text, **kw)
    assert full.shape == (B, n, 100) and torch.isfinite(full).all()
    worst = 0.0
    for i in range(B):
        one, _ = cfm.sample(wav[i:i + 1].cuda(), text[i:i + 1], **kw)
        e = rel_l2(full[i, n_ref:], one[0, n_ref:])
        worst = max(worst, e)
        assert e < TOL_COARSE_GRID, (i, e)
        assert torch.equal(full[i, :n_ref], one[0, :n_ref])     # stitched reference frames: copied, never computed
    print("C3: worst item-vs-batch-1 rel L2 %.3e" % worst)
    i = 17
    ref_out, ref_traj = O.cfm_sample(sd, cfg, wav[i:i + 1], text[i:i + 1], None, **kw)
    check_trajectory(ftraj[:, i:i + 1], ref_traj, n_ref, "C3 item 17 vs oracle", tol=TOL_COARSE_GRID)
    assert rel_l2(full[i, n_ref:], ref_out[0, n_ref:]) < TOL_COARSE_GRID


def test_c3_full_nfe32_batch32_vs_batch1_and_oracle():
    """BASELINE C3 at its OWN NFE: batch 32 x (N_ref 375 / N 938), euler NFE 32, CFG 2, sway -1 -- 60 032-row launches
    through all 32 steps of the ping-pong GEMM / LDS attention / separate-LayerNorm path (reference model/cfm.py:349-482).
    Four items against their batch-1 GPU runs at NFE 32 (size-independent property, the 64 x 64 GEMM + fused-AdaLN path)
    and one item against the fp32 oracle at NFE 32 (~35 s of host CPU), with the per-step trajectory check."""
    cfg, sd, dit, cfm = full_model()
    B, n_ref, n = 32, 375, 938
    wav, text = SY.synthetic_ref_wave(n_ref, batch=B), SY.synthetic_text_ids(n, batch=B)
    kw = dict(duration=n, steps=32, cfg_strength=2.0, sway_sampling_coef=-1.0, seed=0)
    full, ftraj = cfm.sample(wav.cuda(), text, **kw)
    assert full.shape == (B, n, 100) and ftraj.shape == (33, B, n, 100) and torch.isfinite(full).all()
    worst = 0.0
    for i in (0, 9, 17, 31):
        one, _ = cfm.sample(wav[i:i + 1].cuda(), text[i:i + 1], **kw)
        e = rel_l2(full[i, n_ref:], one[0, n_ref:])
        worst = max(worst, e)
        assert e < TOL_REL_L2, (i, e)
        assert torch.equal(full[i, :n_ref], one[0, :n_ref])
    print("C3 NFE 32: worst item-vs-batch-1 rel L2 %.3e" % worst)
    i = 17
    ref_out, ref_traj = O.cfm_sample(sd, cfg, wav[i:i + 1], text[i:i + 1], None, **kw)
    assert torch.equal(ftraj[0, i:i + 1].cpu(), ref_traj[0])          # seeded CPU noise is bit-identical per item
    check_trajectory(ftraj[:, i:i + 1], ref_traj, n_ref, "C3 NFE 32 item 17 vs oracle")
    gen, rgen = full[i, n_ref:].cpu(), ref_out[0, n_ref:]
    print("C3 NFE 32: item 17 final rel L2 %.3e" % rel_l2(gen, rgen))
    assert rel_l2(gen, rgen) < TOL_REL_L2
    assert float((gen - rgen).abs().max()) < TOL_MAXABS_OF_RANGE * float(rgen.max() - rgen.min())


def test_c5_small_ppg_depth18_sample_vc_vs_oracle():
    """BASELINE C5: the shipped configs/F5TTS_Small_PPG.yaml at FULL depth (18 blocks, dim 768 = 48 channels per conv
    group, 12 heads, pe_attn_head 1, PPG input, Gumbel codebook keys in the state dict), sample_vc with the reference
    eval script's weights (alpha_spk 2.5, alpha_ppg 3; eval_infer_batch_vc.py:57-58), NFE 32, N_ref 188 / N 469."""
    import yaml

    import f5e_tts_amd
    from f5e_tts_amd.model import CFM, DiT
    from f5e_tts_amd.train.parse_cfg import parse_model_yaml
    path = os.path.join(os.path.dirname(os.path.abspath(f5e_tts_amd.__file__)), "configs", "F5TTS_Small_PPG.yaml")
    mc = parse_model_yaml(yaml.safe_load(open(path)))
    arch = dict(mc["arch"], text_num_embeds=2545, mel_dim=100)
    assert arch["depth"] == 18 and arch["dim"] == 768
    cfg = O.DiTConfig(dim=768, depth=18, heads=arch["heads"], ff_mult=arch["ff_mult"], text_dim=arch["text_dim"],
                      conv_layers=arch["conv_layers"], text_num_embeds=2545,
                      text_mask_padding=arch.get("text_mask_padding", True), pe_attn_head=arch.get("pe_attn_head"),
                      use_ppg=True, ppg_dim=mc["transformer_ppg_config"]["ppg_dim"])
    sd = SY.init_dit_state(cfg, 4321)
    dit = DiT(**arch, ppg_config=mc["transformer_ppg_config"], cb_config=mc["transformer_codebook_config"])
    missing, unexpected = dit.load_state_dict(sd, strict=False)
    assert not unexpected and all(k.startswith("quantizer.") for k in missing), (missing, unexpected)
    cfm = CFM(transformer=dit, ppg_config=mc["cfm_ppg_config"], cb_config=mc["cfm_codebook_config"]).cuda().eval()
    n_ref, n = 188, 469
    wav = SY.synthetic_ref_wave(n_ref)
    ppg = torch.randn(1, round(0.533 * n), 256, generator=torch.Generator().manual_seed(55))
    kw = dict(duration=n, steps=32, sway_sampling_coef=-1.0, seed=0)
    out, traj = cfm.sample_vc(wav.cuda(), ppg.cuda(), alpha_spk=2.5, alpha_ppg=3.0, **kw)
    ref_out, ref_traj = O.cfm_sample(sd, cfg, wav, None, ppg, mode="vc", alpha_a=2.5, alpha_b=3.0, **kw)
    assert out.shape == ref_out.shape == (1, n, 100)
    check_trajectory(traj, ref_traj, 0, "C5")
    gen, rgen = out[:, n_ref:].cpu(), ref_out[:, n_ref:]
    print("C5: final rel L2 %.3e" % rel_l2(gen, rgen))
    assert rel_l2(gen, rgen) < TOL_REL_L2
    assert float((gen - rgen).abs().max()) < TOL_MAXABS_OF_RANGE * float(rgen.max() - rgen.min())


def _ascii_text(nbytes, seed):
    g = np.random.default_rng(seed)
    words = []
    while sum(len(w) + 1 for w in words) < nbytes:
        words.append("".join(g.choice(list("abcdefghijklmnopqrstuvwxyz"), size=int(g.integers(2, 9)))))
    return " ".join(words)[:nbytes - 1].rstrip() + "."


def test_c4_eval_infer_batch_main_writes_the_oracles_audio(tmp_path):
    """BASELINE C4's caller: eval_infer_batch.main() end to end on one GPU -- model yaml, EMA safetensors checkpoint, local
    Vocos, a .lst test set with the reference's six columns, prompt wavs on disk -- for three utterances whose lengths
    (the shortest, the median and the longest of tests/golden/c4_durations.csv: ~610 / 910 / 1390 total frames) span the
    C4 mix, euler NFE 16, CFG 2, sway -1."""
    import yaml
    from safetensors.torch import save_file

    import f5e_tts_amd
    from f5e_tts_amd.eval import eval_infer_batch as E
    from f5e_tts_amd.infer import utils_infer as U
    from f5e_tts_amd.model.utils import get_tokenizer
    cfg, sd, dit, cfm = full_model(seed=99)
    vs, voc = vocoder()
    save_file({"ema_model." + k: v.contiguous() for k, v in cfm.state_dict().items()}, str(tmp_path / "model.safetensors"))
    vdir = tmp_path / "vocos"
    vdir.mkdir()
    (vdir / "config.yaml").write_text(yaml.safe_dump({
        "backbone": {"init_args": dict(input_channels=100, dim=512, intermediate_dim=1536, num_layers=8)},
        "head": {"init_args": dict(dim=512, n_fft=1024, hop_length=256, padding="center")}}))
    torch.save(voc.state_dict(), str(vdir / "pytorch_model.bin"))
    # the shortest, the median and the longest row of the C4 table
    rows = [tuple(float(x) for x in line.split(",")) for line in
            open(os.path.join(os.path.dirname(__file__), "golden", "c4_durations.csv")) if line[0] != "#" and line.strip()]
    totals = [E.c4_work_list(os.path.join(os.path.dirname(__file__), "golden", "c4_durations.csv"), len(rows))[i]
              for i in range(len(rows))]
    order = sorted(range(len(rows)), key=lambda i: totals[i][1])
    picks = [order[0], order[len(order) // 2], order[-1]]
    audio = tmp_path / "wavs"
    audio.mkdir()
    lst, utts = [], []
    for j, i in enumerate(picks):
        ref_s, ref_b, gen_s, gen_b = rows[i]
        nw = int(ref_s * 24000)
        w = SY.synthetic_ref_wave(nw // 256 + 2, seed=300 + j)[0, :nw] * 3.0     # rms 0.3 > target: no rescale branch
        U.save_wav(str(audio / f"ref{j}.wav"), w.numpy(), 24000)
        ref_txt, gen_txt = _ascii_text(int(ref_b), 10 + j), _ascii_text(int(gen_b), 20 + j)
        lst.append("\t".join([f"ref{j}", f"{ref_s}", ref_txt, f"gen{j}", f"{gen_s}", gen_txt]))
        utts.append((f"gen{j}", str(audio / f"ref{j}.wav"), ref_txt, gen_txt))
    (tmp_path / "test.lst").write_text("\n".join(lst) + "\n")
    cfg_yaml = os.path.join(os.path.dirname(os.path.abspath(f5e_tts_amd.__file__)), "configs", "F5TTS_v1_Base.yaml")
    out_dir = tmp_path / "out"
    argv = ["-n", "F5TTS_v1_Base", "-t", str(tmp_path / "test.lst"), "-nfe", "16", "-s", "0", "--ckpt",
            str(tmp_path / "model.safetensors"), "--audio_root", str(audio), "--vocoder_path", str(vdir), "-mc", cfg_yaml]
    E.main(argv + ["--output_dir", str(out_dir), "--workers", "3"])   # three host threads, one stream each (RankWorkers)
    # in-rank concurrency changes nothing: the files of the one-utterance-at-a-time loop are byte-identical
    out_seq = tmp_path / "out_seq"
    E.main(argv + ["--output_dir", str(out_seq), "--workers", "1"])
    for utt, *_ in utts:
        assert (out_dir / f"{utt}.wav").read_bytes() == (out_seq / f"{utt}.wav").read_bytes(), utt
    vocab, _ = get_tokenizer(U._DEFAULT_VOCAB)
    seen = []
    for j, (utt, path, ref_txt, gen_txt) in enumerate(utts):
        got, sr = U.load_wav(str(out_dir / f"{utt}.wav"))
        assert sr == 24000 and got.shape[0] == 1
        a, _ = U.load_wav(path)
        ref_len = a.shape[-1] // 256
        if len(ref_txt[-1].encode("utf-8")) == 1:
            ref_txt = ref_txt + " "
        gen_txt = " " + gen_txt          # reference eval/utils_eval.py:57 (LibriSpeech-PC list reader)
        tot = E.total_mel_len(ref_len, ref_txt, gen_txt)
        seen.append(tot)
        assert got.shape[1] == 256 * (tot - ref_len - 1)
        # (1) the driver adds nothing: same audio as a direct sample + decode on the same inputs (int16 file round trip)
        chars = U.convert_char_to_pinyin([ref_txt + gen_txt])
        ids = O.list_str_to_idx(chars, vocab)
        mel_in = cfm.mel_spec(a.cuda()).permute(0, 2, 1)[:, :ref_len]
        kw = dict(duration=torch.tensor([tot]), lens=torch.tensor([ref_len]), steps=16, cfg_strength=2.0,
                  sway_sampling_coef=-1.0, seed=0)
        mel, _ = cfm.sample(mel_in, ids, **kw)
        direct = voc.decode(mel[:, ref_len:tot].permute(0, 2, 1)).cpu()
        peak = float(direct.abs().max())
        assert float((got - direct.clamp(-1, 1)).abs().max()) <= 2.0 / 32768, utt   # x32767 on write, /32768 on read
        # (2) the shortest utterance against the fp32 oracle: mel, then the oracle's Vocos on the oracle's mel
        if j == 0:
            ref_mel_in = O.log_mel_spectrogram(a).permute(0, 2, 1)[:, :ref_len]
            ro, _ = O.cfm_sample(sd, cfg, ref_mel_in, ids, None, **kw)
            e = rel_l2(mel[:, ref_len:tot], ro[:, ref_len:tot])
            print("C4 %s: N=%d mel rel L2 %.3e" % (utt, tot, e))
            assert e < TOL_REL_L2
            rw = O.vocos_decode(vs, ro[:, ref_len:tot].permute(0, 2, 1)).clamp(-1, 1)   # the file clips at full scale
            ew = float((got - rw).norm() / rw.norm())
            print("C4 %s: written wav vs oracle wave rel L2 %.3e (peak %.3f)" % (utt, ew, peak))
            assert ew < 5e-2
    print("C4 total frames:", seen)
    assert 500 < seen[0] < 750 and 800 < seen[1] < 1050 and 1250 < seen[2] <= 1500


def test_workspace_arena_matches_the_library_planner():
    """engine.make_plan carves the arena exactly as f5e_workspace_bytes lays it out (SURVEY 8b lower side)."""
    from f5e_tts_amd import _C
    cfg = O.DiTConfig(dim=1024, depth=2, heads=16, ff_mult=2, text_dim=256, conv_layers=1, text_num_embeds=300)
    from f5e_tts_amd.model import DiT
    dit = DiT(dim=1024, depth=2, heads=16, ff_mult=2, text_dim=256, conv_layers=1, text_num_embeds=300)
    dit.load_state_dict(SY.init_dit_state(cfg, 3), strict=True)
    eng = dit.cuda().engine()
    S, B, N = 2, 1, 100
    y = torch.zeros(B, N, 100, device="cuda")
    in_const = torch.zeros(S * N, 1024, device="cuda")
    mod = eng.time_tables(torch.tensor([0.5], device="cuda"))
    cd = eng.cd_tables(mod)
    plan = eng.make_plan(S, B, N, y, in_const, mod, None, eng.rope_table(N), None, cd=cd)
    w, arena = plan.layout, plan.ws["arena"]
    assert arena.numel() == w.total and arena.data_ptr() % 256 == 0
    for i, name in enumerate(_C.WS_NAMES):
        ptr = getattr(plan.c, name)
        if w.bytes[i]:
            assert ptr == arena.data_ptr() + w.offset[i], name
        else:
            assert not ptr, name
    assert plan.c.fuse_ln == 1 and w.bytes[_C.WS_NAMES.index("ln_stats")] == S * N * 16 * 2 * 4


def test_two_threads_on_two_streams_share_cold_tables():
    """the (mod, cd) table cache is filled on the first caller's stream; a second thread on ANOTHER current
    stream that hits the cache must wait for that build (event in the cache entry) -- cold cache, per-thread streams."""
    import threading
    cfg = O.DiTConfig(dim=1024, depth=2, heads=16, ff_mult=2, text_dim=256, conv_layers=2, text_num_embeds=300)
    from f5e_tts_amd.model import CFM, DiT
    sd = SY.init_dit_state(cfg, 1234)
    dit = DiT(dim=1024, depth=2, heads=16, ff_mult=2, text_dim=256, conv_layers=2, text_num_embeds=300)
    dit.load_state_dict(sd, strict=True)
    cfm = CFM(transformer=dit).cuda().eval()
    g = torch.Generator().manual_seed(5)
    cond = torch.randn(1, 30, 100, generator=g).cuda()
    text = torch.randint(0, 300, (1, 9), generator=g)
    kw = dict(duration=90, cfg_strength=2.0, sway_sampling_coef=-1.0, seed=1)
    want = {st: cfm.sample(cond, text, steps=st, **kw)[0].clone() for st in (5, 6, 7)}
    torch.cuda.synchronize()
    for round_ in range(3):
        dit.engine()._tables.clear()                      # cold table cache, warm everything else
        steps = 5 + round_
        outs, streams = [None, None], [torch.cuda.Stream(), torch.cuda.Stream()]
        bar = threading.Barrier(2)

        def work(i):
            with torch.cuda.stream(streams[i]):
                bar.wait()
                outs[i] = cfm.sample(cond, text, steps=steps, **kw)[0]
                streams[i].synchronize()

        ts = [threading.Thread(target=work, args=(i,)) for i in range(2)]
        [t.start() for t in ts]
        [t.join() for t in ts]
        assert torch.equal(outs[0], want[steps]) and torch.equal(outs[1], want[steps]), round_
