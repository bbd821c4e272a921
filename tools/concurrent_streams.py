#!/usr/bin/env python3
"""Throughput of T host threads sampling C2 utterances concurrently on one GPU (one capture stream per thread), against a
single thread.  GPU box only."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from concurrent.futures import ThreadPoolExecutor
import torch
from tools import synth as SY
from f5e_tts_amd.model import CFM, DiT
from f5e_tts_amd.vocoder import Vocos

cfg = SY.DiTConfig(); sd = SY.init_dit_state(cfg, 1234)
dit = DiT(dim=1024, depth=22, heads=16, ff_mult=2, text_dim=512, conv_layers=4, text_num_embeds=2545)
dit.load_state_dict(sd); cfm = CFM(transformer=dit).cuda().eval()
voc = Vocos(); voc.load_state_dict(SY.init_vocos_state(), strict=False); voc = voc.cuda().eval()
wav = SY.synthetic_ref_wave(188).cuda(); text = SY.synthetic_text_ids(469)

def one(_):
    mel, _t = cfm.sample(wav, text, duration=469, steps=32, cfg_strength=2.0, sway_sampling_coef=-1.0, seed=0)
    return voc.decode(mel[:, 188:].permute(0, 2, 1))

ref = one(0).clone(); one(0); torch.cuda.synchronize()
for T in (1, 2, 3, 4):
    n = 12 * T
    with ThreadPoolExecutor(max_workers=T) as ex:
        list(ex.map(one, range(T)))  # per-thread stream warm-up
        torch.cuda.synchronize(); t0 = time.perf_counter()
        outs = list(ex.map(one, range(n)))
        torch.cuda.synchronize(); dt = time.perf_counter() - t0
    ok = all(torch.equal(o, ref) for o in outs)
    print(f"threads {T}: {n} passes in {dt*1e3:.1f} ms = {dt/n*1e3:.2f} ms/pass, {469*n/dt:.0f} mel-frames/s, bit-identical {ok}", flush=True)
