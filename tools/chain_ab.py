#!/usr/bin/env python3
"""A/B in one process: one batched forward vs parallel CFG-branch chains (C2 workload)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from f5e_tts_amd.model import CFM, DiT
from tools import synth as SY
cfg = SY.DiTConfig(); sd = SY.init_dit_state(cfg, 1234)
dit = DiT(dim=1024, depth=22, heads=16, ff_mult=2, text_dim=512, conv_layers=4, text_num_embeds=2545)
dit.load_state_dict(sd); cfm = CFM(transformer=dit).cuda().eval()
wav = SY.synthetic_ref_wave(188).cuda(); text = SY.synthetic_text_ids(469).cuda()
outs = {}
for rnd in range(3):
    for ch in (1, None):
        cfm.chains = ch
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(3):
            mel, traj = cfm.sample(wav, text, duration=469, steps=32, cfg_strength=2.0, sway_sampling_coef=-1.0, seed=0)
        torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 3
        outs[ch] = traj[-1].clone()
        print(f"round {rnd} chains={ch}: {dt*1e3:.2f} ms/pass", flush=True)
print("max abs diff between modes:", float((outs[1] - outs[None]).abs().max()))
