#!/usr/bin/env python3
"""How long is a C2 pass with ONE branch (M = 469 rows per launch) vs the CFG-batched two (M = 938)?"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from tools import synth as SY
from f5e_tts_amd.model import CFM, DiT
cfg = SY.DiTConfig(); sd = SY.init_dit_state(cfg, 1234)
dit = DiT(dim=1024, depth=22, heads=16, ff_mult=2, text_dim=512, conv_layers=4, text_num_embeds=2545)
dit.load_state_dict(sd); cfm = CFM(transformer=dit).cuda().eval()
wav = SY.synthetic_ref_wave(188).cuda(); text = SY.synthetic_text_ids(469)
for cfgs in (2.0, 0.0, 2.0, 0.0):
    kw = dict(duration=469, steps=32, cfg_strength=cfgs, sway_sampling_coef=-1.0, seed=0)
    for _ in range(2): cfm.sample(wav, text, **kw)
    torch.cuda.synchronize(); ts = []
    for _ in range(6):
        t0 = time.perf_counter(); cfm.sample(wav, text, **kw); torch.cuda.synchronize(); ts.append((time.perf_counter() - t0) * 1e3)
    print(f"cfg_strength {cfgs}: min {min(ts):.2f} ms median {sorted(ts)[3]:.2f}", flush=True)
