#!/usr/bin/env python3
"""Host-side profile of one C2 CFM.sample call (GPU box): where the non-loop milliseconds go."""
import os, sys, time, cProfile, pstats
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from f5e_tts_amd.model import CFM, DiT
from tools import synth as SY
cfg = SY.DiTConfig(); sd = SY.init_dit_state(cfg, 1234)
dit = DiT(dim=1024, depth=22, heads=16, ff_mult=2, text_dim=512, conv_layers=4, text_num_embeds=2545)
dit.load_state_dict(sd); cfm = CFM(transformer=dit).cuda().eval()
wav = SY.synthetic_ref_wave(188).cuda(); text = SY.synthetic_text_ids(469).cuda()
kw = dict(duration=469, steps=32, cfg_strength=2.0, sway_sampling_coef=-1.0, seed=0)
for _ in range(3): cfm.sample(wav, text, **kw)
torch.cuda.synchronize()
pr = cProfile.Profile(); pr.enable()
for _ in range(5): cfm.sample(wav, text, **kw)
torch.cuda.synchronize(); pr.disable()
st = pstats.Stats(pr); st.sort_stats("cumulative").print_stats(45)
