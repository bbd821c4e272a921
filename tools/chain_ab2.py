#!/usr/bin/env python3
"""C2 passes with the CFG branches batched in one forward (chains=1) vs as two parallel chains of the captured graph."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from tools import synth as SY
from f5e_tts_amd.model import CFM, DiT
cfg = SY.DiTConfig(); sd = SY.init_dit_state(cfg, 1234)
dit = DiT(dim=1024, depth=22, heads=16, ff_mult=2, text_dim=512, conv_layers=4, text_num_embeds=2545)
dit.load_state_dict(sd); cfm = CFM(transformer=dit).cuda().eval()
wav = SY.synthetic_ref_wave(188).cuda(); text = SY.synthetic_text_ids(469)
kw = dict(duration=469, steps=32, cfg_strength=2.0, sway_sampling_coef=-1.0, seed=0)
ref = None
for rep in range(3):
    for ch in (1, 2):
        cfm.chains = ch
        for _ in range(2): out = cfm.sample(wav, text, **kw)[0]
        torch.cuda.synchronize(); ts = []
        for _ in range(6):
            t0 = time.perf_counter(); out = cfm.sample(wav, text, **kw)[0]; torch.cuda.synchronize()
            ts.append((time.perf_counter() - t0) * 1e3)
        if ref is None: ref = out.clone()
        print(f"chains={ch}: min {min(ts):.2f} ms  median {sorted(ts)[3]:.2f}  max {max(ts):.2f}  identical {torch.equal(out, ref)}", flush=True)
