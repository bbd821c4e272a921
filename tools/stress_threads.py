#!/usr/bin/env python3
"""Stress of concurrent CFM.sample calls on one model (the reference's ThreadPoolExecutor use): counts
results that differ from the sequential ones.  GPU box only.  env: ROUNDS, F5E_FUSE_LN, USE_GRAPH."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from concurrent.futures import ThreadPoolExecutor
import torch
from tools import synth as SY
from f5e_tts_amd.model import CFM, DiT

KW = dict(dim=1024, depth=2, heads=16, ff_mult=2, text_dim=256, conv_layers=2, text_num_embeds=300)
cfg = SY.DiTConfig(**KW)
sd = SY.init_dit_state(cfg, 1234)
dit = DiT(**KW)
dit.load_state_dict(sd)
cfm = CFM(transformer=dit).cuda().eval()
cfm.use_graph = os.environ.get("USE_GRAPH", "1") != "0"
g = torch.Generator().manual_seed(13)
jobs = []
for i, n in enumerate((70, 101, 83, 64, 90, 77)):
    cond = torch.randn(1, 30, 100, generator=g).cuda()
    text = torch.randint(0, 300, (1, 9 + i), generator=g).cuda()
    jobs.append(dict(cond=cond, text=text, duration=n, steps=4, cfg_strength=2.0, sway_sampling_coef=-1.0, seed=20 + i))
seq = [cfm.sample(**j)[0].clone() for j in jobs]
torch.cuda.synchronize()
bad = 0
rounds = int(os.environ.get("ROUNDS", "40"))
for r in range(rounds):
    w = 2 + r % 3
    with ThreadPoolExecutor(max_workers=w) as ex:
        par = list(ex.map(lambda j: cfm.sample(**j)[0].clone(), jobs))
    torch.cuda.synchronize()
    for k, (a, b) in enumerate(zip(seq, par)):
        if not torch.equal(a, b):
            bad += 1
            d = (a - b).abs()
            print(f"round {r} workers {w} job {k}: max diff {float(d.max()):.3e}, {int((d > 0).sum())}/{d.numel()} elements, "
                  f"rows differing {int((d.amax(-1) > 0).sum())}", flush=True)
print(f"fuse={os.environ.get('F5E_FUSE_LN', '1')} graph={cfm.use_graph}: {bad} mismatches in {rounds} rounds x {len(jobs)} jobs")
