#!/usr/bin/env python3
"""Upper bound of what Infinity-Cache-resident block weights would buy the batch-1 chain (C2): the same 22-block model is
timed with its real weights (646 MB cycling through the 256 MB cache: every GEMM streams from HBM) and with the 22 blocks
ALIASED onto the first NSETS weight sets (NSETS x 29.4 MB stay resident).  Results of the aliased run are garbage; only
the time matters.  Usage: mall_probe.py [NSETS=4] [passes=12]"""
import os, sys, threading, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from tools import synth as SY
from f5e_tts_amd.model import CFM, DiT

NSETS = int(sys.argv[1]) if len(sys.argv) > 1 else 4
PASSES = int(sys.argv[2]) if len(sys.argv) > 2 else 12
cfg = SY.DiTConfig()
dit = DiT(dim=1024, depth=22, heads=16, ff_mult=2, text_dim=512, conv_layers=4, text_num_embeds=2545)
dit.load_state_dict(SY.init_dit_state(cfg, 1234), strict=True)
cfm = CFM(transformer=dit).cuda().eval()
wav = SY.synthetic_ref_wave(188).cuda()
text = SY.synthetic_text_ids(469)


def timed(tag):
    for _ in range(3):
        cfm.sample(wav, text, duration=469, steps=32, cfg_strength=2.0, sway_sampling_coef=-1.0, seed=0)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(PASSES):
        cfm.sample(wav, text, duration=469, steps=32, cfg_strength=2.0, sway_sampling_coef=-1.0, seed=0)
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) / PASSES * 1e3
    print(f"{tag}: {ms:.2f} ms per pass ({ms / 32 * 1e3:.0f} us per evaluation)", flush=True)
    return ms


a = timed("22 distinct weight sets (HBM)")
eng = dit.engine()
names = ("w_qkv", "b_qkv", "w_out", "b_out", "w_ff1", "b_ff1", "w_ff2", "b_ff2")
for i in range(eng.L):
    for n in names:
        setattr(eng.block_arr[i], n, getattr(eng.block_arr[i % NSETS], n))
eng._loops = threading.local()     # plans / graphs captured the old pointers
b = timed(f"blocks aliased onto {NSETS} sets ({NSETS * 29.4:.0f} MB, cache resident)")
print(f"upper bound of the MALL benefit: {a - b:.2f} ms per pass = {(a - b) / a * 100:.1f} %")
