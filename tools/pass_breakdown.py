#!/usr/bin/env python3
"""Wall-clock breakdown of one C2 pass (GPU box): prep, graph capture+instantiate, replay loop, vocoder."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from f5e_tts_amd import engine as E, ops
from f5e_tts_amd.model import CFM, DiT
from f5e_tts_amd.vocoder import Vocos
from tools import synth as SY

cfg = SY.DiTConfig(); sd = SY.init_dit_state(cfg, 1234)
dit = DiT(dim=1024, depth=22, heads=16, ff_mult=2, text_dim=512, conv_layers=4, text_num_embeds=2545)
dit.load_state_dict(sd); cfm = CFM(transformer=dit).cuda().eval()
voc = Vocos(); voc.load_state_dict(SY.init_vocos_state(), strict=False); voc = voc.cuda().eval()
wav = SY.synthetic_ref_wave(188).cuda(); text = SY.synthetic_text_ids(469).cuda()

def sync(): torch.cuda.synchronize(); return time.perf_counter()
orig_begin, orig_end, orig_launch = ops.Graph.begin, ops.Graph.end, ops.Graph.launch
T = {}
def begin(self): T['cap0'] = sync(); orig_begin(self)
def end(self):
    t = time.perf_counter(); orig_end(self); T['cap1'] = t; T['inst'] = sync()
def launch(self):
    if 'l0' not in T: T['l0'] = sync()
    orig_launch(self)
ops.Graph.begin, ops.Graph.end, ops.Graph.launch = begin, end, launch
for it in range(4):
    T.clear()
    t0 = sync()
    mel, _ = cfm.sample(wav, text, duration=469, steps=32, cfg_strength=2.0, sway_sampling_coef=-1.0, seed=0)
    t1 = sync()
    w = voc.decode(mel[:, 188:].permute(0, 2, 1))
    t2 = sync()
    print(f"pass {it}: total {1e3*(t2-t0):.2f} ms | prep {1e3*(T['cap0']-t0):.2f} | capture {1e3*(T['cap1']-T['cap0']):.2f} | "
          f"instantiate {1e3*(T['inst']-T['cap1']):.2f} | loop+stitch {1e3*(t1-T['l0']):.2f} | vocoder {1e3*(t2-t1):.2f}", flush=True)
