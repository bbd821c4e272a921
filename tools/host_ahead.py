#!/usr/bin/env python3
"""Does the host run ahead of the GPU across back-to-back C2 passes?  Host wall time of every call (no syncs inside the
loop) against the GPU time per pass; plus where the host time of one call goes (sample / decode, graph begin / end /
launches).  If the host time per call is close to the GPU time per pass, the eager sections (vocoder, next pass's
front-end) are launched just in time and the GPU idles between their kernels."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from f5e_tts_amd import ops
from f5e_tts_amd.model import CFM, DiT
from f5e_tts_amd.vocoder import Vocos
from tools import synth as SY

cfg = SY.DiTConfig(); sd = SY.init_dit_state(cfg, 1234)
dit = DiT(dim=1024, depth=22, heads=16, ff_mult=2, text_dim=512, conv_layers=4, text_num_embeds=2545)
dit.load_state_dict(sd); cfm = CFM(transformer=dit).cuda().eval()
voc = Vocos(); voc.load_state_dict(SY.init_vocos_state(), strict=False); voc = voc.cuda().eval()
wav = SY.synthetic_ref_wave(188).cuda(); text = SY.synthetic_text_ids(469)

marks = []
ob, oe, ol = ops.Graph.begin, ops.Graph.end, ops.Graph.launch
def begin(self): marks.append(("begin", time.perf_counter())); ob(self)
def end(self): oe(self); marks.append(("end", time.perf_counter()))
def launch(self): ol(self); marks.append(("launch", time.perf_counter()))
ops.Graph.begin, ops.Graph.end, ops.Graph.launch = begin, end, launch

def one_pass():
    marks.append(("call", time.perf_counter()))
    mel, _ = cfm.sample(wav, text, duration=469, steps=32, cfg_strength=2.0, sway_sampling_coef=-1.0, seed=0)
    marks.append(("sampled", time.perf_counter()))
    out = voc.decode(mel[:, 188:, :].permute(0, 2, 1))
    marks.append(("decoded", time.perf_counter()))
    return out

for _ in range(3):
    one_pass()
torch.cuda.synchronize()
marks.clear()
K = 8
t0 = time.perf_counter()
for _ in range(K):
    one_pass()
t_host = time.perf_counter()
torch.cuda.synchronize()
t1 = time.perf_counter()
print(f"{K} passes: host loop returned after {(t_host - t0) * 1e3:.1f} ms, GPU done after {(t1 - t0) * 1e3:.1f} ms "
      f"({(t1 - t0) / K * 1e3:.2f} ms per pass)")
calls = [i for i, m in enumerate(marks) if m[0] == "call"] + [len(marks)]
for c in range(K):
    seg = marks[calls[c]:calls[c + 1]]
    t = {k: v for k, v in seg if k != "launch"}
    ls = [v for k, v in seg if k == "launch"]
    base = t["call"]
    print(f"  call {c}: at {(base - t0) * 1e3:7.1f} ms | to graph begin {(t['begin'] - base) * 1e3:5.2f} | capture+instantiate "
          f"{(t['end'] - t['begin']) * 1e3:5.2f} | first launch {(ls[0] - t['end']) * 1e3:5.2f} | 32 launches "
          f"{(ls[-1] - ls[0]) * 1e3:6.2f} | rest of sample {(t['sampled'] - ls[-1]) * 1e3:5.2f} | decode {(t['decoded'] - t['sampled']) * 1e3:5.2f} "
          f"| total {(t['decoded'] - base) * 1e3:6.2f} ms")

# constant part of a pass: pipelined time per pass at NFE 32 and 16 -> per-evaluation time and the rest
def timed(nfe, k=6):
    def p():
        mel, _ = cfm.sample(wav, text, duration=469, steps=nfe, cfg_strength=2.0, sway_sampling_coef=-1.0, seed=0)
        return voc.decode(mel[:, 188:, :].permute(0, 2, 1))
    for _ in range(2):
        p()
    torch.cuda.synchronize(); a = time.perf_counter()
    for _ in range(k):
        p()
    torch.cuda.synchronize()
    return (time.perf_counter() - a) / k * 1e3
t32, t16 = timed(32), timed(16)
ev = (t32 - t16) / 16
print(f"pass at NFE 32: {t32:.2f} ms, NFE 16: {t16:.2f} ms -> {ev * 1e3:.1f} us per evaluation, {t32 - 32 * ev:.2f} ms outside the ODE loop")
