#!/usr/bin/env python3
"""Random sweep of the three-branch samplers of the PPG model (`sample_tts`, `sample_vc`, `sample` with PPG) against the
oracle.  GPU box only.

    python tools/fuzz_sampler_ppg.py [seconds] [seed]"""
import os
import random
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch  # noqa: E402

import test_e2e_gpu as E  # noqa: E402
from oracle import f5e_oracle as O  # noqa: E402

budget = float(sys.argv[1]) if len(sys.argv) > 1 else 120.0
rng = random.Random(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
torch.set_num_threads(16)
cfg = O.DiTConfig(**dict(E.SMALL, use_ppg=True, ppg_dim=256, text_mask_padding=False, pe_attn_head=1))
sd, dit, cfm = E.build(cfg)
t0, n, bad, worst = time.time(), 0, [], 0.0
while time.time() - t0 < budget:
    B = rng.randint(1, 3)
    nc, nt, npg = rng.randint(5, 100), rng.randint(1, 120), rng.randint(1, 260)
    g = torch.Generator().manual_seed(rng.randint(0, 10 ** 6))
    cond = torch.randn(B, nc, 100, generator=g)
    text = torch.randint(0, 300, (B, nt), generator=g)
    for b in range(B):
        if rng.random() < 0.5:
            text[b, rng.randint(1, nt):] = -1
    ppg = torch.randn(B, npg, 256, generator=g)
    duration = rng.randint(2, 240) if rng.random() < 0.5 else torch.tensor([rng.randint(2, 240) for _ in range(B)])
    kw = dict(duration=duration, steps=rng.randint(1, 4), sway_sampling_coef=rng.choice([None, -1.0]), seed=rng.randint(0, 999))
    if rng.random() < 0.4:
        kw["lens"] = torch.tensor([rng.randint(1, nc) for _ in range(B)])
    mode = rng.choice(["tts", "vc", "cfg"])
    a_spk, a_b = rng.choice([1.0, 2.5]), rng.choice([0.5, 3.0])
    cfm.use_graph = rng.random() < 0.6
    case = dict(mode=mode, B=B, nc=nc, nt=nt, npg=npg, graph=cfm.use_graph,
                **{k: (v.tolist() if isinstance(v, torch.Tensor) else v) for k, v in kw.items()})
    n += 1
    try:
        dkw = {k: (v.cuda() if isinstance(v, torch.Tensor) else v) for k, v in kw.items()}
        if mode == "tts":
            ro, rt = O.cfm_sample(sd, cfg, cond, text, None, mode="tts", alpha_a=a_spk, alpha_b=a_b, **kw)
            o, t = cfm.sample_tts(cond.cuda(), text.cuda(), alpha_spk=a_spk, alpha_txt=a_b, **dkw)
        elif mode == "vc":
            ro, rt = O.cfm_sample(sd, cfg, cond, None, ppg, mode="vc", alpha_a=a_spk, alpha_b=a_b, **kw)
            o, t = cfm.sample_vc(cond.cuda(), ppg.cuda(), alpha_spk=a_spk, alpha_ppg=a_b, **dkw)
        else:
            ro, rt = O.cfm_sample(sd, cfg, cond, text, ppg, cfg_strength=2.0, **kw)
            o, t = cfm.sample(cond.cuda(), text.cuda(), ppg.cuda(), cfg_strength=2.0, **dkw)
        assert o.shape == ro.shape and t.shape == rt.shape, (o.shape, ro.shape)
        assert torch.equal(t[0].cpu(), rt[0]), "seeded noise differs"
        e1 = E.rel_l2(t[-1], rt[-1])
        worst = max(worst, e1)
        assert e1 < 2e-2 and bool(torch.isfinite(o).all()), e1
    except Exception as e:  # noqa: BLE001
        bad.append(case)
        print("FAIL", case, repr(e).splitlines()[0][:300], flush=True)
print(f"{n} three-branch sampler calls in {time.time() - t0:.0f} s, {len(bad)} failed, worst rel-L2 {worst:.2e}")
sys.exit(1 if bad else 0)
