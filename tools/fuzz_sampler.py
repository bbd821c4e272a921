#!/usr/bin/env python3
"""Random sweep of `CFM.sample` keyword paths (HIP path) against `oracle.cfm_sample`.  GPU box only.

    python tools/fuzz_sampler.py [seconds] [seed]

A small DiT (2 blocks) with random batch, reference lengths, per-item durations (scalar or tensor, some shorter than the
reference or than the text: the duration bump), `lens`, `max_duration` clamps, `no_ref_audio`, `edit_mask`, CFG strength 0 /
positive, sway coefficient or none, euler / midpoint, a few steps, graph and eager: shapes must match the oracle's, the seeded
noise must be bit-identical, the final mel within the tolerance of tests/test_e2e_gpu.py::test_sampler_flags_and_edges."""
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
cfg = O.DiTConfig(**E.SMALL)
sd, dit, cfm = E.build(cfg)
t0, n, bad, worst = time.time(), 0, [], 0.0
while time.time() - t0 < budget:
    B = rng.randint(1, 4)
    nc = rng.randint(5, 120)
    g = torch.Generator().manual_seed(rng.randint(0, 10 ** 6))
    cond = torch.randn(B, nc, 100, generator=g)
    nt = rng.randint(1, 150)
    text = torch.randint(0, 300, (B, nt), generator=g)
    for b in range(B):
        if rng.random() < 0.6:
            text[b, rng.randint(1, nt):] = -1
    lens = torch.tensor([rng.randint(1, nc) for _ in range(B)]) if rng.random() < 0.6 else None
    if rng.random() < 0.5:
        duration = rng.randint(2, 260)
    else:
        duration = torch.tensor([rng.randint(2, 260) for _ in range(B)])
    kw = dict(duration=duration, lens=lens, steps=rng.randint(1, 5), cfg_strength=rng.choice([0.0, 1.0, 2.0, 2.5]),
              sway_sampling_coef=rng.choice([None, -1.0, 0.5]), seed=rng.randint(0, 1000), method=rng.choice(["euler", "euler", "midpoint"]))
    if rng.random() < 0.25:
        kw["max_duration"] = rng.randint(40, 200)
    if rng.random() < 0.2:
        kw["no_ref_audio"] = True
    if rng.random() < 0.2:
        ne = int(lens.max()) if lens is not None else nc     # the reference ANDs it with lens_to_mask(lens): [B, max(lens)]
        em = torch.ones(B, ne, dtype=torch.bool)
        a0 = rng.randint(0, ne - 1)
        em[:, a0:rng.randint(a0, ne)] = False
        kw["edit_mask"] = em
    cfm.use_graph = rng.random() < 0.6
    case = {k: (v.tolist() if isinstance(v, torch.Tensor) and v.numel() < 8 else v) for k, v in kw.items() if k != "edit_mask"}
    case.update(B=B, nc=nc, nt=nt, graph=cfm.use_graph, edit="edit_mask" in kw)
    n += 1
    try:
        ro, rt = O.cfm_sample(sd, cfg, cond, text, None, **kw)
        cfm.odeint_kwargs = dict(method=kw["method"])       # the reference takes the method at construction
        o, t = cfm.sample(cond.cuda(), text.cuda(), **{k: (v.cuda() if isinstance(v, torch.Tensor) else v)
                                                       for k, v in kw.items() if k != "method"})
        assert o.shape == ro.shape and t.shape == rt.shape, (o.shape, ro.shape, t.shape, rt.shape)
        assert torch.equal(t[0].cpu(), rt[0]), "seeded noise differs"
        if float(ro.norm()) == 0.0:     # max_duration inside a zeroed reference (no_ref_audio): the output is the reference
            assert float(o.abs().max()) == 0.0
            continue
        e1, e2 = E.rel_l2(t[-1], rt[-1]), E.rel_l2(o, ro)
        worst = max(worst, e1, e2)
        assert e1 < 1.5e-2 and e2 < 1.5e-2 and bool(torch.isfinite(o).all()), (e1, e2)
    except Exception as e:  # noqa: BLE001
        bad.append(case)
        print("FAIL", case, repr(e).splitlines()[0][:300], flush=True)
print(f"{n} sampler calls in {time.time() - t0:.0f} s, {len(bad)} failed, worst rel-L2 {worst:.2e}")
sys.exit(1 if bad else 0)
