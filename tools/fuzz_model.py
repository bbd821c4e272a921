#!/usr/bin/env python3
"""Random-configuration sweep of one full DiT forward (HIP path) against the fp32 oracle.  GPU box only.

    python tools/fuzz_model.py [seconds] [seed]

Draws width / depth / heads-with-RoPE / qk-norm / long skip / batch / length / ragged masks / text lengths / per-item times /
drop flags, builds the model through the reference-named classes, and checks `DiT.sample` against `oracle.dit_sample` with the
tolerance of tests/test_e2e_gpu.py (relative L2 1e-2 per two blocks of bf16 contractions, on the valid rows).  Large row
counts (the 256 x 256 GEMM and the LDS attention kernel) are drawn now and then with a narrow model so that the CPU oracle stays
within seconds."""
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
seed0 = int(sys.argv[2]) if len(sys.argv) > 2 else 0
rng = random.Random(seed0)
torch.set_num_threads(16)
t0, n, bad = time.time(), 0, []
worst, n_big, n_masked = (0.0, None), 0, 0
while time.time() - t0 < budget:
    dim = rng.choice([256, 512, 768, 1024])
    heads = dim // 64
    big = rng.random() < 0.2
    depth = 1 if big else rng.randint(1, 2)
    kw = dict(dim=dim, depth=depth, heads=heads, ff_mult=rng.choice([2, 4]) if dim <= 512 else 2, text_dim=rng.choice([256, 512]),
              conv_layers=rng.randint(0, 2), text_num_embeds=300, pe_attn_head=rng.choice([None, 1, heads]) if dim > 256 else None,
              qk_norm=rng.choice([None, None, "rms_norm"]), long_skip_connection=rng.random() < 0.2)
    if big:
        kw.update(dim=256, heads=4, ff_mult=2, pe_attn_head=None)
        B, N = rng.randint(12, 20), rng.randint(700, 1000)
    else:
        B, N = rng.randint(1, 6), rng.randint(8, 400)
    masked = B > 1 and rng.random() < 0.6
    case = dict(kw, B=B, N=N, masked=masked)
    n += 1
    try:
        cfg = O.DiTConfig(**kw)
        sd, dit, _ = E.build(cfg, seed=rng.randint(0, 10 ** 6))
        g = torch.Generator().manual_seed(rng.randint(0, 10 ** 6))
        x, cond = torch.randn(B, N, 100, generator=g), torch.randn(B, N, 100, generator=g)
        nt = rng.randint(1, min(N, 60))
        text = torch.randint(0, 300, (B, nt), generator=g)
        for b in range(1, B):
            cut = rng.randint(1, nt)
            text[b, cut:] = -1
        lens = torch.tensor([N] + [rng.randint(max(1, N // 3), N) for _ in range(B - 1)])
        mask = (torch.arange(N)[None] < lens[:, None]) if masked else None
        tm = torch.rand(B, generator=g) if (B > 1 and rng.random() < 0.5) else torch.tensor(float(rng.random()))
        drops = [rng.random() < 0.3 for _ in range(3)]
        ref = O.dit_sample(sd, cfg, x, cond, text, None, tm, drops[0], drops[1], drops[2], mask)
        out = dit.sample(x.cuda(), cond.cuda(), text.cuda(), None, tm.cuda(), drops[0], drops[1], drops[2],
                         mask.cuda() if masked else None).cpu()
        valid = (mask if masked else torch.ones(B, N, dtype=torch.bool))[..., None].expand_as(ref)
        err = float((out[valid] - ref[valid]).norm() / ref[valid].norm())
        mx = float((out - ref)[valid].abs().max()) / float(ref[valid].abs().max())
        case["rel_l2"], case["max_rel"] = round(err, 5), round(mx, 5)
        if not (err < 1e-2 and mx < 0.04) or not bool(torch.isfinite(out[valid]).all()):
            raise AssertionError(f"rel_l2 {err:.4g} max {mx:.4g}")
        del dit
        n_big += big
        n_masked += masked
        if err > worst[0]:
            worst = (err, case)
    except Exception as e:  # noqa: BLE001
        bad.append(case)
        print("FAIL", case, repr(e).splitlines()[0][:300], flush=True)
print(f"{n} configurations in {time.time() - t0:.0f} s ({n_big} at >= 8400 rows, {n_masked} ragged), {len(bad)} failed; "
      f"worst rel-L2 {worst[0]:.2e} at {worst[1]}")
sys.exit(1 if bad else 0)
