#!/usr/bin/env python3
"""Random-shape sweep of the op-level parity tests (tests/test_ops_gpu.py bodies, called directly).  GPU box only.

    python tools/fuzz_ops.py [seconds] [seed]

Draws shapes inside each op's documented domain (odd row counts, rows per sequence that are not multiples of anything, heads
that are not multiples of 4, N / K at the edges of the tile sizes) and runs the same comparison against the fp32 reference
the fixed-shape tests run.  Prints every failing argument tuple; exit code 1 if any."""
import os
import random
import sys
import time
import traceback

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch  # noqa: E402

import test_ops_gpu as T  # noqa: E402
from f5e_tts_amd import ops  # noqa: E402

budget = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
rng = random.Random(int(sys.argv[2]) if len(sys.argv) > 2 else 0)


def r(lo, hi):
    return rng.randint(lo, hi)


def c(*xs):
    return rng.choice(xs)


def draw():
    k = rng.random()
    if k < 0.2:
        hint = c(0, 1, 2, 3, 9)
        K = 64 * r(1, 6) if hint < 9 else 64 * r(2, 6)
        return T.test_gemm_bf16_bias, (r(1, 700), 4 * r(1, 200), K, hint)
    if k < 0.45:
        hint = c(0, 9)
        rps = r(1, 400)
        S = r(1, 5)
        K = 64 * r(2, 6) if hint >= 9 else 64 * r(1, 6)
        return T.test_gemm_bf16_gate_residual, (S * rps, 4 * r(1, 160), K, rps, hint)
    if k < 0.7:
        hint = c(0, 9)
        H = r(1, 12)
        return T.test_qkv_rope, (r(1, 4), r(1, 420), H, r(0, H), 64 * r(2, 5), hint)
    if k < 0.85:
        S, H, N = r(1, 4), r(1, 8), r(21, 520)
        return T.test_flash_attn, (S, H, N, c(0, 1, 2, 4, -1), c(False, True))
    if k < 0.93:
        return T.test_gemm_f32, (r(1, 600), r(1, 300) * c(1, 2, 4), 4 * r(1, 160))
    if k < 0.96:
        G = c(1, 2, 4, 16)
        return T.test_convpos, (r(1, 3), r(1, 500), G * c(16, 32, 48, 64), G)
    k2 = rng.random()
    if k2 < 0.2:
        hint = c(0, 9)
        N = r(20, 330) if rng.random() < 0.5 else r(331, 1000)   # up to 3000 rows: the 64- and 128-row role-split tiles, the classic ones
        return T.test_fused_adaln_chain, (r(1, 3), N, c(256, 512, 768, 1024), 4 * r(8, 400), c(0.0, 0.3, 0.7), N >= 30 and c(False, True), hint)
    if k2 < 0.3:
        if rng.random() < 0.5:
            return T.test_fused_qkv_rope_consumer_role_split_and_classic, (r(1, 4), r(1, 700), r(1, 16), c(256, 512, 768, 1024))
        return T.test_fused_gelu_consumer_role_split_and_classic, (r(1, 4), r(1, 700), c(256, 512, 768, 1024), c(4 * r(8, 600), 128 * r(1, 20)), 0)
    if k2 < 0.5:
        return T.test_grn, (r(1, 3), r(1, 400), c(64, 66, 192, 512, 1024))
    if k2 < 0.7:
        return T.test_stft_logmel, (r(1, 3), r(3, 200))     # fewer frames: the wave is shorter than the reflect padding (refused)
    if k2 < 0.9:
        return T.test_istft_head, (r(1, 2), r(2, 300))      # torch.istft (the comparison) refuses a single frame
    return T.test_layernorm_variants, (c(256, 512, 768, 1024, 1280, 2048),)


t0, n, bad = time.time(), 0, []
while time.time() - t0 < budget:
    fn, args = draw()
    if fn is T.test_convpos and args[2] % args[3]:
        continue
    n += 1
    try:
        fn(ops, *args)
        torch.cuda.synchronize()
    except Exception as e:  # noqa: BLE001
        bad.append((fn.__name__, args, repr(e).splitlines()[0][:200]))
        print("FAIL", fn.__name__, args, repr(e).splitlines()[0][:300], flush=True)
        if "F5EError" not in repr(e) and "AssertionError" not in repr(e):
            traceback.print_exc()
print(f"{n} cases in {time.time() - t0:.0f} s, {len(bad)} failed")
sys.exit(1 if bad else 0)
