#!/usr/bin/env python3
"""Times f5e_flash_attn at the C3 launch size (64 sequences x 16 heads x 938 frames) and at the C2 size (2 x 16 x 469),
random data with q carrying log2(e) / 8 as the QKV epilogue writes it.  GPU box only.
    python tools/attn_time.py [reps]            (F5E_HIP_LIB / F5E_ATTN_VARIANT select tools-build variants)"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from f5e_tts_amd import ops  # noqa: E402

BF = torch.bfloat16
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 50
WAVES = int(os.environ.get("WAVES", "0"))   # KV splits per 32-query tile (0 = auto, -1 = the LDS-shared kernel)
for S, H, N in ((64, 16, 938), (2, 16, 469)):
    npad = (N + 63) // 64 * 64
    g = torch.Generator(device="cuda").manual_seed(1)
    q = (torch.randn(S, H, npad, 64, device="cuda", generator=g) * 0.18).to(BF)
    k = torch.randn(S, H, npad, 64, device="cuda", generator=g).to(BF)
    v = torch.randn(S, H, npad, 64, device="cuda", generator=g).to(BF)
    if os.environ.get("ZERO"):      # all-zero operands: the clock the chip holds without data toggling (DVFS check)
        q.zero_(); k.zero_(); v.zero_()
    ao = torch.empty(S * N, H * 64, device="cuda", dtype=BF)
    for _ in range(5):
        ops.flash_attn(q, k, v, ao, N, waves=WAVES)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        ops.flash_attn(q, k, v, ao, N, waves=WAVES)
    e1.record()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / reps * 1e3
    fl = 4.0 * N * N * 64 * H * S
    print(f"variant {os.environ.get('F5E_ATTN_VARIANT', '-')} waves {WAVES}: S={S} H={H} N={N}: {us:.1f} us per launch, {fl / us / 1e6:.0f} TFLOP/s", flush=True)
