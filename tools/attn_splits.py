#!/usr/bin/env python3
"""Yardstick: us per launch of f5e_flash_attn for S = 2 sequences x 16 heads at C4's lengths, by number of KV splits
(waves per workgroup: 1, 2, 4; 0 = the library's own pick), 22 launches per graph on rotating q / k / v buffers.
GPU box only:  python tools/attn_splits.py"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

from f5e_tts_amd import ops  # noqa: E402


def main():
    cases = [(2, 16, n) for n in (469, 640, 800, 938, 1024, 1100, 1390)]
    if len(sys.argv) > 1:   # "S,H,N S,H,N ..."
        cases = [tuple(int(x) for x in a.split(",")) for a in sys.argv[1:]]
    for S, H, N in cases:
        n_pad = (N + 63) // 64 * 64
        bufs = [[torch.randn(S, H, n_pad, 64, device="cuda").to(torch.bfloat16) for _ in range(3)] for _ in range(6)]
        out = torch.empty(S * N, H * 64, device="cuda", dtype=torch.bfloat16)
        row = []
        for waves in (0, 1, 2, 4):
            for q, k, v in bufs[:2]:
                ops.flash_attn(q, k, v.view(S, H, 64, n_pad), out, N, waves=waves)
            torch.cuda.synchronize()
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g):
                for i in range(22):
                    q, k, v = bufs[i % 6]
                    ops.flash_attn(q, k, v.view(S, H, 64, n_pad), out, N, waves=waves)
            g.replay()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(20):
                g.replay()
            torch.cuda.synchronize()
            row.append((time.perf_counter() - t0) / (20 * 22) * 1e6)
        print(f"S={S} H={H} N={N:5d} grid={(N + 31) // 32 * H * S:5d}  auto {row[0]:6.2f}  1 split {row[1]:6.2f}  2 splits {row[2]:6.2f}  4 splits {row[3]:6.2f}  us per launch", flush=True)


if __name__ == "__main__":
    sys.exit(main())
