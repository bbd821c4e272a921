#!/usr/bin/env python3
"""FF1-shaped timing of the ping-pong GEMM and its ablation builds (hint 9 + 10*dbg).  GPU box only."""
import sys, os, math
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from f5e_tts_amd import ops
BF = torch.bfloat16
M = int(sys.argv[1]) if len(sys.argv) > 1 else 60032
N, K = int(os.environ.get("N", "2048")), int(os.environ.get("K", "1024"))
a = torch.randn(M, K, device="cuda").to(BF)
ws = [(torch.randn(N, K, device="cuda") / math.sqrt(K)).to(BF) for _ in range(8)]
b = torch.randn(N, device="cuda"); out = torch.empty(M, N, device="cuda", dtype=BF)
for h in [int(x) for x in (sys.argv[2] if len(sys.argv) > 2 else "21,9,19,29,39").split(",")]:
    fn = lambda i: ops.gemm_bf16_bias(a, ws[i % 8], b, out, act=ops.ACT_GELU_TANH, tile_hint=h)
    for i in range(8): fn(i)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for i in range(24): fn(i)
    e1.record(); torch.cuda.synchronize()
    t = e0.elapsed_time(e1) / 24 * 1e3
    print(f"hint {h}: {t:.1f} us ({2.0 * M * N * K / t / 1e6:.0f} TF/s)", flush=True)
