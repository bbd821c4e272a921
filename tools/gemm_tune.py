#!/usr/bin/env python3
"""Times every (tile, stages) variant of the bf16 GEMM kernels on the DiT block shapes, rotating through enough
distinct weight buffers that weights stream from HBM as in the real loop.  GPU box only."""
import sys, os, math
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from f5e_tts_amd import ops

BF = torch.bfloat16
M = int(sys.argv[1]) if len(sys.argv) > 1 else 938
rps = M // 2
NW = int(os.environ.get("NW", "48"))
def timeit(fn, reps=NW * 2):
    for i in range(NW): fn(i)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for i in range(reps): fn(i)
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3

hints = [int(h) for h in sys.argv[2].split(',')] if len(sys.argv) > 2 else [1, 2, 3, 9]  # tile family (f5e_abi.h)
ONLY = os.environ.get("ONLY", "QKV,OUT,FF1,FF2").split(",")
for name, N, K in (("QKV", 3072, 1024), ("OUT", 1024, 1024), ("FF1", 2048, 1024), ("FF2", 1024, 2048)):
    if name not in ONLY:
        continue
    a = torch.randn(M, K, device="cuda").to(BF)
    ws = [(torch.randn(N, K, device="cuda") / math.sqrt(K)).to(BF) for _ in range(NW)]
    b = torch.randn(N, device="cuda")
    res = {}
    for h in hints:
        if name == "QKV":
            npad = (rps + 63) // 64 * 64
            q = torch.zeros(2, 16, npad, 64, device="cuda", dtype=BF); k = torch.zeros_like(q)
            vt = torch.zeros(2, 16, 64, npad, device="cuda", dtype=BF)
            cs = torch.zeros(rps, 32, 2, device="cuda")
            fn = lambda i: ops.gemm_bf16_qkv_rope(a, ws[i % NW], b, q, k, vt, 16, 16, cs, rps, tile_hint=h)
        elif name == "FF1":
            out = torch.empty(M, N, device="cuda", dtype=BF)
            fn = lambda i: ops.gemm_bf16_bias(a, ws[i % NW], b, out, act=ops.ACT_GELU_TANH, tile_hint=h)
        else:
            x = torch.zeros(M, N, device="cuda"); gate = torch.randn(1, N, device="cuda")
            fn = lambda i: ops.gemm_bf16_gate_residual(a, ws[i % NW], b, x, gate, rps, tile_hint=h)
        res[h] = timeit(fn)
    fl = 2.0 * M * N * K
    print(name, f"M={M} N={N} K={K}", " ".join(f"{h}:{t:.1f}us({fl / t / 1e6:.0f}TF)" for h, t in res.items()), flush=True)
