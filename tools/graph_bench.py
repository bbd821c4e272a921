#!/usr/bin/env python3
"""True per-launch cost of each hot kernel: N launches captured in a hipGraph (rotating weight sets), replayed."""
import os, sys, time, math
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from f5e_tts_amd import ops
BF = torch.bfloat16
S = int(sys.argv[1]) if len(sys.argv) > 1 else 2
N = int(sys.argv[2]) if len(sys.argv) > 2 else 469
M, D, FF, H = S * N, 1024, 2048, 16
NWS, ROUNDS = int(os.environ.get("NWS", "22")), int(os.environ.get("ROUNDS", "5"))
npad = (N + 63) // 64 * 64
dev = "cuda"
x = torch.randn(M, D, device=dev); hn = torch.empty(M, D, device=dev, dtype=BF)
mod = torch.randn(1, 6 * D, device=dev)
q = torch.zeros(S, H, npad, 64, device=dev, dtype=BF); k = torch.zeros_like(q); v = torch.zeros_like(q)
cs = torch.zeros(N, 32, 2, device=dev); ao = torch.empty(M, D, device=dev, dtype=BF); ffb = torch.empty(M, FF, device=dev, dtype=BF)
def W(n, kk): return [(torch.randn(n, kk, device=dev) / math.sqrt(kk)).to(BF) for _ in range(NWS)]
wq, wo, w1, w2 = W(3 * D, D), W(D, D), W(FF, D), W(D, FF)
bq, bo, b1 = torch.randn(3 * D, device=dev), torch.randn(D, device=dev), torch.randn(FF, device=dev)
hn.copy_(torch.randn(M, D, device=dev)); ao.copy_(hn); ffb.copy_(torch.randn(M, FF, device=dev))
hint = int(os.environ.get("HINT", "0"))
cases = {
    "ln": lambda i: ops.layernorm(x, hn, scale=mod[:, D:2 * D], shift=mod[:, :D], rows_per_seq=N),
    "qkv": lambda i: ops.gemm_bf16_qkv_rope(hn, wq[i % NWS], bq, q, k, v, H, H, cs, N, tile_hint=hint),
    "attn": lambda i: ops.flash_attn(q, k, v, ao, N),
    "out": lambda i: ops.gemm_bf16_gate_residual(ao, wo[i % NWS], bo, x, mod[:, 2 * D:3 * D], N, tile_hint=hint),
    "ff1": lambda i: ops.gemm_bf16_bias(hn, w1[i % NWS], b1, ffb, act=ops.ACT_GELU_TANH, tile_hint=hint),
    "ff2": lambda i: ops.gemm_bf16_gate_residual(ffb, w2[i % NWS], bo, x, mod[:, 5 * D:6 * D], N, tile_hint=hint),
}
st = torch.cuda.Stream()
with torch.cuda.stream(st):
    for name, fn in cases.items():
        fn(0); torch.cuda.synchronize()
        g = ops.Graph(); g.begin()
        for i in range(NWS * ROUNDS): fn(i)
        g.end()
        for _ in range(3): g.launch()
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(10): g.launch()
        torch.cuda.synchronize(); us = (time.perf_counter() - t0) / 10 / (NWS * ROUNDS) * 1e6
        print(f"{name:5s} S={S} N={N}: {us:7.2f} us per launch (in-graph, dependent)", flush=True)
