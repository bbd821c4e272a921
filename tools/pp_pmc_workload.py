#!/usr/bin/env python3
"""Three launches of each large-M DiT GEMM (auto tile selection -> gemm_bf16_pp) for a rocprofv3 --pmc pass."""
import sys, os, math
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from f5e_tts_amd import ops
BF = torch.bfloat16
M, rps = 60032, 938
S = M // rps
for name, N, K in (("QKV", 3072, 1024), ("OUT", 1024, 1024), ("FF1", 2048, 1024), ("FF2", 1024, 2048)):
    a = torch.randn(M, K, device="cuda").to(BF)
    w = (torch.randn(N, K, device="cuda") / math.sqrt(K)).to(BF)
    b = torch.randn(N, device="cuda")
    for _ in range(3):
        if name == "QKV":
            npad = (rps + 63) // 64 * 64
            q = torch.zeros(S, 16, npad, 64, device="cuda", dtype=BF); k = torch.zeros_like(q); vt = torch.zeros_like(q)
            cs = torch.zeros(rps, 32, 2, device="cuda")
            ops.gemm_bf16_qkv_rope(a, w, b, q, k, vt, 16, 16, cs, rps)
        elif name == "FF1":
            out = torch.empty(M, N, device="cuda", dtype=BF)
            ops.gemm_bf16_bias(a, w, b, out, act=ops.ACT_GELU_TANH)
        else:
            x = torch.zeros(M, N, device="cuda"); gate = torch.randn(1, N, device="cuda")
            ops.gemm_bf16_gate_residual(a, w, b, x, gate, rps)
    torch.cuda.synchronize()
