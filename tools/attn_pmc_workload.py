#!/usr/bin/env python3
"""Three launches of the C3-size attention (64 sequences x 16 heads x 938 frames) for a rocprofv3 --pmc pass."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from f5e_tts_amd import ops
BF = torch.bfloat16
S, H, N = 64, 16, 938
npad = (N + 63) // 64 * 64
q = (torch.randn(S, H, npad, 64, device="cuda") * 0.18).to(BF); k = torch.randn(S, H, npad, 64, device="cuda").to(BF); v = torch.randn_like(k)   # q carries log2(e) / 8 (f5e_abi.h)
ao = torch.empty(S * N, H * 64, device="cuda", dtype=BF)
for _ in range(3):
    ops.flash_attn(q, k, v, ao, N)
torch.cuda.synchronize()
