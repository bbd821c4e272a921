#!/usr/bin/env python3
"""A/B of the ping-pong GEMM launch modes (env switches read per launch) on the four DiT shapes, alternating."""
import sys, os, math
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from f5e_tts_amd import ops
BF = torch.bfloat16
M = int(sys.argv[1]) if len(sys.argv) > 1 else 60032
rps = 938
S = M // rps
def timeit(fn, reps=16):
    for i in range(4): fn(i)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for i in range(reps): fn(i)
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3
modes = {"dynamic": {"F5E_PP_PERSIST": "0"}, "persist": {"F5E_PP_PERSIST": "1", "F5E_PP_STAGGER": "0"},
         "persist+stagger": {"F5E_PP_PERSIST": "1", "F5E_PP_STAGGER": "1"},
         "stagger-all": {"F5E_PP_PERSIST": "1", "F5E_PP_STAGGER": "2"}}
for name, N, K in (("QKV", 3072, 1024), ("OUT", 1024, 1024), ("FF1", 2048, 1024), ("FF2", 1024, 2048)):
    a = torch.randn(M, K, device="cuda").to(BF)
    ws = [(torch.randn(N, K, device="cuda") / math.sqrt(K)).to(BF) for _ in range(6)]
    b = torch.randn(N, device="cuda")
    if name == "QKV":
        npad = (rps + 63) // 64 * 64
        q = torch.zeros(S, 16, npad, 64, device="cuda", dtype=BF); k = torch.zeros_like(q); vt = torch.zeros_like(q)
        cs = torch.zeros(rps, 32, 2, device="cuda")
        fn = lambda i: ops.gemm_bf16_qkv_rope(a, ws[i % 6], b, q, k, vt, 16, 16, cs, rps, tile_hint=9)
    elif name == "FF1":
        out = torch.empty(M, N, device="cuda", dtype=BF)
        fn = lambda i: ops.gemm_bf16_bias(a, ws[i % 6], b, out, act=ops.ACT_GELU_TANH, tile_hint=9)
    else:
        x = torch.zeros(M, N, device="cuda"); gate = torch.randn(1, N, device="cuda")
        fn = lambda i: ops.gemm_bf16_gate_residual(a, ws[i % 6], b, x, gate, rps, tile_hint=9)
    res = {m: [] for m in modes}
    for rep in range(3):
        for m, env in modes.items():
            os.environ.update(env)
            res[m].append(timeit(fn))
    fl = 2.0 * M * N * K
    print(name, " ".join(f"{m}: {min(v):.1f}us({fl / min(v) / 1e6:.0f}TF)" for m, v in res.items()), flush=True)
