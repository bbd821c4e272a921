#!/usr/bin/env python3
"""Socket power and clocks (rocm-smi) while one kernel class runs back to back for a few seconds.  GPU box only.

    python tools/power_probe.py [ff1|out|ff2|qkv|attn|idle] [M] [seconds]

Evidence for DESIGN 4's "the shader clock inside the K loop is 1.4-2.0 GHz": what the power management reports while the
large-M GEMMs / attention run, next to the board's power cap."""
import json
import math
import os
import subprocess
import sys
import threading
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from f5e_tts_amd import ops  # noqa: E402

BF = torch.bfloat16
kind = sys.argv[1] if len(sys.argv) > 1 else "ff1"
M = int(sys.argv[2]) if len(sys.argv) > 2 else 60032
secs = float(sys.argv[3]) if len(sys.argv) > 3 else 3.0


def smi(*args):
    try:
        out = subprocess.run(["rocm-smi", *args, "--json"], capture_output=True, text=True, timeout=20).stdout
        return json.loads(out[out.index("{"):])
    except Exception as e:  # noqa: BLE001
        return {"error": repr(e)}


def make():
    if kind == "idle":
        return lambda i: None
    if kind == "attn":
        S, H, R = 64, 16, M // 64
        n_pad = (R + 63) // 64 * 64
        q = (torch.randn(S, H, n_pad, 64, device="cuda") * 0.18).to(BF)   # q carries log2(e) / 8 (f5e_abi.h)
        k, v = torch.randn_like(q), torch.randn_like(q)
        o = torch.empty(S * R, H * 64, device="cuda", dtype=BF)
        return lambda i: ops.flash_attn(q, k, v, o, R)
    N, K = {"ff1": (2048, 1024), "qkv": (3072, 1024), "ff2": (1024, 2048)}.get(kind, (1024, 1024))
    a = torch.randn(M, K, device="cuda").to(BF)
    ws = [(torch.randn(N, K, device="cuda") / math.sqrt(K)).to(BF) for _ in range(4)]
    b = torch.randn(N, device="cuda")
    if kind == "ff1":
        out = torch.empty(M, N, device="cuda", dtype=BF)
        return lambda i: ops.gemm_bf16_bias(a, ws[i % 4], b, out, act=ops.ACT_GELU_TANH)
    if kind == "qkv":
        S, H, R = 64, 16, M // 64
        n_pad = (R + 63) // 64 * 64
        cs = torch.empty(R, 32, 2, device="cuda")
        ops.rope_table((1.0 / (10000 ** (torch.arange(0, 64, 2).float() / 64))).cuda(), cs)
        q = torch.zeros(S, H, n_pad, 64, device="cuda", dtype=BF)
        k, vt = torch.zeros_like(q), torch.zeros_like(q)
        return lambda i: ops.gemm_bf16_qkv_rope(a, ws[i % 4], b, q, k, vt, H, 1, cs, R)
    resid = torch.randn(M, N, device="cuda")
    gate = torch.randn(1, N, device="cuda") * 0.01
    return lambda i: ops.gemm_bf16_gate_residual(a, ws[i % 4], b, resid, gate, rows_per_seq=M)


fn = make()
stop = False
count = [0]


def runner():
    i = 0
    while not stop:
        for _ in range(50):
            fn(i)
            i += 1
        torch.cuda.synchronize()
        count[0] = i


print("caps:", json.dumps(smi("--showmaxpower")))
th = threading.Thread(target=runner)
t0 = time.time()
th.start()
samples = []
while time.time() - t0 < secs:
    time.sleep(0.4)
    samples.append((round(time.time() - t0, 2), smi("--showpower", "--showclocks")))
stop = True
th.join()
dt = time.time() - t0
for t, sm in samples:
    card = next(iter(sm.values())) if sm and "error" not in sm else sm
    keep = {k: v for k, v in card.items() if any(w in k.lower() for w in ("power", "sclk", "mclk"))} if isinstance(card, dict) else card
    print(f"t={t:5.2f}s {json.dumps(keep)}")
print(f"{kind} M={M}: {count[0]} launches in {dt:.2f} s = {dt / max(count[0], 1) * 1e6:.1f} us per launch (incl. sync gaps)")
