#!/usr/bin/env python3
"""Per-workgroup timeline of the 256x256 ping-pong GEMM (tools build: launches stamp while F5E_PP_TRACE names a buffer).  GPU box only.

    make -C f5e-tts_amd/csrc tools-lib && python tools/pp_timeline.py [M] [ff1|out]

Every workgroup stamps s_memrealtime (100 MHz) per tile at: entry, first K-tile landed (vmcnt is in order, so the previous
tile's store acknowledgements are in this interval), K loop done, epilogue issued.  Prints per tile round the mean of the
three intervals and how far apart the CUs reach "K loop done" (lock step or not)."""
import math
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ["F5E_HIP_LIB"] = os.path.join(ROOT, "f5e-tts_amd", "libf5e_hip_tools.so")
import torch  # noqa: E402

from f5e_tts_amd import ops  # noqa: E402

BF = torch.bfloat16
M = int(sys.argv[1]) if len(sys.argv) > 1 else 60032
kind = sys.argv[2] if len(sys.argv) > 2 else "ff1"
HINT = 9
N, K = {"ff1": (2048, 1024), "qkv": (3072, 1024), "ff2": (1024, 2048)}.get(kind, (1024, 1024))
a = torch.randn(M, K, device="cuda").to(BF)
ws = [(torch.randn(N, K, device="cuda") / math.sqrt(K)).to(BF) for _ in range(4)]
b = torch.randn(N, device="cuda")
trace = torch.zeros(256 * 16 * 6, device="cuda", dtype=torch.int64)
if kind == "ff1":
    out = torch.empty(M, N, device="cuda", dtype=BF)
    fn = lambda i, h: ops.gemm_bf16_bias(a, ws[i % 4], b, out, act=ops.ACT_GELU_TANH, tile_hint=h)   # noqa: E731
elif kind == "qkv":
    S, H, R = 64, 16, M // 64
    n_pad = (R + 63) // 64 * 64
    cs = torch.empty(R, 32, 2, device="cuda")
    ops.rope_table((1.0 / (10000 ** (torch.arange(0, 64, 2).float() / 64))).cuda(), cs)
    q = torch.zeros(S, H, n_pad, 64, device="cuda", dtype=BF)
    k, vt = torch.zeros_like(q), torch.zeros_like(q)
    b = torch.randn(N, device="cuda")
    fn = lambda i, h: ops.gemm_bf16_qkv_rope(a, ws[i % 4], b, q, k, vt, H, int(os.environ.get('ROPE_HEADS', '1')), cs, R, tile_hint=h)   # noqa: E731
else:
    resid = torch.randn(M, N, device="cuda")
    gate = torch.randn(1, N, device="cuda")
    fn = lambda i, h: ops.gemm_bf16_gate_residual(a, ws[i % 4], b, resid, gate, rows_per_seq=M, tile_hint=h)   # noqa: E731

for i in range(4):
    fn(i, 9)
os.environ["F5E_PP_TRACE"] = str(trace.data_ptr())
for i in range(3):
    fn(i, HINT)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
trace.zero_()
e0.record()
fn(3, HINT)
e1.record()
torch.cuda.synchronize()
raw = trace.cpu()
t = raw[:256 * 64].view(256, 16, 4).double() * 0.01   # us
cyc = raw[256 * 64:].view(256, 16, 2).double()
t0 = t[:, 0, 0][t[:, 0, 0] > 0].min()
print(f"{kind} M={M} N={N} K={K}: {e0.elapsed_time(e1) * 1e3:.1f} us")
print("round  n_wg  entry->landed  K loop  epilogue   loop-end spread (p5..p95 of t - t0)")
for r in range(16):
    m = t[:, r, 3] > 0
    if not m.any():
        break
    d = t[m, r]
    le = (d[:, 2] - t0).sort().values
    p = lambda q: float(le[int(q * (len(le) - 1))])   # noqa: E731
    print(f"{r:5d} {int(m.sum()):5d} {float((d[:,1]-d[:,0]).mean()):12.2f} {float((d[:,2]-d[:,1]).mean()):8.2f} "
          f"{float((d[:,3]-d[:,2]).mean()):8.2f}    {p(0.05):7.1f} .. {p(0.5):7.1f} .. {p(0.95):7.1f}")
m = t[:, 1, 3] > 0
mhz = ((cyc[m, 1, 1] - cyc[m, 1, 0]) / (t[m, 1, 2] - t[m, 1, 1])).mean()
print(f"shader clock inside the K loop of round 1: {float(mhz):.0f} MHz (s_memtime ticks / s_memrealtime us)")
ep = (t[:, :, 3] - t[:, :, 2])[t[:, :, 3] > 0].sort().values
print("epilogue over all tiles: " + "  ".join(f"p{q}={float(ep[int(q / 100 * (len(ep) - 1))]):.2f}" for q in (5, 25, 50, 75, 95)) + " us")
last = t[:, :, 3].max() - t0
print(f"last epilogue issued at {float(last):.1f} us after the first entry")
