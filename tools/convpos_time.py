"""Time the conv position embedding pair (mode 0 -> mode 1) as a chain of dependent launches replayed from a hipGraph,
the way it runs inside one ODE step.   usage: python tools/convpos_time.py [S N D G]"""
import importlib
import math
import sys

import torch

sys.path.insert(0, ".")
ops = importlib.import_module("f5e-tts_amd.ops")

S, N, D, G = (int(v) for v in sys.argv[1:5]) if len(sys.argv) >= 5 else (2, 469, 1024, 16)
BF = torch.bfloat16
gen = torch.Generator().manual_seed(5)
cpg = D // G
x = torch.randn(S * N, D, generator=gen).to(BF).cuda()
res = torch.randn(S * N, D, generator=gen).cuda()
b = (torch.randn(D, generator=gen) * 0.1).cuda()
NW = int(__import__("os").environ.get("NW", "12"))  # distinct weight sets, so the chain does not run out of one warm L2 image
wps = [ops.pack_convpos_weight((torch.randn(D, cpg, 31, generator=gen) / math.sqrt(cpg * 31)).to(BF).float(), G).cuda()
       for _ in range(NW)]
c1 = torch.empty(S * N, D, device="cuda", dtype=BF)
o32 = torch.empty(S * N, D, device="cuda")
REP = 24
st = torch.cuda.Stream()
with torch.cuda.stream(st):
    for i in range(2):
        ops.convpos(x, wps[0], b, S, N, out_bf16=c1)
        ops.convpos(c1, wps[1], b, S, N, out_f32=o32, resid=res)
    st.synchronize()
    gr = ops.Graph()
    gr.begin()
    for i in range(REP):
        ops.convpos(x, wps[(2 * i) % NW], b, S, N, out_bf16=c1)
        ops.convpos(c1, wps[(2 * i + 1) % NW], b, S, N, out_f32=o32, resid=res)
    gr.end()
    for _ in range(3):
        gr.launch()
    st.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    ts = []
    for _ in range(7):
        e0.record(st)
        gr.launch()
        e1.record(st)
        st.synchronize()
        ts.append(e0.elapsed_time(e1) * 1e3 / (2 * REP))
    ts.sort()
    flops = 2.0 * S * N * D * cpg * 31
    print(f"convpos S={S} N={N} D={D} G={G}: {ts[len(ts) // 2]:.2f} us per launch incl. the 1.56 us launch gap "
          f"(min {ts[0]:.2f}); {flops / ts[len(ts) // 2] * 1e-6:.0f} TFLOP/s")
    gr.destroy()
    if "--trace" in sys.argv:
        import ctypes
        _C = importlib.import_module("f5e-tts_amd._C")
        tiles = (N + 63) // 64
        nwg = G * tiles * S
        buf = torch.zeros(nwg * 8, dtype=torch.int64, device="cuda")
        # the hook exists in the TOOLS build only: make -C f5e-tts_amd/csrc tools-lib; F5E_HIP_LIB=.../libf5e_hip_tools.so
        hook = _C.lib().f5e_debug_convpos_trace
        hook.argtypes, hook.restype = [ctypes.c_void_p], None
        hook(ctypes.c_void_p(buf.data_ptr()))
        for i in range(6):  # last launch is what stays in the buffer
            ops.convpos(x, wps[(2 * i) % NW], b, S, N, out_bf16=c1)
            ops.convpos(c1, wps[(2 * i + 1) % NW], b, S, N, out_f32=o32, resid=res)
        st.synchronize()
        hook(ctypes.c_void_p(0))
        t = buf.view(nwg, 8).cpu().double()
        cyc = (t[:, 6] - t[:, 1])
        real_ns = (t[:, 7] - t[:, 0]) * 10.0
        mhz = (cyc / real_ns * 1e3).median().item()
        med = lambda v: v.median().item()
        print(f"  trace (mode 1 launch, {nwg} workgroups, clock ~{mhz:.0f} MHz) cycles from entry: first tile landed {med(t[:, 2] - t[:, 1]):.0f} | "
              f"mark A {med(t[:, 3] - t[:, 1]):.0f} | mark B {med(t[:, 4] - t[:, 1]):.0f} | loop done {med(t[:, 5] - t[:, 1]):.0f} | "
              f"stores acknowledged {med(t[:, 6] - t[:, 1]):.0f} (= {med(cyc) / mhz:.2f} us); "
              f"start spread {(t[:, 0].max() - t[:, 0].min()).item() * 10:.0f} ns, first start -> last end {(t[:, 7].max() - t[:, 0].min()).item() * 10:.0f} ns")
