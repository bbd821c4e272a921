#!/usr/bin/env python3
"""GB/s of the HBM-bound kernels (LayerNorm / modulate, CFG + Euler, log-mel, iSTFT head) from a rocprofv3
--kernel-trace --stats summary: algorithmic bytes per launch (stated below, per workload) / average launch time, against
the 6.29 TB/s measured copy peak of MI355X_MICROARCH.md (8 TB/s spec).
Usage: hbm_kernels.py <kernel_stats.csv> <C2|C3> [out.json]"""
import csv, json, re, sys

path, wl = sys.argv[1], sys.argv[2]
B, NREF, N = (1, 188, 469) if wl == "C2" else (32, 375, 938)
S, D, MEL = 2 * B, 1024, 100
M = S * N
nw = 256 * (NREF - 1) + 128
T_gen = N - NREF                      # frames decoded by Vocos
spec = [  # (regex on the kernel name, what, algorithmic bytes per launch)
    (r"layernorm_kernel<4>", "LayerNorm + modulate, f32 in / bf16 out [M, 1024]", M * D * (4 + 2)),
    (r"adaln_pre_kernel<4>", "fused-AdaLN head: f32 in, bf16 xs out [M, 1024]", M * D * (4 + 2)),
    (r"ln_finalize_kernel", "row statistics: [M, 16, 2] f32 in, [M, 2] out", M * (16 * 8 + 8)),
    (r"ode_update_kernel", "CFG + Euler: 2 pred branches + y in, y + trajectory row out", B * N * MEL * 4 * 5),
    (r"stft_logmel(_banded)?_kernel", "log-mel: wave in (each sample once), [T, 100] out", B * (nw * 4 + NREF * MEL * 4)),
    (r"istft_frames_kernel", "iSTFT head: [T, 1026] in, [T, 1024] frames out", B * T_gen * (1026 + 1024) * 4),
    (r"istft_ola_kernel", "overlap-add: [T, 1024] frames in, 256 (T - 1) samples out", B * (T_gen * 1024 + 256 * (T_gen - 1)) * 4),
    (r"stitch_kernel", "where(mask, cond, y): 2 in, 1 out [B, N, 100]", B * N * MEL * 4 * 3),
]
rows = list(csv.DictReader(open(path)))
out = {"workload": wl, "peak_measured_GBps": 6290.0, "peak_spec_GBps": 8000.0, "kernels": {}}
for r in rows:
    name = r["Name"].replace("(anonymous namespace)::", "").replace("void ", "")
    for rx, what, nbytes in spec:
        if re.search(rx, name):
            us = float(r["AverageNs"]) / 1e3
            gbps = nbytes / (us * 1e-6) / 1e9
            out["kernels"][name.split("(")[0]] = {"what": what, "calls": int(r["Calls"]), "avg_us": round(us, 2),
                                                  "algorithmic_bytes": int(nbytes), "GBps": round(gbps, 1),
                                                  "frac_of_measured_peak": round(gbps / 6290.0, 4)}
            break
for k, v in out["kernels"].items():
    print(f"{k[:44]:44s} n={v['calls']:6d} {v['avg_us']:9.2f} us  {v['algorithmic_bytes'] / 1e6:9.2f} MB  {v['GBps']:8.1f} GB/s ({v['frac_of_measured_peak'] * 100:.1f} % of 6.29 TB/s)")
if len(sys.argv) > 3:
    json.dump(out, open(sys.argv[3], "w"), indent=1)
