#!/usr/bin/env python3
"""Per-OP (QKV / ATTN / OUT / FF1 / FF2) summaries from rocprofv3 outputs of a bench.py run, stamped with the hash of the
kernel sources (tools/src_hash.py) so that bench.py can carry them in its line beside the live HIP-event brackets.

  prof_ops.py trace <dir with *_kernel_trace.csv> <workload C2|C3> <out.json>
      per op: launches, average / median kernel duration (rocprofv3's begin -> end timestamps), TFLOP/s and fraction of
      the 2.5 PFLOP/s dense bf16 peak on the op's ALGORITHMIC flop.
  prof_ops.py pmc <dir with *_counter_collection.csv> <workload> <out.json>
      per op, from ONE eager pass with SQ_VALU_MFMA_BUSY_CYCLES, SQ_WAVE_CYCLES, SQ_WAIT_ANY (SQ block) and GRBM_GUI_ACTIVE:
      mfma_busy = SQ_VALU_MFMA_BUSY_CYCLES / (4 SIMDs x 256 CUs x GRBM_GUI_ACTIVE / 8)   (busy cycles are summed over
      the SIMDs, GRBM_GUI_ACTIVE over the 8 XCDs; MI355X_MICROARCH.md, DVFS note), eff_clock_ghz = GRBM_GUI_ACTIVE / 8 /
      kernel wall time (reads high on dispatches shorter than ~0.3 ms, same note: derive() below nulls it when it exceeds
      the chip's 2.4 GHz and adds mfma_busy_span_min), wave_cycles_parked = SQ_WAIT_ANY / SQ_WAVE_CYCLES.

The two gate+residual GEMMs of a block (out-projection, FF2) run the SAME kernel instantiation, so ops are told apart by
launch order inside a block: ... QKV, attention, OUT, FF1, FF2 ... -- a gate+residual launch behind attention is OUT,
behind the GELU GEMM it is FF2."""
import csv
import glob
import json
import os
import re
import sys
from collections import defaultdict

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tools.src_hash import csrc_sha256  # noqa: E402

PEAK = 2500.0
SHAPES = {"C2": dict(M=938, S=2, N=469), "C3": dict(M=60032, S=64, N=938)}


def flops(op, w):
    s = SHAPES[w]
    if op == "ATTN":
        return 4.0 * s["N"] * s["N"] * 64 * 16 * s["S"]
    return {"QKV": 2.0 * s["M"] * 1024 * 3072, "OUT": 2.0 * s["M"] * 1024 * 1024, "FF1": 2.0 * s["M"] * 1024 * 2048,
            "FF2": 2.0 * s["M"] * 2048 * 1024}[op]


def kclass(name):
    """EPI class of a block kernel by name, or None."""
    if "attn_fwd" in name:
        return "ATTN"
    m = re.search(r"gemm_bf16_kernel<\d+, \d+, (\d)", name) or re.search(r"gemm_bf16_pp_kernel<(\d)", name)
    if not m:
        return None
    return {"3": "QKV", "1": "FF1", "2": "GATE"}.get(m.group(1))


def classify(rows, name_key):
    """rows in dispatch order -> op per row (None for everything that is not one of the five block ops)."""
    out, last = [], None
    for r in rows:
        c = kclass(r[name_key])
        if c == "GATE":
            c = "OUT" if last == "ATTN" else ("FF2" if last == "FF1" else None)
        if c is not None:
            last = c
        out.append(c)
    return out


def med(v):
    s = sorted(v)
    return s[len(s) // 2]


MAX_CLOCK_GHZ = 2.4


def derive(row, c, us):
    """Ratios of one op from its averaged counters and its kernel span (us).  GRBM_GUI_ACTIVE covers the dispatch's whole
    counter window (command processor, launch, drain), not only begin -> end of the kernel: on a 10 us launch the window is
    1.4-1.7 x the span, so mfma_busy (busy / window) reads LOW and eff_clock_ghz (window cycles / span) reads above the
    2.4 GHz the chip can clock.  Such rows get eff_clock_ghz = null and mfma_busy_span_min = busy cycles / (1024 SIMDs x
    span x 2.4 GHz): the matrix pipe's share of the kernel's own span if the clock was at its maximum (a lower clock means
    a higher share)."""
    if c.get("GRBM_GUI_ACTIVE"):
        row["mfma_busy"] = round(c.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0) / (1024.0 * c["GRBM_GUI_ACTIVE"] / 8.0), 4)
        clk = c["GRBM_GUI_ACTIVE"] / 8.0 / (us * 1e3)
        row["eff_clock_ghz"] = round(clk, 3)
        if clk > MAX_CLOCK_GHZ * 1.02:
            row["eff_clock_ghz"] = None
            row["counter_window_over_span"] = round(clk / MAX_CLOCK_GHZ, 2)
            row["mfma_busy_span_min"] = round(c.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0) / (1024.0 * us * 1e3 * MAX_CLOCK_GHZ), 4)
    if c.get("SQ_WAVE_CYCLES"):
        row["wave_cycles_parked"] = round(c.get("SQ_WAIT_ANY", 0.0) / c["SQ_WAVE_CYCLES"], 4)


def main():
    if sys.argv[1] == "redigest":   # prof_ops.py redigest <ops_pmc json>: re-derive the ratios from the stored counters
        res = json.load(open(sys.argv[2]))
        for row in res["ops"].values():
            for k in ("mfma_busy", "eff_clock_ghz", "wave_cycles_parked", "counter_window_over_span", "mfma_busy_span_min"):
                row.pop(k, None)
            derive(row, row["counters"], row["avg_us_under_pmc"])
        note = ("; short dispatches: eff_clock_ghz null where the counter window exceeds the kernel span (window / span given), "
                "mfma_busy_span_min = busy cycles / (1024 SIMDs x span x 2.4 GHz)")
        if note not in res.get("source", ""):
            res["source"] = res.get("source", "") + note
        json.dump(res, open(sys.argv[2], "w"), indent=1)
        return
    mode, d, w, outp = sys.argv[1:5]
    res = {"workload": w, "csrc_sha256": csrc_sha256(), "peak_tflops": PEAK, "ops": {}}
    if mode == "trace":
        f = glob.glob(d + "/**/*_kernel_trace.csv", recursive=True)[0]
        rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
        ops = classify(rows, "Kernel_Name")
        by = defaultdict(list)
        for r, o in zip(rows, ops):
            if o:
                by[o].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) * 1e-3)
        for o, v in by.items():
            avg = sum(v) / len(v)
            res["ops"][o] = {"launches": len(v), "avg_us": round(avg, 3), "median_us": round(med(v), 3),
                             "tflops": round(flops(o, w) / avg / 1e6, 1), "frac": round(flops(o, w) / avg / 1e6 / PEAK, 4)}
        res["source"] = "rocprofv3 --kernel-trace: kernel begin -> end per dispatch, ops told apart by launch order"
    else:
        f = glob.glob(d + "/**/*_counter_collection.csv", recursive=True)[0]
        disp = {}
        for r in csv.DictReader(open(f)):
            k = int(r["Dispatch_Id"])
            e = disp.setdefault(k, {"Kernel_Name": r["Kernel_Name"], "t0": int(r["Start_Timestamp"]), "t1": int(r["End_Timestamp"]), "c": {}})
            e["c"][r["Counter_Name"]] = e["c"].get(r["Counter_Name"], 0.0) + float(r["Counter_Value"])
        rows = [disp[k] for k in sorted(disp)]
        ops = classify(rows, "Kernel_Name")
        by = defaultdict(list)
        for r, o in zip(rows, ops):
            if o:
                by[o].append(r)
        for o, v in by.items():
            n = len(v)
            c = {k: sum(r["c"].get(k, 0.0) for r in v) / n for k in v[0]["c"]}
            us = sum(r["t1"] - r["t0"] for r in v) / n * 1e-3
            row = {"launches": n, "avg_us_under_pmc": round(us, 3), "counters": {k: round(x, 1) for k, x in c.items()}}
            derive(row, c, us)
            res["ops"][o] = row
        res["source"] = ("rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY GRBM_GUI_ACTIVE, one eager pass; "
                         "mfma_busy = MFMA busy cycles / (1024 SIMDs x GRBM_GUI_ACTIVE / 8)"
                         "; short dispatches: eff_clock_ghz null where the counter window exceeds the kernel span (window / span given), "
                         "mfma_busy_span_min = busy cycles / (1024 SIMDs x span x 2.4 GHz)")
    json.dump(res, open(outp, "w"), indent=1)
    for o in ("QKV", "ATTN", "OUT", "FF1", "FF2"):
        if o in res["ops"]:
            print(o, res["ops"][o])


if __name__ == "__main__":
    main()
