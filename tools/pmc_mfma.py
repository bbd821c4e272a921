#!/usr/bin/env python3
"""Summarises a rocprofv3 --pmc counter_collection.csv for kernels matching a substring: per kernel the averaged
counters and the derived MFMA-pipe utilisation = SQ_VALU_MFMA_BUSY_CYCLES / (4 SIMDs x 256 CUs x GRBM_GUI_ACTIVE / 8)
(GRBM_GUI_ACTIVE is summed over the 8 XCDs; microarch guide, DVFS note).  usage: pmc_mfma.py <dir> <substr> [out.json]"""
import csv, glob, json, sys
from collections import defaultdict
d, sub = sys.argv[1], sys.argv[2]
f = glob.glob(d + "/**/*_counter_collection.csv", recursive=True)[0]
acc = defaultdict(lambda: defaultdict(list))
for r in csv.DictReader(open(f)):
    k = r["Kernel_Name"]
    if sub not in k:
        continue
    k = k.replace("(anonymous namespace)::", "").replace("void ", "")[:60]
    acc[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
    acc[k]["_ns"].append(float(r["End_Timestamp"]) - float(r["Start_Timestamp"]))
out = {}
for k, c in acc.items():
    row = {n: sum(v) / len(v) for n, v in c.items()}
    row["launches"] = len(c["_ns"]) // max(1, len(c) - 1)
    if "SQ_VALU_MFMA_BUSY_CYCLES" in row and "GRBM_GUI_ACTIVE" in row:
        row["mfma_pipe_util"] = row["SQ_VALU_MFMA_BUSY_CYCLES"] / (1024.0 * row["GRBM_GUI_ACTIVE"] / 8.0)
    if "SQ_WAIT_ANY" in row and "SQ_WAVE_CYCLES" in row:
        row["wave_cycles_parked_frac"] = row["SQ_WAIT_ANY"] / row["SQ_WAVE_CYCLES"]
    out[k] = row
    print(k, {n: (round(v, 4) if v < 10 else int(v)) for n, v in row.items()})
if len(sys.argv) > 3:
    json.dump(out, open(sys.argv[3], "w"), indent=1)
