#!/usr/bin/env python3
"""Per-kernel HBM traffic from two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE), corrected as
/opt/skills/guides/MI355X_MICROARCH.md section HBM prescribes for gfx950: the counters are in KiB; FETCH_SIZE reports
exactly half of the bytes of wide (16 B/lane) coalesced streaming reads -> doubled; WRITE_SIZE is exact.
The summary is stamped with the hash of the kernel sources it was measured on (tools/src_hash.py).
Usage: pmc_traffic.py <fetch_dir> <write_dir> <out.json>"""
import csv, glob, json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tools.src_hash import csrc_sha256
from collections import defaultdict

def load(d, name):
    f = glob.glob(d + "/**/*_counter_collection.csv", recursive=True)[0]
    acc = defaultdict(lambda: [0.0, 0])
    for r in csv.DictReader(open(f)):
        if r["Counter_Name"] != name:
            continue
        k = r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "")
        acc[k][0] += float(r["Counter_Value"]); acc[k][1] += 1
    return acc

fetch, write = load(sys.argv[1], "FETCH_SIZE"), load(sys.argv[2], "WRITE_SIZE")
out = {}
for k in sorted(set(fetch) | set(write), key=lambda k: -(fetch.get(k, [0, 1])[0])):
    fk = fetch[k][0] / max(fetch[k][1], 1) if k in fetch else 0.0
    wk = write[k][0] / max(write[k][1], 1) if k in write else 0.0
    out[k] = {"launches": fetch[k][1] if k in fetch else write[k][1], "fetch_size_kib_avg": round(fk, 2),
              "write_size_kib_avg": round(wk, 2), "hbm_bytes_per_launch": int((2.0 * fk + wk) * 1024)}
json.dump({"correction": "bytes = (2 * FETCH_SIZE + WRITE_SIZE) * 1024 (gfx950: FETCH_SIZE halves wide coalesced reads)",
           "csrc_sha256": csrc_sha256(), "kernels": out}, open(sys.argv[3], "w"), indent=1)
for k, v in list(out.items())[:12]:
    print(f"{k[:58]:58s} n={v['launches']:5d} fetch={v['fetch_size_kib_avg']:10.1f} KiB write={v['write_size_kib_avg']:9.1f} KiB -> {v['hbm_bytes_per_launch'] / 1e6:8.2f} MB/launch")
