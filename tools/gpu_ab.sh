#!/bin/bash
# Multi-arm bench driver for gpurun:  tools/gpu_ab.sh <tag> [--args "<bench args>"] "<env arm 0>" "<env arm 1>" ...
# Runs bench.py --no-cpu-baseline once per arm (same box, back to back), JSON lines to gpurun_out/<tag>_<i>.json,
# and prints one summary line per arm.  An empty string is a valid arm (defaults).
tag=$1; shift
bargs=""
if [ "$1" = "--args" ]; then bargs=$2; shift 2; fi
i=0
for e in "$@"; do
  env $e timeout -k 10 400 python bench.py --no-cpu-baseline $bargs > gpurun_out/${tag}_${i}.json 2> gpurun_out/${tag}_${i}.err || { echo "arm $i failed"; tail -5 gpurun_out/${tag}_${i}.err; exit 1; }
  i=$((i+1))
done
python - "$tag" "$@" <<'PY'
import json, sys
tag, arms = sys.argv[1], sys.argv[2:]
for i, e in enumerate(arms):
    d = json.load(open(f"gpurun_out/{tag}_{i}.json"))
    r3 = d.get("roofline_c3") or {}
    print(f"[{i}] {e or '(defaults)'}\n    C2 {d['value']} mel-frames/s {d['ms_per_step']} ms  ops {(d.get('roofline') or {}).get('all_ops_us')}  conc {(d.get('concurrent') or {}).get('value')}"
          f"\n    C3 {r3.get('mel_frames_per_sec')}  ops { {k: v['us'] for k, v in (r3.get('ops') or {}).items()} }  gemm frac {r3.get('gemm_flop_weighted_frac')}")
PY
