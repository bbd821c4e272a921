#!/bin/bash
# A/B bench driver for gpurun: tools/gpu_ab.sh <tag> "<env A>" "<env B>" [extra bench args]
# prints one summary line per arm; the JSON lines land in gpurun_out/<tag>_{A,B}.json
tag=$1; envA=$2; envB=$3; shift 3
for arm in A B; do
  if [ $arm = A ]; then e=$envA; else e=$envB; fi
  env $e timeout -k 10 400 python bench.py --no-cpu-baseline "$@" > gpurun_out/${tag}_${arm}.json 2> gpurun_out/${tag}_${arm}.err || { echo "$arm failed"; tail -5 gpurun_out/${tag}_${arm}.err; exit 1; }
done
python - "$tag" <<'PY'
import json, sys
for t in "AB":
    d = json.load(open(f"gpurun_out/{sys.argv[1]}_{t}.json"))
    r3 = d.get("roofline_c3") or {}
    print(t, d["value"], d["ms_per_step"], (d.get("roofline") or {}).get("all_ops_us"), r3.get("mel_frames_per_sec"),
          {k: v["us"] for k, v in (r3.get("ops") or {}).items()}, r3.get("gemm_flop_weighted_frac"),
          (d.get("concurrent") or {}).get("value"))
PY
