#!/bin/bash
# usage: tools/ab_env.sh VAR v1 v2 ... : short bench per value of an environment variable, prints ms/pass + per-class GEMM medians
var=$1; shift
for v in "$@"; do
  env $var=$v timeout -k 10 200 python bench.py --steps 4 --warmup 2 --no-cpu-baseline --streams 0 2>/dev/null > gpurun_out/ab_$v.json
  python - "$var=$v" gpurun_out/ab_$v.json <<'PY'
import sys, json
d = json.loads(open(sys.argv[2]).read().strip().splitlines()[-1])
print(sys.argv[1], d["ms_per_step"], d["roofline"]["all_gemm_avg_us"])
PY
done
