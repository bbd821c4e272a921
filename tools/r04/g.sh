cd $GRAFT_REPO_ROOT
SKIP_MFMA=1 bash tools/gpu_profile.sh r04af > gpurun_out/r04af_profile.log 2>&1 || { tail -20 gpurun_out/r04af_profile.log; exit 1; }
grep -E "^(QKV|ATTN|OUT|FF1|FF2) " gpurun_out/r04af_profile.log
timeout -k 10 900 python bench.py > gpurun_out/r04af_bench_c2_default.json 2> gpurun_out/r04af_bench_c2_default.err || { tail -20 gpurun_out/r04af_bench_c2_default.err; exit 1; }
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r04af_bench_c2_default.json').read().strip().splitlines()[-1])
print(d['value'], d['ms_per_step'], d.get('scaling_c4'), (d.get('concurrent') or {}).get('value'), d.get('c3',{}).get('value') if isinstance(d.get('c3'),dict) else None)
PY
