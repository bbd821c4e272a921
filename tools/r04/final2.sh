cd $GRAFT_REPO_ROOT
bash tools/gpu_profile.sh r04zz > gpurun_out/r04zz_profile.log 2>&1 || { tail -20 gpurun_out/r04zz_profile.log; exit 1; }
grep -E "^(QKV|ATTN|OUT|FF1|FF2) " gpurun_out/r04zz_profile.log
for f in ops_trace_c2 ops_trace_c3 ops_pmc_c2 ops_pmc_c3 pmc_traffic_c2; do cp gpurun_out/r04zz_$f.json profiles/$f.json; done
timeout -k 10 900 python bench.py > gpurun_out/r04zz_bench_c2_default.json 2> gpurun_out/r04zz_bench_c2_default.err || { tail -20 gpurun_out/r04zz_bench_c2_default.err; exit 1; }
tail -4 gpurun_out/r04zz_bench_c2_default.err
for w in C3 C4 C5; do timeout -k 10 400 python bench.py --workload $w --no-cpu-baseline --steps 5 > gpurun_out/r04zz_bench_$w.json 2> gpurun_out/r04zz_bench_$w.err || { tail -5 gpurun_out/r04zz_bench_$w.err; exit 1; }; python -c "import json; d=json.loads(open('gpurun_out/r04zz_bench_$w.json').read().strip().splitlines()[-1]); print('$w', d['value'], d['ms_per_step'], (d.get('concurrent') or {}).get('value'))"; done
