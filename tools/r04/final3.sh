cd $GRAFT_REPO_ROOT
timeout -k 10 600 python bench.py --workload C4 --no-cpu-baseline --steps 200 > gpurun_out/r04zz_bench_C4.json 2> gpurun_out/r04zz_bench_C4.err || { tail -5 gpurun_out/r04zz_bench_C4.err; exit 1; }
python -c "import json; d=json.loads(open('gpurun_out/r04zz_bench_C4.json').read().strip().splitlines()[-1]); print('C4', d['value'], d['ms_per_step'], d['steps'])"
timeout -k 10 400 python bench.py --no-cpu-baseline --no-c3 --c4-total 0 --steps 20 > gpurun_out/r04zz_bench_c2_second_box.json 2>/dev/null; python -c "import json; d=json.loads(open('gpurun_out/r04zz_bench_c2_second_box.json').read().strip().splitlines()[-1]); print('C2', d['value'], d['ms_per_step'], d['roofline']['all_ops_us'], d['concurrent']['value'])"
