cd $GRAFT_REPO_ROOT
T=tools/gemm_trace.bin
for v in 10 74; do timeout -k 5 60 $T $v 938 3072 1024 3 || exit 1; done > gpurun_out/r04x_trace_qkv.txt 2>&1
grep -h "variant\|per workgroup\|per launch\|consumer epi" gpurun_out/r04x_trace_qkv.txt
timeout -k 10 600 python -m pytest tests/test_ops_gpu.py tests/test_e2e_gpu.py -x -q -m gpu > gpurun_out/r04x_tests.log 2>&1 || { tail -30 gpurun_out/r04x_tests.log; exit 1; }
tail -2 gpurun_out/r04x_tests.log
python bench.py --no-cpu-baseline --no-c3 --c4-total 0 --streams 0 --steps 20 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'], d['roofline']['all_ops_us'])"
