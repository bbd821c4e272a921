cd $GRAFT_REPO_ROOT
timeout -k 10 400 python -m pytest tests/test_ops_gpu.py tests/test_e2e_gpu.py -x -q -m gpu 2>&1 | tail -2 || exit 1
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
timeout -k 10 300 python bench.py --no-cpu-baseline --no-c3 --c4-total 0 --steps 10 > gpurun_out/r04bo.json 2>/dev/null; python -c "
import json; d=json.loads(open('gpurun_out/r04bo.json').read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'], d['roofline']['rocprof'] and d['roofline']['rocprof']['block_sum_us'], d['roofline']['traffic'])"
