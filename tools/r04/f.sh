cd $GRAFT_REPO_ROOT
timeout -k 10 600 python -m pytest tests/test_ops_gpu.py -x -q -m gpu > gpurun_out/r04y_ops.log 2>&1 || { tail -40 gpurun_out/r04y_ops.log; exit 1; }
tail -2 gpurun_out/r04y_ops.log
timeout -k 10 420 python tools/fuzz_ops.py 300 51 > gpurun_out/r04y_fuzz_ops.log 2>&1; tail -3 gpurun_out/r04y_fuzz_ops.log
timeout -k 10 300 python tools/fuzz_model.py 200 52 > gpurun_out/r04y_fuzz_model.log 2>&1; tail -3 gpurun_out/r04y_fuzz_model.log
