cd $GRAFT_REPO_ROOT
timeout -k 10 1500 python -m pytest tests -x -q -m gpu > gpurun_out/r04zz_tests.log 2>&1 || { tail -40 gpurun_out/r04zz_tests.log; exit 1; }
tail -3 gpurun_out/r04zz_tests.log
timeout -k 10 260 python tools/fuzz_ops.py 200 64 > gpurun_out/r04zz_fuzz_ops.log 2>&1; tail -3 gpurun_out/r04zz_fuzz_ops.log
timeout -k 10 200 python tools/fuzz_model.py 120 65 > gpurun_out/r04zz_fuzz_model.log 2>&1; tail -2 gpurun_out/r04zz_fuzz_model.log
timeout -k 10 200 python tools/fuzz_sampler.py 120 61 > gpurun_out/r04zz_fuzz_sampler.log 2>&1; tail -2 gpurun_out/r04zz_fuzz_sampler.log
timeout -k 10 200 python tools/fuzz_sampler_ppg.py 100 62 > gpurun_out/r04zz_fuzz_sampler_ppg.log 2>&1; tail -2 gpurun_out/r04zz_fuzz_sampler_ppg.log
timeout -k 10 160 python tools/fuzz_frontend.py 80 63 > gpurun_out/r04zz_fuzz_frontend.log 2>&1; tail -2 gpurun_out/r04zz_fuzz_frontend.log
python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/r04zz_smoke.log 2>&1; tail -1 gpurun_out/r04zz_smoke.log
