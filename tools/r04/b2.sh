cd $GRAFT_REPO_ROOT
( time timeout -k 10 1100 python bench.py > gpurun_out/r04w_bench_default.json 2> gpurun_out/r04w_bench_default.err ) 2> gpurun_out/r04w_time.txt || { tail -20 gpurun_out/r04w_bench_default.err; exit 1; }
cat gpurun_out/r04w_time.txt; tail -25 gpurun_out/r04w_bench_default.err
