cd $GRAFT_REPO_ROOT
timeout -k 10 1500 python -m pytest tests -x -q -m gpu > gpurun_out/r04u_tests.log 2>&1 || { tail -40 gpurun_out/r04u_tests.log; exit 1; }
tail -3 gpurun_out/r04u_tests.log
