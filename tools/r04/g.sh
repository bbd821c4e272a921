cd $GRAFT_REPO_ROOT
timeout -k 10 900 python -m pytest tests/test_ops_gpu.py tests/test_e2e_gpu.py -x -q -m gpu > gpurun_out/r04z_tests.log 2>&1 || { tail -30 gpurun_out/r04z_tests.log; exit 1; }
tail -2 gpurun_out/r04z_tests.log
timeout -k 10 900 python -m pytest tests/test_baseline_configs_gpu.py -x -q -m gpu -k "c3" > gpurun_out/r04z_c3.log 2>&1 || { tail -30 gpurun_out/r04z_c3.log; exit 1; }
tail -2 gpurun_out/r04z_c3.log
T="F5E_HIP_LIB=$GRAFT_REPO_ROOT/f5e-tts_amd/libf5e_hip_tools.so"
bash tools/gpu_ab.sh r04z --args "--c4-total 0 --streams 0 --steps 5" "$T F5E_PP_TAIL=0" "$T" "$T F5E_PP_TAIL=0" "$T"
