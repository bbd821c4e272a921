cd $GRAFT_REPO_ROOT
T="F5E_HIP_LIB=$GRAFT_REPO_ROOT/f5e-tts_amd/libf5e_hip_tools.so"
timeout -k 10 600 python -m pytest tests/test_baseline_configs_gpu.py tests/test_model_gpu.py -x -q -m gpu 2>&1 | tail -2 || exit 1
bash tools/gpu_ab.sh r04bi --args "--workload C4 --steps 100 --streams 0" "$T F5E_PF_SCHEME=0" "" "$T F5E_PF_SCHEME=0" "" | grep "C2 "
bash tools/gpu_ab.sh r04bj --args "--workload C5 --steps 10 --streams 0" "$T F5E_PF_SCHEME=0" "" "$T F5E_PF_SCHEME=0" "" | grep "C2 "
bash tools/gpu_ab.sh r04bk --args "--no-c3 --c4-total 0 --streams 0 --steps 20" "$T F5E_PF_SCHEME=0" "" | grep "C2 "
