cd $GRAFT_REPO_ROOT
timeout -k 10 300 python -m pytest tests/test_ops_gpu.py -x -q -m gpu -k "gemm or qkv or gate or fused" > gpurun_out/r04c_ops.log 2>&1 || { tail -20 gpurun_out/r04c_ops.log; exit 1; }
tail -2 gpurun_out/r04c_ops.log
T="F5E_HIP_LIB=$GRAFT_REPO_ROOT/f5e-tts_amd/libf5e_hip_tools.so"
bash tools/gpu_ab.sh r04c --args "--no-c3 --c4-total 0 --streams 0 --steps 20" "$T F5E_GEMM_GROUP_SHIFT=0" "$T" "$T F5E_GEMM_GROUP_SHIFT=0" "$T"
