cd $GRAFT_REPO_ROOT
T=tools/gemm_trace.bin
timeout -k 10 400 python -m pytest tests/test_ops_gpu.py -x -q -m gpu 2>&1 | tail -3 || exit 1
{
for spec in "100 1876 2048 1024 0" "90 1876 2048 1024 1" "90 1876 3072 1024 3" "90 1876 1024 1024 2" "90 1876 1024 2048 2" "81 938 1024 1024 2" "81 938 1024 2048 2" "83 938 2048 1024 1" "74 938 3072 1024 3"; do
  timeout -k 5 60 $T $spec || exit 1
done
} > gpurun_out/r04bd_trace.txt 2>&1
grep -E "^variant|per workgroup|K-step" gpurun_out/r04bd_trace.txt
OLD="F5E_HIP_LIB=$GRAFT_REPO_ROOT/f5e-tts_amd/libf5e_hip_old.so"
NEW="F5E_HIP_LIB=$GRAFT_REPO_ROOT/f5e-tts_amd/libf5e_hip.so"
bash tools/gpu_ab.sh r04be --args "--no-c3 --c4-total 0 --streams 0 --steps 20" "$OLD" "$NEW" "$OLD" "$NEW" | grep -v "C3 None"
bash tools/gpu_ab.sh r04bf --args "--workload C4 --steps 100 --streams 0" "$OLD" "$NEW" "$OLD" "$NEW" | grep -v "C3 None"
