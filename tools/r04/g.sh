cd $GRAFT_REPO_ROOT
T=tools/gemm_trace.bin
timeout -k 10 400 python -m pytest tests/test_ops_gpu.py -x -q -m gpu 2>&1 | tail -3 || exit 1
{
for spec in "74 938 3072 1024 3" "77 938 3072 1024 3"; do timeout -k 5 60 $T $spec || exit 1; done
} > gpurun_out/r04bm_trace.txt 2>&1
grep -E "^variant|per workgroup|K-step|epilogue" gpurun_out/r04bm_trace.txt
OLD="F5E_HIP_LIB=$GRAFT_REPO_ROOT/f5e-tts_amd/libf5e_hip_old.so"
NEW="F5E_HIP_LIB=$GRAFT_REPO_ROOT/f5e-tts_amd/libf5e_hip.so"
bash tools/gpu_ab.sh r04bn --args "--no-c3 --c4-total 0 --streams 0 --steps 20" "$OLD" "$NEW" "$OLD" "$NEW" | grep "C2 "
