cd $GRAFT_REPO_ROOT
T="F5E_HIP_LIB=$GRAFT_REPO_ROOT/f5e-tts_amd/libf5e_hip_tools.so F5E_GEMM_VAR=10"
bash tools/gpu_ab.sh r04t --args "--no-c3 --c4-total 0 --streams 0 --steps 20" "$T F5E_GEMM_WIDE=0" "$T F5E_GEMM_WIDE=1" "$T F5E_GEMM_WIDE=2" "$T F5E_GEMM_WIDE=3" "$T F5E_GEMM_WIDE=0" "$T F5E_GEMM_WIDE=1" "$T F5E_GEMM_WIDE=2" "$T F5E_GEMM_WIDE=3"
