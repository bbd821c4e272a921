cd $GRAFT_REPO_ROOT
TL="F5E_HIP_LIB=$GRAFT_REPO_ROOT/f5e-tts_amd/libf5e_hip_tools.so"
bash tools/gpu_ab.sh r04av --args "--workload C5 --steps 10" "$TL F5E_GEMM_BIG=0" "$TL F5E_GEMM_BIG=1" "$TL F5E_GEMM_BIG=0" "$TL F5E_GEMM_BIG=1" | grep -v "C3 None"
