cd $GRAFT_REPO_ROOT
T="F5E_HIP_LIB=$GRAFT_REPO_ROOT/f5e-tts_amd/libf5e_hip_tools.so"
bash tools/gpu_ab.sh r04bv --args "--workload C4 --steps 100 --streams 0" "$T F5E_PF_SCHEME=3" "$T F5E_PF_SCHEME=7" "$T F5E_PF_SCHEME=6" "$T F5E_PF_SCHEME=5" "$T F5E_PF_SCHEME=0" "$T F5E_PF_SCHEME=3" "$T F5E_PF_SCHEME=7" "$T F5E_PF_SCHEME=6" "$T F5E_PF_SCHEME=5" "$T F5E_PF_SCHEME=0" | grep "C2 " | awk '{print NR, $2, $4}'
bash tools/gpu_ab.sh r04bw --args "--batch 2 --no-c3 --c4-total 0 --streams 0 --steps 10" "$T F5E_PF_SCHEME=3" "$T F5E_PF_SCHEME=7" "$T F5E_PF_SCHEME=6" | grep "C2 "
