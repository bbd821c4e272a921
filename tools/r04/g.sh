cd $GRAFT_REPO_ROOT
T="F5E_HIP_LIB=$GRAFT_REPO_ROOT/f5e-tts_amd/libf5e_hip_tools.so"
bash tools/gpu_ab.sh r04ae --args "--no-c3 --c4-total 0 --streams 0 --steps 20" "$T F5E_PF_SCHEME=0" "$T F5E_PF_SCHEME=1" "$T F5E_PF_SCHEME=5" "$T F5E_PF_SCHEME=6" "$T F5E_PF_SCHEME=7" "$T F5E_PF_SCHEME=0" "$T F5E_PF_SCHEME=1" "$T F5E_PF_SCHEME=5" "$T F5E_PF_SCHEME=6" "$T F5E_PF_SCHEME=7"
