cd $GRAFT_REPO_ROOT
OLD="F5E_HIP_LIB=$GRAFT_REPO_ROOT/f5e-tts_amd/libf5e_hip_old.so"
NEW="F5E_HIP_LIB=$GRAFT_REPO_ROOT/f5e-tts_amd/libf5e_hip.so"
bash tools/gpu_ab.sh r04bl --args "--no-c3 --c4-total 0 --streams 0 --steps 20" "$OLD" "$NEW" "$OLD" "$NEW" | grep "C2 "
