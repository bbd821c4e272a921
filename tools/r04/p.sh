cd $GRAFT_REPO_ROOT
bash tools/gpu_profile.sh r04v > gpurun_out/r04v_profile.log 2>&1 || { tail -20 gpurun_out/r04v_profile.log; exit 1; }
tail -60 gpurun_out/r04v_profile.log
