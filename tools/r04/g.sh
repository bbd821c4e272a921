cd $GRAFT_REPO_ROOT
timeout -k 10 900 python -m pytest tests/test_ops_gpu.py tests/test_baseline_configs_gpu.py -x -q -m gpu 2>&1 | tail -3 || exit 1
TL="F5E_HIP_LIB=$GRAFT_REPO_ROOT/f5e-tts_amd/libf5e_hip_tools.so"
for E in "F5E_GEMM_BIG=1" "F5E_GEMM_BIG=0"; do
  env $TL $E timeout -k 10 300 python bench.py --no-cpu-baseline --no-c3 --c4-total 0 --streams 0 --steps 10 --batch 2 > gpurun_out/r04as_b.json 2>gpurun_out/r04as_b.err || { tail -5 gpurun_out/r04as_b.err; exit 1; }
  python -c "
import json; d=json.loads(open('gpurun_out/r04as_b.json').read().strip().splitlines()[-1]); print('$E', d['value'], d['ms_per_step'], d['roofline']['all_ops_us'])"
done
bash tools/gpu_ab.sh r04at --args "--workload C4 --steps 100 --streams 0" "$TL F5E_GEMM_BIG=0" "$TL F5E_GEMM_BIG=1" "$TL F5E_GEMM_BIG=0" "$TL F5E_GEMM_BIG=1" | grep -v "C3 None"
timeout -k 10 900 python bench.py --no-cpu-baseline --no-c3 --steps 10 > gpurun_out/r04at_default.json 2>/dev/null; python -c "
import json; d=json.loads(open('gpurun_out/r04at_default.json').read().strip().splitlines()[-1]); print('default', d['value'], d['ms_per_step'], d['scaling_c4']['mel_frames_per_sec'], d['concurrent']['value'])"
