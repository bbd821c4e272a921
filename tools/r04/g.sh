cd $GRAFT_REPO_ROOT
T="F5E_HIP_LIB=$GRAFT_REPO_ROOT/f5e-tts_amd/libf5e_hip_tools.so"
timeout -k 10 400 python -m pytest tests/test_ops_gpu.py -x -q -m gpu 2>&1 | tail -5 || exit 1
for b in 0 1; do for v in x 0; do
  E="F5E_GEMM_BIG=$b"; [ $v = 0 ] && E="$E F5E_GEMM_VAR=0"; [ $b = 0 ] && [ $v = 0 ] && continue
  env $T $E timeout -k 10 300 python bench.py --no-cpu-baseline --no-c3 --c4-total 0 --streams 0 --steps 10 --batch 2 > gpurun_out/r04aj_$b$v.json 2>gpurun_out/r04aj_$b$v.err || { tail -5 gpurun_out/r04aj_$b$v.err; exit 1; }
  python -c "
import json; d=json.loads(open('gpurun_out/r04aj_$b$v.json').read().strip().splitlines()[-1]); print('$E', d['value'], d['ms_per_step'], d['roofline']['all_ops_us'], d.get('parity'))"
done; done
bash tools/gpu_ab.sh r04ak --args "--workload C4 --steps 100 --streams 0" "$T F5E_GEMM_BIG=0" "$T F5E_GEMM_BIG=1" "$T F5E_GEMM_BIG=0" "$T F5E_GEMM_BIG=1" | grep -v "C3 None"
