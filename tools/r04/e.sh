cd $GRAFT_REPO_ROOT
T=tools/gemm_trace.bin
run() { out=$1; shift; M=$1; N=$2; K=$3; E=$4; shift 4; for v in "$@"; do timeout -k 5 60 $T $v $M $N $K $E || { echo "variant $v failed"; exit 1; }; done > gpurun_out/$out 2>&1; }
run r04f_trace_out.txt 938 1024 1024 2   10 39 42 || exit 1
run r04f_trace_ff2.txt 938 1024 2048 2   10 39 42 || exit 1
grep -h "variant\|per workgroup\|per launch\|loader\|epilogue:" gpurun_out/r04f_trace_*.txt
timeout -k 10 600 python -m pytest tests/test_ops_gpu.py -x -q -m gpu > gpurun_out/r04f_ops.log 2>&1 || { tail -30 gpurun_out/r04f_ops.log; exit 1; }
tail -2 gpurun_out/r04f_ops.log
timeout -k 10 900 python -m pytest tests/test_e2e_gpu.py -x -q -m gpu > gpurun_out/r04f_e2e.log 2>&1 || { tail -30 gpurun_out/r04f_e2e.log; exit 1; }
tail -2 gpurun_out/r04f_e2e.log
T="F5E_HIP_LIB=$GRAFT_REPO_ROOT/f5e-tts_amd/libf5e_hip_tools.so"
bash tools/gpu_ab.sh r04f --args "--no-c3 --c4-total 0 --streams 0 --steps 20" "$T F5E_GEMM_ROLE=0" "$T" "$T F5E_GEMM_ROLE=0" "$T"
