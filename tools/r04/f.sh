cd $GRAFT_REPO_ROOT
T=tools/gemm_trace.bin
run() { out=$1; shift; M=$1; N=$2; K=$3; E=$4; shift 4; for v in "$@"; do timeout -k 5 60 $T $v $M $N $K $E || { echo "variant $v failed"; exit 1; }; done > gpurun_out/$out 2>&1; }
run r04n_trace_out.txt 938 1024 1024 2   10 42 81 || exit 1
run r04n_trace_ff2.txt 938 1024 2048 2   10 42 81 || exit 1
run r04n_trace_ff1.txt 938 2048 1024 1   10 75 83 || exit 1
run r04n_trace_qkv.txt 938 3072 1024 3   10 73 74 || exit 1
grep -h "variant\|per workgroup\|per launch\|K-step" gpurun_out/r04n_trace_*.txt
