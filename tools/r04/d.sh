cd $GRAFT_REPO_ROOT
T=tools/gemm_trace.bin
run() { out=$1; shift; M=$1; N=$2; K=$3; E=$4; shift 4; for v in "$@"; do timeout -k 5 60 $T $v $M $N $K $E || { echo "variant $v failed"; exit 1; }; done > gpurun_out/$out 2>&1; }
run r04e_trace_out.txt 938 1024 1024 2   0 38 39 10 41 || exit 1
run r04e_trace_ff2.txt 938 1024 2048 2   3 38 39 || exit 1
run r04e_trace_ff1.txt 938 2048 1024 1   0 38 51 54 52 10 41 || exit 1
run r04e_trace_qkv.txt 938 3072 1024 3   0 38 50 51 52 53 10 41 || exit 1
grep -h "variant\|per workgroup\|per launch" gpurun_out/r04e_trace_*.txt
