cd $GRAFT_REPO_ROOT
T=tools/gemm_trace.bin
run() { out=$1; shift; M=$1; N=$2; K=$3; E=$4; shift 4; for v in "$@"; do timeout -k 5 90 $T $v $M $N $K $E || { echo "variant $v failed"; exit 1; }; done > gpurun_out/$out 2>&1; }
run r04ab_trace_out.txt 938 1024 1024 2   80 90 91 93 81 92 || { cat gpurun_out/r04ab_trace_out.txt | tail -5; exit 1; }
run r04ab_trace_ff2.txt 938 1024 2048 2   80 91 81 92 || exit 1
grep -h "variant\|per launch\|K-step" gpurun_out/r04ab_trace_*.txt
