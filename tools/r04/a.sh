set -x
cd $GRAFT_REPO_ROOT
T=tools/gemm_trace.bin
for v in 0 20 21 22 3 25; do timeout -k 5 60 $T $v 938 1024 1024 2 || exit 1; done > gpurun_out/r04a_trace_out.txt 2>&1
for v in 0 20 21; do timeout -k 5 60 $T $v 938 3072 1024 3 || exit 1; done > gpurun_out/r04a_trace_qkv.txt 2>&1
for v in 0 20 21; do timeout -k 5 60 $T $v 938 2048 1024 1 || exit 1; done > gpurun_out/r04a_trace_ff1.txt 2>&1
for v in 3 23 24; do timeout -k 5 60 $T $v 938 1024 2048 2 || exit 1; done > gpurun_out/r04a_trace_ff2.txt 2>&1
timeout -k 5 300 python tools/blas_yardstick.py > gpurun_out/r04a_blas.txt 2>&1
cat gpurun_out/r04a_trace_out.txt | grep -v "^+"
