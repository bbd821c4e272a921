cd $GRAFT_REPO_ROOT
P=tools/l2_probe.bin
{
for per in 1 2 3; do for mode in 0 1 2; do for nreg in 8 64 512; do timeout -k 5 60 $P $mode $per $nreg || exit 1; done; done; done
} > gpurun_out/r04b_l2_probe.txt 2>&1
cd /tmp && export TMPDIR=/tmp
timeout -k 5 300 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/r04b_blas_prof -o blas -- python3 $GRAFT_REPO_ROOT/tools/blas_yardstick.py > $GRAFT_REPO_ROOT/gpurun_out/r04b_blas.txt 2>&1
cd $GRAFT_REPO_ROOT
f=$(find gpurun_out/r04b_blas_prof -name "*kernel_stats.csv" | head -1); cp $f gpurun_out/r04b_blas_kernel_stats.csv
find gpurun_out/r04b_blas_prof -type f -size +1M -delete
cat gpurun_out/r04b_l2_probe.txt
