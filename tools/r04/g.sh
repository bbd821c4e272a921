cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/r04bp_prof -o b2 -- python3 $R/bench.py --batch 2 --no-cpu-baseline --no-c3 --no-roofline --streams 0 --c4-total 0 --steps 5 --warmup 2 > $R/gpurun_out/r04bp.json 2> $R/gpurun_out/r04bp.err || { tail -5 $R/gpurun_out/r04bp.err; exit 1; }
cd $R
f=$(find gpurun_out/r04bp_prof -name "*kernel_stats.csv" | head -1); cp $f gpurun_out/r04bp_kernel_stats_batch2.csv
python3 tools/prof_ops.py trace gpurun_out/r04bp_prof C2 gpurun_out/r04bp_ops_trace_batch2.json
find gpurun_out/r04bp_prof -type f -size +2M -delete
