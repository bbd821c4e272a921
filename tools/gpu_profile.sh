#!/bin/bash
# Profile driver for gpurun (run from the repo root on the GPU box):  tools/gpu_profile.sh <tag>
#   rocprofv3 --kernel-trace --stats of the C2 and C3 bench workloads (graph replay only: --no-roofline), then two separate
#   --pmc passes (FETCH_SIZE, WRITE_SIZE: they do not fit one pass, MI355X_MICROARCH.md "rocprofv3 PMC slots") of the C2
#   workload, digested by tools/pmc_traffic.py.  Summaries -> gpurun_out/<tag>_*; copy what is to be judged into profiles/.
tag=$1
R=$(pwd)
C2="--no-cpu-baseline --no-c3 --no-roofline --streams 0 --c4-total 0 --steps 5 --warmup 2"
cd /tmp && export TMPDIR=/tmp
if [ -z "$SKIP_STATS" ]; then
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/${tag}_prof_c2 -o c2 -- python3 $R/bench.py $C2 > $R/gpurun_out/${tag}_prof_c2.json 2> $R/gpurun_out/${tag}_prof_c2.err || { echo "prof c2 failed"; tail -3 $R/gpurun_out/${tag}_prof_c2.err; exit 1; }
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/${tag}_prof_c3 -o c3 -- python3 $R/bench.py --workload C3 --no-cpu-baseline --no-roofline --streams 0 --steps 2 --warmup 1 > $R/gpurun_out/${tag}_prof_c3.json 2> $R/gpurun_out/${tag}_prof_c3.err || { echo "prof c3 failed"; exit 1; }
fi
# Counter passes launch eagerly (--eager): counter collection crashed (SIGSEGV inside rocprofv3) on the 3700-node whole-loop
# graph and once hung on replays of the one-step graph; per-kernel counters do not depend on how a kernel was launched.
PMC="--no-cpu-baseline --no-c3 --no-roofline --streams 0 --c4-total 0 --steps 2 --warmup 1 --eager"
timeout -k 10 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $R/gpurun_out/${tag}_pmc_fetch -o f -- python3 $R/bench.py $PMC > /dev/null 2> $R/gpurun_out/${tag}_pmc_fetch.err || { echo "pmc fetch failed"; tail -3 $R/gpurun_out/${tag}_pmc_fetch.err; exit 1; }
timeout -k 10 300 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $R/gpurun_out/${tag}_pmc_write -o w -- python3 $R/bench.py $PMC > /dev/null 2> $R/gpurun_out/${tag}_pmc_write.err || { echo "pmc write failed"; exit 1; }
# Matrix-pipe counters (round 4): ONE eager pass each for C2 and C3 with the SQ / GRBM counters tools/prof_ops.py digests
# into per-op mfma_busy / effective clock (separate from the traffic passes: TCC and SQ counters are different passes anyway)
MF="SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY GRBM_GUI_ACTIVE"
if [ -z "$SKIP_MFMA" ]; then
timeout -k 10 400 rocprofv3 --kernel-trace --pmc $MF --output-format csv -d $R/gpurun_out/${tag}_pmc_mfma_c2 -o m -- python3 $R/bench.py $PMC > /dev/null 2> $R/gpurun_out/${tag}_pmc_mfma_c2.err || { echo "pmc mfma c2 failed"; tail -3 $R/gpurun_out/${tag}_pmc_mfma_c2.err; exit 1; }
timeout -k 10 600 rocprofv3 --kernel-trace --pmc $MF --output-format csv -d $R/gpurun_out/${tag}_pmc_mfma_c3 -o m -- python3 $R/bench.py --workload C3 --no-cpu-baseline --no-roofline --streams 0 --steps 1 --warmup 1 --eager > /dev/null 2> $R/gpurun_out/${tag}_pmc_mfma_c3.err || { echo "pmc mfma c3 failed"; tail -3 $R/gpurun_out/${tag}_pmc_mfma_c3.err; exit 1; }
fi
cd $R
if [ -z "$SKIP_MFMA" ]; then
python3 tools/prof_ops.py pmc gpurun_out/${tag}_pmc_mfma_c2 C2 gpurun_out/${tag}_ops_pmc_c2.json
python3 tools/prof_ops.py pmc gpurun_out/${tag}_pmc_mfma_c3 C3 gpurun_out/${tag}_ops_pmc_c3.json
fi
python3 tools/pmc_traffic.py gpurun_out/${tag}_pmc_fetch gpurun_out/${tag}_pmc_write gpurun_out/${tag}_pmc_traffic_c2.json
for w in c2 c3; do
  [ -n "$SKIP_STATS" ] && continue
  f=$(find gpurun_out/${tag}_prof_$w -name "*kernel_stats.csv" | head -1); cp $f gpurun_out/${tag}_kernel_stats_$w.csv
  python3 tools/prof_summary.py gpurun_out/${tag}_prof_$w 18
  W=$(echo $w | tr a-z A-Z); python3 tools/prof_ops.py trace gpurun_out/${tag}_prof_$w $W gpurun_out/${tag}_ops_trace_$w.json
done
if [ -z "$SKIP_STATS" ]; then
python3 tools/hbm_kernels.py gpurun_out/${tag}_kernel_stats_c2.csv C2 gpurun_out/${tag}_hbm_kernels_c2.json
python3 tools/hbm_kernels.py gpurun_out/${tag}_kernel_stats_c3.csv C3 gpurun_out/${tag}_hbm_kernels_c3.json
fi
# keep the merged scratch small: raw traces stay on the box
for d in gpurun_out/${tag}_prof_c2 gpurun_out/${tag}_prof_c3 gpurun_out/${tag}_pmc_fetch gpurun_out/${tag}_pmc_write gpurun_out/${tag}_pmc_mfma_c2 gpurun_out/${tag}_pmc_mfma_c3; do [ -d $d ] && find $d -type f -size +2M -delete; done; true
