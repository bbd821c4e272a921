#!/bin/bash
# Counter passes over the C3-size attention launch (tools/attn_pmc_workload.py), run from the repo root on the GPU box:
#   tools/attn_pmc.sh <tag>   ->  gpurun_out/<tag>_pmc_attn_{a,b}.json  (tools/pmc_mfma.py digests)
# Separate --pmc passes with --kernel-trace only (8 SQ slots per pass; MI355X_MICROARCH.md "rocprofv3 PMC slots").
tag=$1
R=$(pwd)
cd /tmp && export TMPDIR=/tmp
timeout -k 10 240 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS GRBM_GUI_ACTIVE --output-format csv -d $R/gpurun_out/${tag}_pmc_attn_a -o a -- python3 $R/tools/attn_pmc_workload.py > /dev/null 2> $R/gpurun_out/${tag}_pmc_attn_a.err || { echo "pmc a failed"; tail -3 $R/gpurun_out/${tag}_pmc_attn_a.err; exit 1; }
timeout -k 10 240 rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS GRBM_GUI_ACTIVE --output-format csv -d $R/gpurun_out/${tag}_pmc_attn_b -o b -- python3 $R/tools/attn_pmc_workload.py > /dev/null 2> $R/gpurun_out/${tag}_pmc_attn_b.err || { echo "pmc b failed"; tail -3 $R/gpurun_out/${tag}_pmc_attn_b.err; exit 1; }
cd $R
python3 tools/pmc_mfma.py gpurun_out/${tag}_pmc_attn_a attn_fwd gpurun_out/${tag}_pmc_attn_a.json
python3 tools/pmc_mfma.py gpurun_out/${tag}_pmc_attn_b attn_fwd gpurun_out/${tag}_pmc_attn_b.json
for d in gpurun_out/${tag}_pmc_attn_a gpurun_out/${tag}_pmc_attn_b; do [ -d $d ] && find $d -type f -size +2M -delete; done; true
