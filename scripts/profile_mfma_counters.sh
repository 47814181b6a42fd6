#!/bin/bash
# MFMA-pipe / wave-state / LDS counters for every conv and FC kernel of the bench.py pipeline (bf16, 10 240 clips per launch).
# Separate --pmc passes with --kernel-trace only (no --stats, no other trace domain). Output: gpurun_out/mfma_counters.txt
set -e
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
TAG=${1:-pmc}
cd /tmp
B="python3 $R/bench.py --steps 3 --warmup 1 --prewarm-seconds 0 --no-cpu-baseline --no-small-batch --no-parity-mode --no-h2d --no-train-leg"
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $R/gpurun_out/${TAG}_a -- $B > /dev/null 2>$R/gpurun_out/${TAG}_a.err
rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_INSTS_VALU SQ_INSTS_MFMA GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $R/gpurun_out/${TAG}_b -- $B > /dev/null 2>$R/gpurun_out/${TAG}_b.err
cd $R
python3 scripts/pmc_summary.py gpurun_out/${TAG}_mfma_counters.txt gpurun_out/${TAG}_a gpurun_out/${TAG}_b > /dev/null
echo counters done
