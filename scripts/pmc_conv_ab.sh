#!/bin/bash
# MFMA-pipe counters of the conv micro-benchmark for two library builds (A/B on one device). usage: pmc_conv_ab.sh <layers> <libB path>
set -e
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
LAYERS=${1:-4}
LIBB=$2
cd /tmp
for V in main alt; do
  if [ "$V" = "alt" ]; then export MLA_HIP_LIB=$R/$LIBB; fi
  rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $R/gpurun_out/pmcab_$V -- python3 $R/scripts/conv_bench.py 10240 bf16 4 $LAYERS > /dev/null 2>&1
done
cd $R
python3 scripts/pmc_summary.py gpurun_out/pmcab_main.txt gpurun_out/pmcab_main > /dev/null
python3 scripts/pmc_summary.py gpurun_out/pmcab_alt.txt gpurun_out/pmcab_alt > /dev/null
echo MAIN; grep -E "^conv|mean_us|clock|MFMA pipe|WAIT_ANY /|WAIT_INST_ANY /" gpurun_out/pmcab_main.txt
echo ALT; grep -E "^conv|mean_us|clock|MFMA pipe|WAIT_ANY /|WAIT_INST_ANY /" gpurun_out/pmcab_alt.txt
