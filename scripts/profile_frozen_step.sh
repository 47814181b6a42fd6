#!/bin/bash
# Kernel trace of the frozen-CNN training step (512 bags, bf16), eager and as a HIP graph: per step the CNN forward's span, the span of
# everything behind it (head forward / backward, Adam), the sum of those kernels' durations and the GPU idle time between them
# (scripts/head_span.py), plus the per-kernel totals of the eager run. Output: gpurun_out/frozen_step_trace.txt
set -e
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/frozen_step_trace.txt
: > $OUT
cd /tmp
for G in 0 1; do
  MLA_TRAIN_GRAPH=$G rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/frozen_trace_$G -- python3 $R/scripts/train_prof.py bf16 frozen > /dev/null 2>&1
  echo "# MLA_TRAIN_GRAPH=$G (rocprofv3 --kernel-trace, scripts/train_prof.py bf16 frozen: 512 bags, last 4 of 8 steps)" >> $OUT
  python3 $R/scripts/head_span.py $(ls $R/gpurun_out/frozen_trace_$G/*/*kernel_trace.csv | head -1) >> $OUT
done
echo "# per-kernel totals of the eager run (ms per step = total / 8 steps), head kernels only" >> $OUT
python3 - $(ls $R/gpurun_out/frozen_trace_0/*/*kernel_stats.csv | head -1) >> $OUT <<'P'
import csv, sys
for r in csv.DictReader(open(sys.argv[1])):
    n = r["Name"].replace("(anonymous namespace)::", "")
    if "conv3x3_kernel" in n or "conv1_patch" in n or "gemm_kernel<mma::bf16_t" in n:
        continue
    print("%-100s calls/step %5.1f  avg %7.1f us  ms/step %6.3f" % (n[:100], int(r["Calls"]) / 8, float(r["AverageNs"]) / 1e3, float(r["TotalDurationNs"]) / 8e6))
P
cat $OUT
