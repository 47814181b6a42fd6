#!/bin/bash
# One gpurun call that regenerates the round's artefacts under gpurun_out/ (copy the ones to keep into profiles/):
#   bench line (N = 1, default flags), rocprofv3 --kernel-trace --stats of the same command, MFMA / wave-state / LDS counters,
#   conv / gemm / front-end micro-benchmarks, training benchmark, in-kernel stamps of the conv kernel (diagnostic build).
set -e
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
TAG=${1:-r02}
cd $R
python3 bench.py > gpurun_out/${TAG}_bench.json 2> gpurun_out/${TAG}_bench.err
echo bench done
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/${TAG}_stats -- python3 $R/bench.py --no-cpu-baseline --no-small-batch --no-parity-mode --no-h2d --no-train-leg > $R/gpurun_out/${TAG}_bench_profiled.json 2>/dev/null
echo stats done
cd $R
bash scripts/profile_mfma_counters.sh ${TAG}
python3 scripts/conv_bench.py 10240 bf16 > gpurun_out/${TAG}_conv_bench.txt 2>/dev/null
python3 scripts/conv_bench.py 10240 f32 5 >> gpurun_out/${TAG}_conv_bench.txt 2>/dev/null
python3 scripts/gemm_bench.py 10240 bf16 > gpurun_out/${TAG}_gemm_bench.txt 2>/dev/null
python3 scripts/gemm_bench.py 10240 f32 >> gpurun_out/${TAG}_gemm_bench.txt 2>/dev/null
python3 scripts/fe_bench.py > gpurun_out/${TAG}_fe_bench.txt 2>/dev/null
python3 scripts/train_bench.py > gpurun_out/${TAG}_train_bench.txt 2>/dev/null
python3 scripts/wgrad_bench.py > gpurun_out/${TAG}_wgrad_bench.txt 2>/dev/null
echo all done
