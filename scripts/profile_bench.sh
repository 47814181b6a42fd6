#!/bin/bash
# round-1 recipe: bench line + rocprofv3 --kernel-trace --stats of the same command (round 2: scripts/profile_round.sh)
set -e
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/bench_stats2 -- python3 $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-small-batch > $R/gpurun_out/bench_prof_line.json 2>/dev/null
cd $R
python3 bench.py --steps 10 --warmup 3 2>/dev/null | tail -1 > gpurun_out/bench_line.json
python3 scripts/fe_bench.py > gpurun_out/fe_bench.txt 2>/dev/null
python3 scripts/conv_bench.py 10240 bf16 > gpurun_out/conv_bench.txt 2>/dev/null
python3 scripts/conv_bench.py 10240 f32 5 >> gpurun_out/conv_bench.txt 2>/dev/null
python3 scripts/gemm_bench.py 10240 bf16 > gpurun_out/gemm_bench.txt 2>/dev/null
python3 scripts/gemm_bench.py 10240 f32 >> gpurun_out/gemm_bench.txt 2>/dev/null
python3 scripts/train_bench.py > gpurun_out/train_bench.txt 2>/dev/null
echo done
