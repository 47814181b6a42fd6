#!/bin/bash
# whole-pipeline A/B on one device: bench.py with the in-tree library ("main") and with variants, alternating. usage: ab_pipeline.sh <rounds> <variant.so>...
R=${GRAFT_REPO_ROOT:-/root/repo}
N=$1; shift
for i in $(seq $N); do
  for lib in main "$@"; do
    if [ "$lib" = "main" ]; then unset MLA_HIP_LIB; else export MLA_HIP_LIB=$R/$lib; fi
    python3 $R/bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-small-batch --no-parity-mode --no-h2d --no-train-leg 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('$lib', round(d['value']), round(d['ms_per_step'],3), 'conv_stack', round(d['roofline_conv_stack']['frac'],4), {k: v for k, v in d['roofline_conv_stack']['per_kernel_frac'].items() if k != 'conv1'}, 'sum_kernel_ms', round(sum(d['kernel_ms'].values()),3))"
  done
done
