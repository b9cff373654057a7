#!/bin/bash
# Bench a set of differently compiled libraries on ONE GPU box (development aid): openroborl_amd/lib_var_<name>.so (tools/build_variants.py)
#   usage (GPU box): tools/ab_variants.sh [rounds] [bench args...]
ROOT=$(cd "$(dirname "$0")/.." && pwd)
R=${1:-2}; shift
for i in $(seq 1 $R); do
  export ORR_BENCH_SEED=$((i - 1))
  for L in $ROOT/openroborl_amd/lib_var_*.so; do
    export ORR_LIB_PATH=$L ORR_ALLOW_STALE_LIB=1
    python3 $ROOT/bench.py --steps 1500 --warmup 100 --no-cpu-baseline "$@" 2>/dev/null | python3 -c "
import json,sys,os; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(os.path.basename('$L'), round(d['value']/1e6,3), 'M steps/s  kernel', round(d['roofline']['kernel_ms'],4), 'ms')"
  done
done
