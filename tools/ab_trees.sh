#!/bin/bash
# A/B of the working tree against ANOTHER TREE (e.g. `git archive <commit>` unpacked into ab_old_tree/, its library built there) on ONE
# GPU box: alternating bench runs, a different episode stream per round.  For comparisons across an ABI change, where
# tools/ab_commit.sh (old kernels under the new host code) cannot be used.
#   usage (GPU box): tools/ab_trees.sh <old tree> [rounds] [bench args...]
ROOT=$(cd "$(dirname "$0")/.." && pwd)
OLD=$ROOT/${1:-ab_old_tree}; R=${2:-3}; shift 2
for i in $(seq 1 $R); do
  export ORR_BENCH_SEED=$((i - 1))
  for W in old new; do
    if [ $W = old ]; then T=$OLD; else T=$ROOT; fi
    (cd $T && python3 bench.py --steps 1500 --warmup 100 --no-cpu-baseline "$@" 2>/dev/null) | python3 -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$W', round(d['value']/1e6,3), 'M steps/s  kernel', round(d['roofline']['kernel_ms'],4), 'ms')"
  done
done
