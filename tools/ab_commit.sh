#!/bin/bash
# A/B of the working tree against a git commit on ONE GPU box (development aid).
#   here (build container):  tools/ab_commit.sh build <commit>     -> openroborl_amd/lib_ab_old.so from that commit's kernels
#   on the GPU box (gpurun): tools/ab_commit.sh run [rounds]        -> alternating bench runs old / new
ROOT=$(cd "$(dirname "$0")/.." && pwd)
if [ "$1" = build ]; then
  C=${2:-HEAD}; TMP=$(mktemp -d)
  git -C "$ROOT" archive "$C" openroborl_amd/csrc include | tar -x -C "$TMP"
  HIPCC=/opt/rocm/bin/hipcc
  FL="--offload-arch=gfx950 -O2 -fPIC -std=c++17 -Wno-unused-value -fno-slp-vectorize -mllvm -amdgpu-sched-strategy=iterative-ilp"
  $HIPCC $FL -c -o $TMP/k.o $TMP/openroborl_amd/csrc/orr_kernels.hip && $HIPCC --offload-arch=gfx950 -O3 -fPIC -std=c++17 -c -o $TMP/p.o $TMP/openroborl_amd/csrc/orr_policy.hip && \
    $HIPCC --offload-arch=gfx950 -shared -fPIC -o $ROOT/openroborl_amd/lib_ab_old.so $TMP/k.o $TMP/p.o && echo "built lib_ab_old.so from $C"
  python3 -c "import sys; sys.path.insert(0, '$ROOT'); from openroborl_amd import _lib; _lib.build()" && echo "working-tree library up to date"
  exit 0
fi
R=${2:-3}
for i in $(seq 1 $R); do
  export ORR_BENCH_SEED=$((i - 1))    # a different episode stream per round: changes of numerics move the workload (contacts, joint limits, resets) by +-1 %
  for W in old new; do
    if [ $W = old ]; then export ORR_LIB_PATH=$ROOT/openroborl_amd/lib_ab_old.so; else unset ORR_LIB_PATH; fi
    python3 $ROOT/bench.py --steps 1500 --warmup 100 --no-cpu-baseline 2>/dev/null | python3 -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$W', round(d['value']/1e6,3), 'M steps/s  kernel', round(d['roofline']['kernel_ms'],4), 'ms')"
  done
done
