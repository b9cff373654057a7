#!/bin/bash
# A/B of the working tree against a git commit on ONE GPU box (development aid).
#   here (build container):  tools/ab_commit.sh build <commit>     -> openroborl_amd/lib_ab_old.so from that commit's kernels
#   on the GPU box (gpurun): tools/ab_commit.sh run [rounds]        -> alternating bench runs old / new
ROOT=$(cd "$(dirname "$0")/.." && pwd)
if [ "$1" = build ]; then
  C=${2:-HEAD}; TMP=$(mktemp -d)
  git -C "$ROOT" archive "$C" openroborl_amd/csrc include | tar -x -C "$TMP"
  # the SAME compiler flags as the shipped library (openroborl_amd/_lib.py HIPCC_FLAGS)
  python3 - "$ROOT" "$TMP" <<'PY' && echo "built lib_ab_old.so from $C"
import subprocess, sys
root, tmp = sys.argv[1], sys.argv[2]
sys.path.insert(0, root)
from openroborl_amd import _lib
fl = [f for f in _lib.HIPCC_FLAGS if f != "-shared"]
subprocess.check_call([_lib.HIPCC] + fl + ["-c", "-o", tmp + "/k.o", tmp + "/openroborl_amd/csrc/orr_kernels.hip"])
subprocess.check_call([_lib.HIPCC, "--offload-arch=gfx950", "-O3", "-fPIC", "-std=c++17", "-c", "-o", tmp + "/p.o", tmp + "/openroborl_amd/csrc/orr_policy.hip"])
subprocess.check_call([_lib.HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", root + "/openroborl_amd/lib_ab_old.so", tmp + "/k.o", tmp + "/p.o"])
PY
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
