#!/bin/bash
# Run on the GPU box (through gpurun): kernel-trace stats + separate PMC passes for the step kernel.
# usage: tools/profile_gpu.sh <tag> [bench.py config] [extra bench.py flags, e.g. --no-randomizer]      outputs under gpurun_out/<tag>/
# Every pass keeps its log (gpurun_out/<tag>/*.log): if a profiler pass crashes, copy that log to profiles/ (the r02
# --pmc crash with the 6000-launch warm-up left no record).
set -u
TAG=${1:-prof}
CFG=${2:-laikago4096}
EXTRA=${3:-}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/$TAG
mkdir -p $OUT
python3 -c "import sys; sys.path.insert(0, '$ROOT'); from openroborl_amd import _lib; print(_lib.library_hash())" > $OUT/source_hash.txt
cd /tmp && export TMPDIR=/tmp
BENCH="python3 $ROOT/bench.py --config $CFG --steps 40 --warmup 10 --no-cpu-baseline $EXTRA"
BENCH_TRACE="python3 $ROOT/bench.py --config $CFG --no-cpu-baseline $EXTRA"   # = the default bench.py run
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- $BENCH_TRACE > $OUT/trace.log 2>&1 || echo "trace pass failed (log: $OUT/trace.log)"
export ORR_BENCH_WARMUP_FLOOR=50     # PMC passes: a short warm-up (every dispatch is serialised and recorded)
for P in "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVE_CYCLES SQ_BUSY_CYCLES" \
         "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_SMEM" \
         "FETCH_SIZE" "WRITE_SIZE" "GRBM_GUI_ACTIVE SQ_INSTS_VALU_MFMA_MOPS_F32"; do
  N=$(echo $P | tr ' ' '_' | cut -c1-40)
  timeout -k 10 300 rocprofv3 --kernel-trace --pmc $P --output-format csv -d $OUT/pmc_$N -- $BENCH > $OUT/pmc_$N.log 2>&1 || echo "pass $N failed (log: $OUT/pmc_$N.log)"
done
find $OUT -name "*.csv" | head -40
