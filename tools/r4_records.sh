#!/bin/bash
# GPU box: the round-4 measurement set on the final code, in ONE call.  Everything lands under gpurun_out/summ/ (copied into profiles/ afterwards).
# usage: tools/r4_records.sh [part ...]   parts: bench profiles probe drift dual train    (default: all but train)
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/summ
mkdir -p $OUT
PARTS=${@:-bench profiles probe drift dual}
cd $ROOT
for P in $PARTS; do
case $P in
bench)
  python3 bench.py > $OUT/r04_laikago4096_bench.json 2> $OUT/bench.err
  python3 bench.py --gpus 1 --steps 20 --warmup 5 > $OUT/r04_driver20_bench.json 2>> $OUT/bench.err
  python3 bench.py --no-cpu-baseline --repeats 5 > $OUT/r04_laikago4096_median5_bench.json 2>> $OUT/bench.err
  python3 bench.py --no-cpu-baseline --no-randomizer > $OUT/r04_laikago4096_norand_bench.json 2>> $OUT/bench.err
  python3 bench.py --no-cpu-baseline --config minicheetah4096 > $OUT/r04_minicheetah4096_bench.json 2>> $OUT/bench.err
  python3 bench.py --no-cpu-baseline --config mixed8192 > $OUT/r04_mixed8192_bench.json 2>> $OUT/bench.err
  python3 bench.py --no-cpu-baseline --config mixed8192 --gpus 1 --steps 20 --warmup 5 > $OUT/r04_mixed8192_driver20_bench.json 2>> $OUT/bench.err
  python3 bench.py --no-cpu-baseline --robots-per-gpu 16384 > $OUT/r04_laikago16384_bench.json 2>> $OUT/bench.err
  ORR_FORCE_DIST=1 python3 bench.py --spawn --gpus 1 --no-cpu-baseline > $OUT/r04_launcher_rccl_one_rank_bench.json 2>> $OUT/bench.err
  echo "bench done"; python3 -c "
import json,glob
for f in sorted(glob.glob('$OUT/r04_*bench.json')):
    try:
        d=json.load(open(f)); r=d['roofline']; print(f.split('/')[-1], round(d['value']/1e6,2), 'M', 'kern', round(r['kernel_ms'],4), 'b2b', round(r['kernel_ms_back_to_back'],4), 'stale', r['pmc_stale'])
    except Exception as e: print(f, 'ERR', e)"
  ;;
profiles)
  bash tools/profile_all.sh r04 > $OUT/profile_all.log 2>&1
  echo "profiles done"; ls $OUT | grep -c r04_
  ;;
probe)
  python3 tools/policy_probe.py --robots 1024 --seeds 1 2 --out $OUT/r04_policy_probe.json > $OUT/r04_policy_probe.txt 2>&1
  python3 tools/policy_probe.py --sensitivity --robots 256 --seeds 1 --out $OUT/r04_laikago_sensitivity.json > $OUT/r04_laikago_sensitivity.txt 2>&1
  echo "probe done"; grep -c finished $OUT/r04_policy_probe.txt
  ;;
drift)
  python3 tools/drift_floor_spread.py --runs 12 --out-dir $OUT > $OUT/drift_floor_spread.log 2>&1
  echo "drift done"; tail -1 $OUT/drift_floor_spread.log
  ;;
dual)
  python3 tools/dual_contact.py 2000 300 2>&1 | grep -v "^{" | grep -v amdgpu.ids > $OUT/r04_dual_contact.txt
  cp gpurun_out/dual_contact.json $OUT/r04_dual_contact.json
  echo "dual done"
  ;;
train)
  python3 train.py --iters 24000 --log $OUT/r04_train_laikago_pace_fused.json --save $OUT/r04_laikago_pace_scratch.zip > $OUT/train.log 2>&1
  python3 train.py --eval $OUT/r04_laikago_pace_scratch.zip > $OUT/r04_train_eval.txt 2>&1
  echo "train done"; tail -2 $OUT/r04_train_eval.txt
  ;;
esac
done
