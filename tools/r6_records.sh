#!/bin/bash
# GPU box: the round-6 measurement set, part by part (each part fits one gpurun call).  Everything lands under gpurun_out/r6/ and is copied
# into profiles/r06_* (and tests/golden/drift_factors.json) afterwards.   usage: tools/r6_records.sh <part> [...]
#   constants  rule P5 of tools/identify_r6.py (cross-robot choice of the solver constants)          - ONCE, before everything else
#   cv A B ..  (via tools/identify_r6.py cv --only A B --sequential --minutes 5)                       - the six splits, once each
#   all4       P6: the run whose fit set is all four Laikago policies (what ships, in sample)          minimal   P7 on it
#   mc         P8: the mini-cheetah run + its P7
#   probe      all five shipped policies on the shipped tables          drift     float32 drift factors on the current sources
#   bench      bench lines (--repeats 5 for the three configs)          profiles  rocprofv3 kernel trace + PMC passes (tools/profile_all.sh)
#   train      train.py from scratch + evaluation                       soak      long runs of all kernel variants
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r6
mkdir -p $OUT
cd $ROOT
for P in "$@"; do
case $P in
constants) python3 tools/identify_r6.py constants --out $OUT/r06_constants_rule.json > $OUT/r06_constants_rule.txt 2>&1 ;;
all4)      python3 tools/identify_r6.py run --robot laikago --fit laikago_pace laikago_spin laikago_trot laikago_trot0 --minutes 8 --seed 200 \
             --out $OUT/r06_laikago_all4.json --dump-all $OUT/r06_laikago_all4_candidates.jsonl.gz > $OUT/r06_laikago_all4_log.txt 2>&1 ;;
all4p9)    python3 tools/identify_r6.py run --robot laikago --freeze-geometry --fit laikago_pace laikago_spin laikago_trot laikago_trot0 --minutes 8 --seed 800 \
             --out $OUT/r06_laikago_all4_p9.json --dump-all $OUT/r06_laikago_all4_p9_candidates.jsonl.gz > $OUT/r06_laikago_all4_p9_log.txt 2>&1
           python3 tools/identify_r6.py minimal --record $OUT/r06_laikago_all4_p9.json --out $OUT/r06_laikago_minimal_p9.json > $OUT/r06_laikago_minimal_p9.txt 2>&1 ;;
minimal)   python3 tools/identify_r6.py minimal --record $OUT/r06_laikago_all4.json --out $OUT/r06_laikago_minimal.json > $OUT/r06_laikago_minimal.txt 2>&1 ;;
mc)        python3 tools/identify_r6.py run --robot mini_cheetah --fit minicheetah_trot --minutes 4 --seed 300 \
             --out $OUT/r06_mc_identify.json --dump-all $OUT/r06_mc_candidates.jsonl.gz > $OUT/r06_mc_identify_log.txt 2>&1
           python3 tools/identify_r6.py minimal --record $OUT/r06_mc_identify.json --out $OUT/r06_mc_minimal.json > $OUT/r06_mc_minimal.txt 2>&1 ;;
probe)     python3 tools/policy_probe.py --robots 1024 --seeds 1 2 --out $OUT/r06_policy_probe.json > $OUT/r06_policy_probe.txt 2>&1 ;;
drift)     python3 tools/drift_floor_spread.py --runs 12 --out-dir $OUT > $OUT/drift_floor_spread.log 2>&1 ;;
bench)     python3 bench.py --repeats 5 > $OUT/r06_laikago4096_bench.json 2> $OUT/bench.err
           python3 bench.py --gpus 1 --steps 20 --warmup 5 > $OUT/r06_driver20_bench.json 2>> $OUT/bench.err
           for C in minicheetah4096 mixed8192; do python3 bench.py --no-cpu-baseline --repeats 5 --config $C > $OUT/r06_${C}_bench.json 2>> $OUT/bench.err; done
           python3 bench.py --no-cpu-baseline --no-randomizer > $OUT/r06_laikago4096_norand_bench.json 2>> $OUT/bench.err ;;
profiles)  bash tools/profile_all.sh r06 > $OUT/profile_all.log 2>&1 ;;
train)     python3 train.py --iters 24000 --log $OUT/r06_train_laikago_pace_fused.json --save $OUT/r06_laikago_pace_scratch.zip > $OUT/train.log 2>&1
           python3 train.py --eval $OUT/r06_laikago_pace_scratch.zip > $OUT/r06_train_eval.txt 2>&1 ;;
soak)      (python3 tools/soak.py 4096 30000; python3 tools/soak.py 4096 30000 imitation_learning_minicheetah; python3 tools/soak.py 8192 20000; ORR_SOAK_ANCHOR=1 python3 tools/soak.py 4096 30000) 2>&1 | grep -v amdgpu.ids > $OUT/r06_soak.txt ;;
esac
echo "$P done"
done
