#!/bin/bash
# GPU box: the round-5 measurement set, part by part (each part fits one gpurun call).  Everything lands under gpurun_out/r5/ and is
# copied into profiles/r05_* (and tests/golden/drift_factors.json) afterwards.   usage: tools/r5_records.sh <part> [...]
#   identify   the joint, held-out identification of the Laikago table (protocol: tools/laikago_identify.py docstring) - ONCE; the
#              shipped table is that run's chosen candidate (profiles/r05_laikago_identify.json)
#   ablate     its fit-set-only ablation                                   replicate  the protocol again, seed 1 (chooses nothing)
#   survey     fit-set-only run with every candidate dumped + statistics   probe      all five shipped policies on the shipped tables
#   drift      float32 drift factors on the current sources               bench      bench lines
#   profiles   rocprofv3 kernel trace + PMC passes (tools/profile_all.sh)  train      train.py from scratch + evaluation
#   soak       long runs of all kernel variants                            mc         mini-cheetah contact consistency check
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r5
mkdir -p $OUT
cd $ROOT
for P in "$@"; do
case $P in
identify)  python3 tools/laikago_identify.py --minutes 12 --robots 128 --out $OUT/r05_laikago_identify.json > $OUT/r05_laikago_identify_log.txt 2>&1 ;;
ablate)    python3 tools/laikago_identify.py --robots 128 --ablate profiles/r05_laikago_identify.json --out $OUT/r05_laikago_identify_ablation.json > $OUT/r05_laikago_identify_ablation.txt 2>&1 ;;
replicate) python3 tools/laikago_identify.py --minutes 12 --robots 128 --seed 1 --out $OUT/r05_laikago_identify_seed1.json > $OUT/r05_laikago_identify_seed1_log.txt 2>&1 ;;
survey)    python3 tools/laikago_identify.py --minutes 12 --robots 128 --seed 2 --no-holdout --dump-all $OUT/r05_laikago_survey_seed2.jsonl --out $OUT/r05_laikago_survey_seed2.json > $OUT/r05_laikago_survey_seed2_log.txt 2>&1
           python3 tools/laikago_accept_stats.py $OUT/r05_laikago_survey_seed2.jsonl > $OUT/r05_laikago_accept_stats.txt ;;
probe)     python3 tools/policy_probe.py --robots 1024 --seeds 1 2 --out $OUT/r05_policy_probe.json > $OUT/r05_policy_probe.txt 2>&1 ;;
drift)     python3 tools/drift_floor_spread.py --runs 12 --out-dir $OUT > $OUT/drift_floor_spread.log 2>&1 ;;
bench)     python3 bench.py > $OUT/r05_laikago4096_bench.json 2> $OUT/bench.err
           python3 bench.py --gpus 1 --steps 20 --warmup 5 > $OUT/r05_driver20_bench.json 2>> $OUT/bench.err
           for C in minicheetah4096 mixed8192; do python3 bench.py --no-cpu-baseline --config $C > $OUT/r05_${C}_bench.json 2>> $OUT/bench.err; done
           python3 bench.py --no-cpu-baseline --no-randomizer > $OUT/r05_laikago4096_norand_bench.json 2>> $OUT/bench.err ;;
profiles)  bash tools/profile_all.sh r05 > $OUT/profile_all.log 2>&1 ;;
train)     python3 train.py --iters 24000 --log $OUT/r05_train_laikago_pace_fused.json --save $OUT/r05_laikago_pace_scratch.zip > $OUT/train.log 2>&1
           python3 train.py --eval $OUT/r05_laikago_pace_scratch.zip > $OUT/r05_train_eval.txt 2>&1 ;;
soak)      (python3 tools/soak.py 4096 30000; python3 tools/soak.py 4096 30000 imitation_learning_minicheetah; python3 tools/soak.py 8192 20000; ORR_SOAK_ANCHOR=1 python3 tools/soak.py 4096 30000) 2>&1 | grep -v amdgpu.ids > $OUT/r05_soak.txt ;;
mc)        python3 tools/mc_contact_check.py --out $OUT/r05_mc_contact_check.json 2>&1 | grep -v amdgpu > $OUT/r05_mc_contact_check.txt ;;
esac
echo "$P done"
done
