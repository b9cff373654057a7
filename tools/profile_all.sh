#!/bin/bash
# GPU box: all round profiles in one call, keeping only the summaries (the raw rocprofv3 CSVs of five configurations exceed what gpurun
# copies back).  usage: tools/profile_all.sh <round tag, e.g. r03>      -> gpurun_out/summ/<round>_*
R=${1:-r03}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
mkdir -p $ROOT/gpurun_out/summ
run() {   # tag, config, extra bench flags, env assignment
  ( export $4; $ROOT/tools/profile_gpu.sh ${R}_$1 $2 "$3" > $ROOT/gpurun_out/summ/${R}_$1_profile.log 2>&1 )
  python3 $ROOT/tools/summarize_profile.py ${R}_$1 ${R}_$1 > /dev/null 2>&1
  cp $ROOT/profiles/${R}_$1_kernel_stats.csv $ROOT/profiles/${R}_$1_pmc_summary.json $ROOT/gpurun_out/summ/ 2>/dev/null
  grep -l "failed" $ROOT/gpurun_out/summ/${R}_$1_profile.log && cp $ROOT/gpurun_out/${R}_$1/*.log $ROOT/gpurun_out/summ/ 2>/dev/null   # keep the logs of a failed pass
  rm -rf $ROOT/gpurun_out/${R}_$1
}
run laikago4096 laikago4096 "" X=1
run mixed8192 mixed8192 "" X=1
run mixed8192_wpe1 mixed8192 "" ORR_STEP_WAVES_PER_EU=1
run laikago4096_norand laikago4096 "--no-randomizer" X=1
run minicheetah4096 minicheetah4096 "" X=1
cd $ROOT
python3 tools/wave_timeline.py 50 2>/dev/null | grep -v amdgpu.ids > gpurun_out/summ/${R}_wave_timeline.txt
python3 tools/phase_cycles.py 800 2>/dev/null | grep -v amdgpu.ids > gpurun_out/summ/${R}_phase_cycles.txt
python3 tools/wave_phases.py 40 2>/dev/null | grep -v amdgpu.ids > gpurun_out/summ/${R}_wave_phases.txt
ls gpurun_out/summ | wc -l
