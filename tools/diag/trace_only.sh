#!/bin/bash
# GPU box: rocprofv3 kernel-trace average of the step kernel for two configurations (the default bench.py run of each)
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; cd /tmp && export TMPDIR=/tmp
for CFG in laikago4096 mixed8192; do
  rm -rf /tmp/tr_$CFG
  timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/tr_$CFG -- python3 $ROOT/bench.py --config $CFG --no-cpu-baseline > /tmp/tr_$CFG.log 2>&1
  F=$(find /tmp/tr_$CFG -name "*kernel_stats.csv" | head -1)
  grep orr_step_kernel $F | head -1 | awk -F, -v n=$CFG '{print n, $(NF-4)/1000 " us avg over", $(NF-6), "launches"}'
done
