#!/bin/bash
# GPU box: shader clock and socket power while bench.py runs a configuration (development aid).  usage: tools/diag/clock_sample.sh <config> [steps]
CFG=${1:-laikago4096}; STEPS=${2:-40000}
python3 bench.py --config $CFG --steps $STEPS --warmup 2000 --no-cpu-baseline > /tmp/clock_bench_$CFG.json 2>/dev/null &
BP=$!
sleep 6
for i in 1 2 3 4 5 6; do
  rocm-smi --showclocks --showpower 2>/dev/null | grep -E "sclk|Power" | tr '\n' ' '; echo
  sleep 0.7
done
wait $BP
python3 -c "
import json; d=json.loads(open('/tmp/clock_bench_$CFG.json').read().strip().splitlines()[-1]); print('$CFG', round(d['value']/1e6,2), 'M', d['roofline']['kernel_ms'])"
