#!/bin/bash
# one perf iteration on the GPU box: parity tests, bench (kernel ms), phase cycles.   usage: tools/gpu_iter.sh <tag>
TAG=${1:-it}
python -m pytest tests/test_gpu_parity.py tests/test_gpu_scale.py -x -q > gpurun_out/${TAG}_tests.log 2>&1; tail -3 gpurun_out/${TAG}_tests.log
python bench.py --no-cpu-baseline --steps 2000 > gpurun_out/${TAG}_bench.json 2> gpurun_out/${TAG}_bench.err
python -c "
import json; d=json.load(open('gpurun_out/${TAG}_bench.json')); print('BENCH', d['value'], d['roofline']['kernel_ms'], d['roofline']['kernel_ms_back_to_back'])"
python tools/phase_cycles.py 300 > gpurun_out/${TAG}_phase.txt 2>&1; tail -17 gpurun_out/${TAG}_phase.txt
