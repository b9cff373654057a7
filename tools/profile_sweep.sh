set -u
for C in laikago4096 minicheetah4096 mixed8192; do
  bash tools/profile_gpu.sh r02_$C $C > gpurun_out/sweep_$C.log 2>&1
  python bench.py --config $C > gpurun_out/r02_${C}_bench.json 2> gpurun_out/r02_${C}_bench.err
  echo "$C done"; tail -c 300 gpurun_out/r02_${C}_bench.json | head -c 200; echo
done
python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r02_driver20_bench.json 2> gpurun_out/r02_driver20_bench.err
python tools/phase_cycles.py 800 > gpurun_out/r02_phase_cycles.txt 2>&1
python tools/wave_timeline.py 60 > gpurun_out/r02_wave_timeline.txt 2>&1
echo sweep finished
