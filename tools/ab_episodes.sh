ROOT=$(pwd)
for W in old new; do
  if [ $W = old ]; then export ORR_LIB_PATH=$ROOT/openroborl_amd/lib_ab_old.so; else unset ORR_LIB_PATH; fi
  python3 bench.py --steps 1500 --warmup 100 --no-cpu-baseline 2>/dev/null | python3 -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$W', round(d['value']/1e6,3), 'kernel', round(d['roofline']['kernel_ms'],4), 'episodes', d['config']['episodes_gathered'])"
done
