# usage: bash tools/ab_defs.sh "<defs A>" "<defs B>" [rounds]   -- interleaved A/B of compile-time variants on one box
A="$1"; B="$2"; R=${3:-3}
for i in $(seq 1 $R); do for D in "$A" "$B"; do export ORR_EXTRA_DEFS="$D"; python -c "from openroborl_amd import _lib; _lib.build(force=True)" >/dev/null 2>&1; echo "[$D]: $(timeout -k 10 200 python bench.py --steps 300 --warmup 100 --no-cpu-baseline 2>/dev/null | tail -1 | cut -c52-90)"; done; done
