#!/bin/bash
# Interleaved A/B of the product library against tools/_build/lib_base.so on the bench step (same box): ROUNDS x (new, base) runs,
# 300 timed steps each, no side legs.  usage: tools/ab_lib.sh [rounds] [extra bench args...]
rounds=${1:-3}; shift 1 2>/dev/null
for r in $(seq 1 $rounds); do
  for v in new base; do
    lib=""; [ $v = base ] && lib=tools/_build/lib_base.so
    SS_TOOL_LIB=$lib python tools/bench_with_lib.py --steps 300 --warmup 20 --steady-seconds 0 --power-seconds 0 --no-cpu-baseline --no-other-engines --no-side-rooflines --side-config-steps 0 --train-steps 0 --parity-pairs 0 --detail /tmp/ab_detail.json "$@" 2>/dev/null \
      | python -c "import json,sys; d=json.loads(sys.stdin.readline()); print('$v', round(d['value'],1), 'pairs/s pipelined;', round(d['rates']['single_stream_pairs_per_s'],1), 'single stream')"
  done
done
