#!/bin/bash
# Interleaved A/B of one environment switch on the bench step (same box, same process sequence): ROUNDS x (A, B) runs of bench.py
# with 300 timed steps each, no side legs.  usage: tools/ab_env.sh VAR A B [rounds] [extra bench args...]
var=$1; a=$2; b=$3; rounds=${4:-3}; shift 4 2>/dev/null || shift $#
for r in $(seq 1 $rounds); do
  for v in $a $b; do
    env $var=$v python bench.py --steps 300 --warmup 20 --steady-seconds 0 --power-seconds 0 --no-cpu-baseline --no-other-engines --no-side-rooflines --side-config-steps 0 --train-steps 0 --detail /tmp/ab_detail.json "$@" 2>/dev/null \
      | python -c "import json,sys; d=json.loads(sys.stdin.readline()); print('$var=$v', round(d['value'],1), 'pairs/s', round(d['ms_per_step'],4), 'ms')"
  done
done
