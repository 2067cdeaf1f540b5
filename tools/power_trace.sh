#!/bin/bash
# Power / clock samples (rocm-smi, twice a second) while the bench step runs ~15 s on the pipelined lanes: is the step at the power cap?
mkdir -p gpurun_out
( for i in $(seq 1 40); do rocm-smi --showpower --showclocks --showuse 2>/dev/null | grep -E "Power|sclk|GPU use" | tr -s ' ' | tr '\n' ';'; echo; sleep 0.5; done ) > gpurun_out/power_trace.txt &
smi=$!
python bench.py --steps 4000 --warmup 20 --steady-seconds 0 --power-seconds 0 --no-cpu-baseline --no-other-engines --no-side-rooflines --parity-pairs 0 --pipelined-only --detail /tmp/pd.json 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.readline()); print(round(d['value'],1), 'pairs/s')"
wait $smi
rocm-smi --showmaxpower 2>/dev/null | grep -i "power" | head -3
sed -n 1,40p gpurun_out/power_trace.txt | cut -c1-220
