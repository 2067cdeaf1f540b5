#!/bin/bash
# Socket power and shader clock (rocm-smi) while ONE kernel of the path runs back to back for ~5 s: is that kernel alone at the power cap?
# usage (GPU box): tools/kernel_power.sh <kernel> [<kernel> ...]   -> gpurun_out/kernel_power.txt  (the 4 highest-power samples of each run)
mkdir -p gpurun_out
out=gpurun_out/kernel_power.txt; [ -z "$KP_APPEND" ] && : > $out      # KP_APPEND=1: keep earlier records (variant builds through SS_TOOL_LIB)
for k in "$@"; do
  SS_WARM_MS=5000 python3 tools/run_kernel.py $k 1 50 > /tmp/kp_$k.txt 2>&1 &
  pid=$!
  : > /tmp/kp_samples.txt
  while kill -0 $pid 2>/dev/null; do
    rocm-smi --showpower --showclocks 2>/dev/null | grep -E "Socket Graphics Package Power|sclk" | sed -E 's/.*: //' | tr -d '()' | tr '\n' ' ' >> /tmp/kp_samples.txt
    echo >> /tmp/kp_samples.txt
  done
  wait $pid
  echo "$k${SS_TOOL_LIB:+ [$(basename $SS_TOOL_LIB .so)]}: $(tail -1 /tmp/kp_$k.txt)" | tee -a $out
  awk '{print $NF, $0}' /tmp/kp_samples.txt | sort -rn | head -4 | awk '{printf "    sclk %s  power %s W\n", $2, $NF}' | tee -a $out
done
