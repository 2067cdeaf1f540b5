#!/bin/bash
# Socket power and shader clock (rocm-smi) while a command runs: prints the command's last line and the 4 highest-power samples.
# usage (GPU box): tools/power_of.sh <command ...>
"$@" > /tmp/po_out.txt 2>&1 &
pid=$!
: > /tmp/po_samples.txt
while kill -0 $pid 2>/dev/null; do
  rocm-smi --showpower --showclocks 2>/dev/null | grep -E "Socket Graphics Package Power|sclk" | sed -E 's/.*: //' | tr -d '()' | tr '\n' ' ' >> /tmp/po_samples.txt
  echo >> /tmp/po_samples.txt
done
wait $pid
echo "$(tail -1 /tmp/po_out.txt)"
awk 'NF>=2 {print $NF, $0}' /tmp/po_samples.txt | sort -rn | head -4 | awk '{printf "    sclk %s  power %s W\n", $2, $NF}'
