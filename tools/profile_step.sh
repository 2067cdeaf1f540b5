#!/bin/bash
# rocprofv3 kernel trace of a short bench.py run -> kernel stats CSV + the per-step breakdown (tools/step_breakdown.py).
# usage (on the GPU box): tools/profile_step.sh <tag> [extra bench.py args]     -> gpurun_out/<tag>_{bench.json,kernel_stats.csv,step_breakdown.txt}
set -u
tag=$1; shift
root=$(pwd)
out=$root/gpurun_out
mkdir -p $out/prof_$tag
export TMPDIR=/tmp
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv rocpd -d $out/prof_$tag -o run -- python3 $root/bench.py --steps 12 --warmup 3 --no-cpu-baseline --no-other-engines --steady-seconds 0 --power-seconds 0 --no-side-rooflines --side-config-steps 0 --train-steps 0 --streams 1 "$@" > $out/${tag}_bench.json 2> $out/${tag}_prof.err
cd $root
db=$(find $out/prof_$tag -name "*.db" | head -1)
stats=$(find $out/prof_$tag -name "*kernel_stats.csv" | head -1)
[ -n "$stats" ] && cp $stats $out/${tag}_kernel_stats.csv
[ -n "$db" ] && python3 tools/step_breakdown.py $db "${SS_MARKER:-topk_regress_kernel<2>}" > $out/${tag}_step_breakdown.txt 2>&1
[ -n "$db" ] && python3 tools/step_timeline.py $db "${SS_MARKER:-topk_regress_kernel<2>}" > $out/${tag}_step_timeline.txt 2>&1
ls -R $out/prof_$tag | head -20 >> $out/${tag}_prof.err; rm -rf $out/prof_$tag
head -40 $out/${tag}_step_breakdown.txt
