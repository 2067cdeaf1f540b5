#!/bin/bash
# rocprofv3 kernel trace of a short tools/bench_train.py run -> kernel stats CSV + breakdown by kernel family.
# usage (on the GPU box): tools/profile_train.sh <tag> [batch]     -> gpurun_out/<tag>_train_{kernel_stats.csv,breakdown.txt}
tag=$1; batch=${2:-1}
root=$(pwd); out=$root/gpurun_out
export TMPDIR=/tmp
cd /tmp; rm -rf $out/prof_train
rocprofv3 --kernel-trace --stats --output-format csv -d $out/prof_train -o run -- python3 $root/tools/bench_train.py --batches $batch --steps 3 --warmup 1 --warmup-seconds 0 --no-checks --out /tmp/bt.json > /tmp/bt.log 2>&1
cd $root
st=$(find $out/prof_train -name "*kernel_stats.csv" | head -1); [ -n "$st" ] && cp $st $out/${tag}_train_kernel_stats.csv
rm -rf $out/prof_train
python tools/bench_train.py --kernel-stats $out/${tag}_train_kernel_stats.csv --batches $batch --steps 4 > $out/${tag}_train_breakdown.txt 2>&1
head -40 $out/${tag}_train_breakdown.txt
