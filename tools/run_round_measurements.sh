#!/bin/bash
# The measurements a round's profiles/ are refreshed from (GPU box): tests, PMC byte passes, the bench lines, per-kernel timings.
# usage: bash tools/run_round_measurements.sh <tag>      then: python tools/collect_pmc_traffic.py <tag>; copy gpurun_out/<tag>_* to profiles/
tag=${1:-r03_x}
python -m pytest tests -m gpu -q 2>&1 | tail -8 > gpurun_out/${tag}_tests.log
cp gpurun_out/parity_report.json gpurun_out/${tag}_parity_report.json
for k in "stem 1" "gwc 8" "gwc_fused 8" "head_cl 1" "warp 1" "strength 1" "stem_left 1" "ssr 8"; do bash tools/pmc_bytes.sh $k > /dev/null 2>&1; done
python bench.py > gpurun_out/${tag}_bench_b1.json 2> gpurun_out/${tag}_bench_b1.err
python bench.py --batch 4 --no-cpu-baseline --no-other-engines > gpurun_out/${tag}_bench_b4.json 2>/dev/null
python bench.py --batch 8 --no-cpu-baseline --no-other-engines > gpurun_out/${tag}_bench_b8.json 2>/dev/null
python bench.py --height 2048 --width 2048 --maxdisp 192 --no-cpu-baseline --no-other-engines > gpurun_out/${tag}_bench_2048_b1.json 2>/dev/null
bash tools/profile_step.sh ${tag} > /dev/null 2>&1
for k in warp strength stem_left ssr ssr2048 topk upsoft patch gwc_fused head_cl catt4 catt8 deconv conv_s2 conv_mid conv_low attn; do python tools/run_kernel.py $k 1 20 2>/dev/null | tail -1; done > gpurun_out/${tag}_ops_b1.txt
tail -3 gpurun_out/${tag}_tests.log
