#!/bin/bash
# The measurements a round's profiles/ are refreshed from (GPU box): tests, PMC byte passes, the bench lines, per-kernel timings.
# usage: bash tools/run_round_measurements.sh <tag>      then: python tools/collect_pmc_traffic.py <tag>; copy gpurun_out/<tag>_* to profiles/
tag=${1:-r04_x}
o=gpurun_out
python -m pytest tests -m gpu -q 2>&1 | grep -v "^RCCL\|^HIP ver\|^ROCm\|^Hostname\|^Librccl" | tail -8 > $o/${tag}_tests.log
cp $o/parity_report.json $o/${tag}_parity_report.json
cp $o/fullsize_strict_f1024_md128_cal.json $o/${tag}_fullsize_strict_f1024_md128_cal.json
cp $o/fullsize_strict_f2048_md192_cal.json $o/${tag}_fullsize_strict_f2048_md192_cal.json
for k in "stem 1" "gwc 8" "gwc_fused 8" "head_cl 1" "warp 1" "strength 1" "stem_left 1" "ssr 8"; do bash tools/pmc_bytes.sh $k > /dev/null 2>&1; done
python bench.py --detail $o/${tag}_bench_detail_b1.json > $o/${tag}_bench_b1.json 2> $o/${tag}_bench_b1.err
python bench.py --batch 4 --no-cpu-baseline --no-other-engines --no-side-rooflines --detail /tmp/d4.json > $o/${tag}_bench_b4.json 2>/dev/null
python bench.py --batch 8 --no-cpu-baseline --no-other-engines --no-side-rooflines --detail /tmp/d8.json > $o/${tag}_bench_b8.json 2>/dev/null
python bench.py --height 2048 --width 2048 --maxdisp 192 --no-cpu-baseline --no-other-engines --no-side-rooflines --detail /tmp/d2k.json > $o/${tag}_bench_2048_b1.json 2>/dev/null
bash tools/profile_step.sh ${tag} > /dev/null 2>&1
python tools/strict_report.py f1024_md128_cal f2048_md192_cal > $o/${tag}_strict_report.txt 2>/dev/null
python tools/err_stages.py > $o/${tag}_err_stages.txt 2>/dev/null
for k in warp strength stem_left ssr ssr2048 topk upsoft patch gwc_fused head_cl catt4 catt8 deconv conv_s2 conv_mid conv_low attn; do python tools/run_kernel.py $k 1 20 2>/dev/null | tail -1; done > $o/${tag}_ops_b1.txt
tail -3 $o/${tag}_tests.log
python - <<PY
import json
for f in ("b1", "b4", "b8", "2048_b1"):
    try:
        d = json.load(open("$o/${tag}_bench_%s.json" % f)); print(f, round(d["value"], 1), "pairs/s; steady", d["steady_state"])
    except Exception as e:
        print(f, "failed", e)
PY
