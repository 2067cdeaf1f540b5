#!/bin/bash
# The measurements a round's profiles/ are refreshed from (GPU box): tests, PMC byte passes, the bench lines, per-kernel timings.
# usage: bash tools/run_round_measurements.sh <tag>      then: python tools/collect_pmc_traffic.py <tag>; copy gpurun_out/<tag>_* to profiles/
tag=${1:-r05_x}
o=gpurun_out
timeout 1200 python -m pytest tests -m gpu -q 2>&1 | grep -v "^RCCL\|^HIP ver\|^ROCm\|^Hostname\|^Librccl" | tail -8 > $o/${tag}_tests.log
cp $o/parity_report.json $o/${tag}_parity_report.json
cp $o/fullsize_strict_f1024_md128_cal.json $o/${tag}_fullsize_strict_f1024_md128_cal.json
cp $o/fullsize_strict_f2048_md192_cal.json $o/${tag}_fullsize_strict_f2048_md192_cal.json
for x in b c; do cp $o/fullsize_strict_f1024_md128_cal_$x.json $o/${tag}_fullsize_strict_f1024_md128_cal_$x.json; done
cp $o/fullsize_strict_f1024_md128.json $o/${tag}_fullsize_strict_f1024_md128.json      # r06: the record with default (uncalibrated) BatchNorm statistics
rm -f $o/pmc_*_b*.json $o/pmc_*_b*.md          # (gpurun_out persists between calls: only THIS run's records are collected)
for k in "stem_gather 1" "gwc 8" "gwc_fused 8" "head_cl 1" "strength 1" "stem_left 1" "ssr 8" "deconv 1" "conv_s2 1"; do timeout 300 bash tools/pmc_bytes.sh $k > /dev/null 2>&1; done
# the DURATION of the batch-8 cost-volume kernel from rocprofv3 beside its PMC bytes, and SQ counters (SQ_INSTS_MFMA,
# matrix pipe busy, wait_inst_any) of the dominant launch, the largest transposed conv and the largest stride-2 conv
( export TMPDIR=/tmp; root=$(pwd); cd /tmp; rm -rf $root/$o/prof_gwc8
  rocprofv3 --kernel-trace --stats --output-format csv -d $root/$o/prof_gwc8 -o run -- python3 $root/tools/run_kernel.py gwc 8 100 > $root/$o/${tag}_gwc_b8_run.txt 2>&1
  cd $root; st=$(find $o/prof_gwc8 -name "*kernel_stats.csv" | head -1); [ -n "$st" ] && cp $st $o/${tag}_gwc_b8_kernel_stats.csv; rm -rf $o/prof_gwc8 )
for k in stem_gather deconv conv_s2 wgrad; do timeout 300 bash tools/pmc_sq.sh $k 1 > /dev/null 2>&1; cp $o/pmc_sq_$k.txt $o/${tag}_pmc_sq_$k.txt; done
# r06: the training step at the size the reference trains at -- ms per step by batch, then a rocprofv3 kernel trace of it
timeout 400 python tools/bench_train.py --batches 1,2,4 --out $o/${tag}_bench_train.json > /dev/null 2>&1
timeout 300 bash tools/profile_train.sh ${tag} 1 > /dev/null 2>&1
timeout 300 bash tools/profile_train.sh ${tag}_b4 4 > /dev/null 2>&1
timeout 300 python tools/train_glue_sources.py > $o/${tag}_glue_sources.txt 2>&1
timeout 600 python bench.py --steps 20 --warmup 5 --detail $o/${tag}_bench_detail_k20.json > $o/${tag}_bench_k20.json 2> /dev/null     # the driver's form
timeout 900 python bench.py --detail $o/${tag}_bench_detail_b1.json > $o/${tag}_bench_b1.json 2> $o/${tag}_bench_b1.err
timeout 300 python bench.py --batch 4 --side-config-steps 0 --train-steps 0 --no-cpu-baseline --no-other-engines --no-side-rooflines --detail /tmp/d4.json > $o/${tag}_bench_b4.json 2>/dev/null
timeout 300 python bench.py --batch 8 --side-config-steps 0 --train-steps 0 --no-cpu-baseline --no-other-engines --no-side-rooflines --detail /tmp/d8.json > $o/${tag}_bench_b8.json 2>/dev/null
timeout 300 python bench.py --height 2048 --width 2048 --maxdisp 192 --no-cpu-baseline --no-other-engines --no-side-rooflines --detail /tmp/d2k.json > $o/${tag}_bench_2048_b1.json 2>/dev/null
timeout 300 bash tools/profile_step.sh ${tag} > /dev/null 2>&1
timeout 600 python tools/strict_report.py f1024_md128_cal f1024_md128_cal_b f1024_md128_cal_c f2048_md192_cal f1024_md128 > $o/${tag}_strict_report.txt 2>/dev/null
timeout 600 python tools/err_stages.py > $o/${tag}_err_stages.txt 2>/dev/null
for k in stem_gather stem_left strength ssr ssr2048 topk upsoft patch gwc_fused head_cl catt4 catt8 deconv deconv5 deconv_att6 deconv_att5 conv_s2 conv_mid conv_low attn wgrad wgrad_stem wgrad_mid wgrad_low wgrad_s2 wgrad_head strength_bwd_ws warp_bwd_smooth warp_bwd; do timeout 120 python tools/run_kernel.py $k 1 20 2>/dev/null | tail -1; done > $o/${tag}_ops_b1.txt
tail -3 $o/${tag}_tests.log
python - <<PY
import json
for f in ("b1", "b4", "b8", "2048_b1"):
    try:
        d = json.load(open("$o/${tag}_bench_%s.json" % f)); print(f, round(d["value"], 1), "pairs/s; steady", d["steady_state"])
    except Exception as e:
        print(f, "failed", e)
PY
