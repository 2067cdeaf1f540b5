python -m pytest tests -m gpu -q 2>&1 | tail -8 > gpurun_out/r03_f_tests.log
cp gpurun_out/parity_report.json gpurun_out/r03_f_parity_report.json
for k in "stem 1" "gwc 8" "gwc_fused 8" "head_cl 1" "warp 1" "strength 1" "stem_left 1" "ssr 8"; do bash tools/pmc_bytes.sh $k > /dev/null 2>&1; done
python bench.py > gpurun_out/r03_f_bench_b1.json 2> gpurun_out/r03_f_bench_b1.err
for k in warp strength stem_left ssr topk upsoft patch gwc_fused head_cl catt4 catt8 deconv conv_s2 conv_mid conv_low attn; do python tools/run_kernel.py $k 1 20 2>/dev/null | tail -1; done > gpurun_out/r03_f_ops_b1.txt
tail -3 gpurun_out/r03_f_tests.log; cat gpurun_out/r03_f_ops_b1.txt
