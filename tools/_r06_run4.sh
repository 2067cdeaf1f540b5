python -m pytest tests/test_parity_gpu.py -q -k "training or train" 2>&1 | tail -2
for k in wgrad wgrad_stem wgrad_mid wgrad_low wgrad_s2 wgrad_s2_low wgrad_head; do python tools/run_kernel.py $k 1 20 2>/dev/null | tail -1; done
python tools/bench_train.py --batches 1,4 --out gpurun_out/r06_g_bench_train.json 2>&1 | tail -1
bash tools/pmc_sq.sh wgrad 1 wgrad_bf16s > /dev/null 2>&1; tail -22 gpurun_out/pmc_sq_wgrad.txt
