python -m pytest tests/test_parity_gpu.py -q -k "training or train" 2>&1 | tail -2
for k in wgrad wgrad_stem wgrad_mid wgrad_low wgrad_head; do python tools/run_kernel.py $k 1 20 2>/dev/null | tail -1; done
echo per-wave form:; for k in wgrad wgrad_low; do SS_WGRAD_COOP=0 python tools/run_kernel.py $k 1 20 2>/dev/null | tail -1; done
python tools/bench_train.py --batches 1,4 --out gpurun_out/r06_i_bench_train.json 2>&1 | tail -1 | cut -c1-900
