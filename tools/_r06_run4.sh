timeout 300 python -m pytest tests/test_parity_gpu.py -q -k "training or train" 2>&1 | tail -2
for k in wgrad_head wgrad; do timeout 60 python tools/run_kernel.py $k 1 20 2>/dev/null | tail -1; done
timeout 200 python tools/bench_train.py --batches 1,4 --out gpurun_out/r06_m_bench_train.json 2>&1 | tail -1 | cut -c1-400
