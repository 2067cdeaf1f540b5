python -m pytest tests/test_parity_gpu.py -q -k "training or train" 2>&1 | tail -8
python tools/bench_train.py --batches 1,4 --out gpurun_out/r06_b_bench_train.json 2>&1 | tail -1
