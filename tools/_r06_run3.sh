python -m pytest tests/test_parity_gpu.py -q -k "training or train or warp or tail" 2>&1 | tail -3
python tools/bench_train.py --batches 1,4 --out gpurun_out/r06_d_bench_train.json 2>&1 | tail -1
bash tools/profile_train.sh r06_d 1 | head -36
