timeout 400 python -m pytest tests/test_parity_gpu.py -q -x -k "training or train or batchnorm or bn" 2>&1 | tail -3
timeout 200 python tools/bench_train.py --batches 1,4 --out gpurun_out/r06_l_bench_train.json 2>&1 | tail -1 | cut -c1-700
timeout 200 bash tools/profile_train.sh r06_l 1 | sed -n 1,32p
