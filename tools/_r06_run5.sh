python -m pytest tests/test_parity_gpu.py -q -k "training or train or warp or tail or stride2 or conv3d" 2>&1 | tail -3
python tools/bench_train.py --batches 1,4 --out gpurun_out/r06_d_bench_train.json 2>&1 | tail -1
bash tools/profile_train.sh r06_d 1 | sed -n 1,30p
for k in conv_s2 conv_s2_att; do python tools/run_kernel.py $k 1 30 2>/dev/null | tail -1; done
bash tools/pmc_sq.sh conv_s2 1 > /dev/null 2>&1; cp gpurun_out/pmc_sq_conv_s2.txt gpurun_out/r06_d_pmc_sq_conv_s2.txt; cat gpurun_out/pmc_sq_conv_s2.txt | tail -25
