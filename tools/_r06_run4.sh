timeout 300 python -m pytest tests/test_parity_gpu.py -q -k "training or train" 2>&1 | tail -2
timeout 200 python tools/bench_train.py --batches 1,2,4 --out gpurun_out/r06_o_bench_train.json 2>&1 | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.readline())['bench_train']
for b,r in d.items(): print(b, round(r['ms_per_step'],2), 'ms', round(r['pairs_per_s'],1), 'pairs/s', round(r['peak_allocated_gb'],1), 'GB', r['rerun_max_relative_gradient_difference'])"
timeout 200 bash tools/profile_train.sh r06_o 1 | grep -E "wgrad_k1|whole run"
