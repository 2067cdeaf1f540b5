timeout 300 python -m pytest tests/test_parity_gpu.py -q -k "training or train or weight_gradient" 2>&1 | tail -2
for k in wgrad_s2 wgrad_s2_low; do timeout 60 python tools/run_kernel.py $k 1 20 2>/dev/null | tail -1; SS_WGRAD_COOP=0 timeout 60 python tools/run_kernel.py $k 1 20 2>/dev/null | tail -1; done
timeout 200 python tools/bench_train.py --batches 1,4 --out gpurun_out/r06_v_bench_train.json 2>&1 | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.readline())['bench_train']
for b,r in d.items(): print(b, round(r['ms_per_step'],2), 'ms', round(r['pairs_per_s'],1), 'pairs/s')"
