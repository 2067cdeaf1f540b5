timeout 900 python -m pytest tests -m gpu -q 2>&1 | grep -v "^RCCL\|^HIP ver\|^ROCm\|^Hostname\|^Librccl" | tail -8 > gpurun_out/r06_k_tests.log; tail -4 gpurun_out/r06_k_tests.log
timeout 400 python bench.py --steps 20 --warmup 5 --detail gpurun_out/r06_k_bench_detail.json > gpurun_out/r06_k_bench_k20.json 2> gpurun_out/r06_k_bench_k20.err
python -c "
import json
t=open('gpurun_out/r06_k_bench_k20.json').read().strip().splitlines()[-1]
print(len(t)); d=json.loads(t); print(d['value'], d['rates']); print(d['roofline']['frac'], d['roofline']['launch_ms'])"
