mkdir -p gpurun_out
python -m pytest tests -m gpu -q -x 2>&1 | grep -v "^RCCL\|^HIP ver\|^ROCm\|^Hostname\|^Librccl" | tail -15 > gpurun_out/r06_a_tests.log
tail -5 gpurun_out/r06_a_tests.log
python tools/bench_train.py --batches 1,2,4 --out gpurun_out/r06_a_bench_train.json 2>&1 | tail -5
( export TMPDIR=/tmp; root=$(pwd); cd /tmp; rm -rf $root/gpurun_out/prof_train
  rocprofv3 --kernel-trace --stats --output-format csv -d $root/gpurun_out/prof_train -o run -- python3 $root/tools/bench_train.py --batches 1 --steps 3 --warmup 1 --no-checks --out /tmp/bt.json > /tmp/bt.log 2>&1
  cd $root; st=$(find gpurun_out/prof_train -name "*kernel_stats.csv" | head -1); [ -n "$st" ] && cp $st gpurun_out/r06_a_train_kernel_stats.csv; rm -rf gpurun_out/prof_train )
python tools/bench_train.py --kernel-stats gpurun_out/r06_a_train_kernel_stats.csv --batches 1 --steps 3 > gpurun_out/r06_a_train_breakdown.txt 2>&1
head -30 gpurun_out/r06_a_train_breakdown.txt
python bench.py --steps 20 --warmup 5 --detail gpurun_out/r06_a_bench_detail.json > gpurun_out/r06_a_bench_k20.json 2> gpurun_out/r06_a_bench_k20.err
python -c "
import json
t=open('gpurun_out/r06_a_bench_k20.json').read().strip().splitlines()[-1]
print(len(t)); d=json.loads(t); print(d['value'], d['rates']); print(d.get('parity_seeded_pairs')); print(d['parity'].get('fixture_uncal'))"
cp gpurun_out/fullsize_strict_f1024_md128.json gpurun_out/r06_a_fullsize_strict_f1024_md128.json
