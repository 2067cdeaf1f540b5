# alone times of the small stride-1 layers: product library vs tools/_build/lib_base.so (the previous commit's conv kernel), interleaved
mkdir -p gpurun_out
{
for rep in 1 2; do
for k in conv_low conv_low_att conv_mid_att; do
  echo -n "new  "; timeout 120 python tools/run_kernel.py $k 1 50 2>&1 | tail -1
  echo -n "base "; SS_TOOL_LIB=tools/_build/lib_base.so timeout 120 python tools/run_kernel.py $k 1 50 2>&1 | tail -1
done; done
} > gpurun_out/ab_small.txt
cat gpurun_out/ab_small.txt
