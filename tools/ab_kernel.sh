# usage: tools/ab_kernel.sh <run_kernel.py kernel> [batch] -- alone times, product library vs tools/_build/lib_base.so, interleaved x3
k=$1; b=${2:-1}
for rep in 1 2 3; do
  echo -n "new  "; timeout 120 python tools/run_kernel.py $k $b 50 2>&1 | tail -1
  echo -n "base "; SS_TOOL_LIB=tools/_build/lib_base.so timeout 120 python tools/run_kernel.py $k $b 50 2>&1 | tail -1
done
