for rep in 1 2 3; do
  for k in conv_s2 conv_s2_att; do
  echo -n "natural "; python tools/run_kernel.py $k 1 50 2>&1 | tail -1
  echo -n "deint_w "; SS_TOOL_LIB=tools/_build/lib_deintw.so python tools/run_kernel.py $k 1 50 2>&1 | tail -1
  done
done
