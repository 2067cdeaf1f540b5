echo -n "as is            "; python tools/run_kernel.py wgrad 1 20 2>/dev/null | tail -1
for v in NOSTAGE NOLOAD NOMFMA NOLOAD_NOSTAGE; do echo -n "$v "; SS_TOOL_LIB=tools/_build/lib_wg_$v.so python tools/run_kernel.py wgrad 1 20 2>/dev/null | tail -1; done
