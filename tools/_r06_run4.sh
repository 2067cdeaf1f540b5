for k in wgrad wgrad_stem wgrad_mid wgrad_low wgrad_s2 wgrad_s2_low wgrad_head; do python tools/run_kernel.py $k 1 20 2>/dev/null | tail -2; done
SS_WGRAD_ENGINE=f32 python tools/run_kernel.py wgrad 1 5 2>/dev/null | tail -1
bash tools/pmc_sq.sh wgrad 1 wgrad_bf16s > /dev/null 2>&1; cat gpurun_out/pmc_sq_wgrad.txt
