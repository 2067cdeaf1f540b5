bash tools/ab_kernel.sh conv_s2 > gpurun_out/r06_e_ab_conv_s2_deint.txt 2>&1
bash tools/ab_kernel.sh conv_s2_att >> gpurun_out/r06_e_ab_conv_s2_deint.txt 2>&1
cat gpurun_out/r06_e_ab_conv_s2_deint.txt
bash tools/ab_lib.sh 2 2>&1 | tee gpurun_out/r06_e_ab_step_deint.txt
