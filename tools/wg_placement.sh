export SS_TOOL_LIB=tools/_build/lib_timing.so
mkdir -p gpurun_out
{
STRIDE=1 timeout 120 python tools/wg_placement.py 128 128 6 64 64
STRIDE=1 timeout 120 python tools/wg_placement.py 128 128 8 32 32
STRIDE=1 timeout 120 python tools/wg_placement.py 64 64 16 64 64
STRIDE=2 timeout 120 python tools/wg_placement.py 64 128 16 64 64
STRIDE=2 timeout 120 python tools/wg_placement.py 32 64 32 128 128
STRIDE=1 timeout 120 python tools/wg_placement.py 32 32 24 256 256
} > gpurun_out/wg_placement.txt 2>&1
cat gpurun_out/wg_placement.txt
