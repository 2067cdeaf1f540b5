#!/bin/bash
# tools/_build/lib_timing.so: the product library with the tiled conv kernel replaced by its instrumented build
# (tools/conv_timing.hip).  Then on the GPU box: SS_TOOL_LIB=tools/_build/lib_timing.so python tools/wg_phases.py
set -e
cd "$(dirname "$0")/.."
make -s -j8 -C semstereo_amd/csrc
mkdir -p tools/_build
hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -Isemstereo_amd/csrc -c tools/conv_timing.hip -o tools/_build/conv_timing.o
objs=$(ls semstereo_amd/csrc/*.o | grep -v "/conv3d_bf16s.o")
hipcc --offload-arch=gfx950 -shared -fPIC -o tools/_build/lib_timing.so tools/_build/conv_timing.o $objs
echo built tools/_build/lib_timing.so
