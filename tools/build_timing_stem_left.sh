#!/bin/bash
# tools/_build/lib_timing_sl.so: the product library with stem_left.hip replaced by its instrumented build (tools/stem_left_timing.hip)
set -e
cd "$(dirname "$0")/.."
make -s -j8 -C semstereo_amd/csrc
mkdir -p tools/_build
hipcc -O3 -std=c++17 -fPIC -fno-slp-vectorize --offload-arch=gfx950 -Isemstereo_amd/csrc -Iinclude -c tools/stem_left_timing.hip -o tools/_build/stem_left_timing.o
objs=$(ls semstereo_amd/csrc/*.o | grep -v "/stem_left.o")
hipcc --offload-arch=gfx950 -shared -fPIC -o tools/_build/lib_timing_sl.so tools/_build/stem_left_timing.o $objs
echo built tools/_build/lib_timing_sl.so
