#!/bin/bash
# tools/_build/lib_<name>.so: the product library with ONE source file recompiled with extra -D flags (compile-time sweeps of a
# kernel's constants).  usage: tools/build_variant.sh <name> <file.hip> [-DFOO=1 ...];  then SS_TOOL_LIB=tools/_build/lib_<name>.so python tools/run_kernel.py ...
set -e
cd "$(dirname "$0")/.."
name=$1; src=$2; shift 2
make -s -j8 -C semstereo_amd/csrc
mkdir -p tools/_build
base=$(basename "$src" .hip)
hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -Isemstereo_amd/csrc -Iinclude "$@" -c "semstereo_amd/csrc/$base.hip" -o "tools/_build/${base}_$name.o"
objs=$(ls semstereo_amd/csrc/*.o | grep -v "/$base.o")
hipcc --offload-arch=gfx950 -shared -fPIC -o "tools/_build/lib_$name.so" "tools/_build/${base}_$name.o" $objs
echo "built tools/_build/lib_$name.so"
