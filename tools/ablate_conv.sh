#!/bin/bash
# Timing ablations of the split-bf16 conv: rebuilds conv3d_bf16s.hip with one pipeline stage removed
# (results are WRONG, only the time is meaningful) into tools/_build/lib_abl_*.so.
# usage: tools/ablate_conv.sh   (then on the GPU: SS_TOOL_LIB=tools/_build/lib_abl_X.so python tools/run_conv.py)
set -e
cd "$(dirname "$0")/../semstereo_amd/csrc"
make -s -j8
OUT=../../tools/_build
mkdir -p $OUT
VARIANTS=(SPLIT A IN B MAX RES STORE "RES -DSS_ABL_STORE" "SPLIT -DSS_ABL_IN" "SPLIT -DSS_ABL_IN -DSS_ABL_A"
          "SPLIT -DSS_ABL_IN -DSS_ABL_A -DSS_ABL_B" "SPLIT -DSS_ABL_IN -DSS_ABL_A -DSS_ABL_B -DSS_ABL_RES -DSS_ABL_STORE")
objs=$(ls *.o | grep -vx conv3d_bf16s.o)
build_one() {
  local v="$1" name
  name=$(echo "$v" | sed 's/ -DSS_ABL_/_/g')
  hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -DSS_ABL_$v -c conv3d_bf16s.hip -o /tmp/abl_$name.o
  hipcc --offload-arch=gfx950 -shared -fPIC -o $OUT/lib_abl_$name.so /tmp/abl_$name.o $objs
  echo built $OUT/lib_abl_$name.so
}
n=0
for v in "${VARIANTS[@]}"; do
  build_one "$v" &
  n=$((n + 1)); if [ $((n % 6)) -eq 0 ]; then wait; fi
done
wait
# occupancy variant of the fp16 form: compiled for 3 workgroups per CU (168 VGPRs)
hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -DSS_F16_WGS=3 -c conv3d_bf16s.hip -o /tmp/abl_wgs3.o
hipcc --offload-arch=gfx950 -shared -fPIC -o $OUT/lib_wgs3.so /tmp/abl_wgs3.o $objs
[ -n "$SS_ABL_CONV_ONLY" ] && exit 0
# the split-bf16 transposed conv: main loop / skip projection / output stores compiled out
# (then: SS_TOOL_LIB=tools/_build/lib_dabl_X.so python tools/run_deconv.py bf16x6)
for v in MAIN SKIP STORE "MAIN -DSS_ABL_D_SKIP"; do
  name=$(echo "$v" | sed 's/ -DSS_ABL_D_/_/g')
  hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -DSS_ABL_D_$v -c deconv3d_bf16s.hip -o /tmp/dabl_$name.o
  objs=$(ls *.o | grep -vx deconv3d_bf16s.o)
  hipcc --offload-arch=gfx950 -shared -fPIC -o $OUT/lib_dabl_$name.so /tmp/dabl_$name.o $objs
  echo built $OUT/lib_dabl_$name.so
done
