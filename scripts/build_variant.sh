#!/bin/bash
# Alternate build of libcc_hip.so with extra compiler flags on ONE .hip file (A/B and timing-only builds; loaded through
# CC_LIB_PATH). Usage: scripts/build_variant.sh NAME cc_rig.hip -DCC_RIG_TIMING   ->  scripts/ablate_build/libcc_NAME.so
set -e
name=$1; file=$2; shift 2
cd "$(dirname "$0")/../camera_calibrator_amd/csrc"
mkdir -p ../../scripts/ablate_build
make -s
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wno-unused-result -Wno-unused-value "$@" -c $file -o /tmp/cc_variant_$name.o
objs=""
for o in cc_intrinsics.o cc_intrinsics_persist.o cc_rig.o cc_zhang.o cc_points.o cc_common.o cc_comm.o data_generator.o rig_scenario.o geometry.o; do
  if [ "$o" = "${file%.hip}.o" ]; then objs="$objs /tmp/cc_variant_$name.o"; else objs="$objs $o"; fi
done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../../scripts/ablate_build/libcc_$name.so $objs -ldl -pthread -Wl,-rpath,/opt/rocm/lib
echo scripts/ablate_build/libcc_$name.so
