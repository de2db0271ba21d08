#!/bin/bash
# Alternate build of libcc_hip.so from a SCRATCH COPY of the sources: optional patches from scripts/variants/ (timing marks, ablation
# returns, exact-arithmetic forms -- none of that text lives in the product translation units) and extra compiler flags on ONE .hip
# file. The result is loaded through CC_LIB_PATH.
#   scripts/build_variant.sh NAME FILE.hip [--patch timing] [--patch exact_arith] [-DFLAG ...]  ->  scripts/ablate_build/libcc_NAME.so
#   e.g. scripts/build_variant.sh rigtime cc_rig.hip --patch timing -DCC_RIG_TIMING
# The patches were cut against the sources of the commit that added them (round 6); after an edit of a patched region regenerate
# them (apply with fuzz, fix, `diff -u -r` the product sources against the patched copy).
set -e
name=$1; file=$2; shift 2
patches=(); flags=()
while [ $# -gt 0 ]; do
  if [ "$1" = "--patch" ]; then patches+=("$2"); shift 2; else flags+=("$1"); shift; fi
done
root="$(cd "$(dirname "$0")/.." && pwd)"
src=$root/camera_calibrator_amd/csrc
out=$root/scripts/ablate_build
mkdir -p $out
make -s -C $src
work=$out/src_$name
rm -rf $work && mkdir -p $work/camera_calibrator_amd $work/include
cp $src/*.hip $src/*.hpp $src/*.cpp $src/*.hh $work/camera_calibrator_amd/ 2>/dev/null || true
mkdir -p $work/camera_calibrator_amd/csrc && mv $work/camera_calibrator_amd/*.* $work/camera_calibrator_amd/csrc/
cp $root/include/*.h $work/include/
for p in "${patches[@]}"; do (cd $work/camera_calibrator_amd && patch -s -p0 < $root/scripts/variants/$p.patch); done
(cd $work/camera_calibrator_amd/csrc && /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wno-unused-result -Wno-unused-value "${flags[@]}" -c $file -o /tmp/cc_variant_$name.o)
objs=""
for o in cc_intrinsics.o cc_intrinsics_persist.o cc_rig.o cc_zhang.o cc_points.o cc_common.o cc_comm.o data_generator.o rig_scenario.o geometry.o; do
  if [ "$o" = "${file%.hip}.o" ]; then objs="$objs /tmp/cc_variant_$name.o"; else objs="$objs $src/$o"; fi
done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $out/libcc_$name.so $objs -ldl -pthread -Wl,-rpath,/opt/rocm/lib
rm -rf $work
echo scripts/ablate_build/libcc_$name.so
