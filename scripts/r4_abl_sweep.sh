# timing-only ablations of k_rig_sweep_frame (results are wrong): per-launch sweep time without the cross-lane reduction, the passes, the assembly
R=$PWD
mkdir -p gpurun_out/r4l
for lib in cur abl_NO_RS abl_NO_PASS abl_NO_ASM; do
  if [ $lib = cur ]; then unset CC_LIB_PATH; else export CC_LIB_PATH=$R/scripts/ablate_build/libcc_$lib.so; fi
  echo "== $lib" >> gpurun_out/r4l/abl.txt
  timeout -k 10 200 python scripts/time_rig_sweep_scaling.py 2>&1 | cut -c1-230 >> gpurun_out/r4l/abl.txt
done
cat gpurun_out/r4l/abl.txt
