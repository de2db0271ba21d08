mkdir -p gpurun_out/r4f
one() { python scripts/bench_rig.py 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['cams'],d['frames'],d['pts'],d['iterations'],round(d['gpu_us_per_iteration'],2))"; }
for i in 1 2; do
echo -n "cur 2x1000x4 : " >> gpurun_out/r4f/lean.txt; C=2 F=1000 M=4 REPS=20 one >> gpurun_out/r4f/lean.txt
echo -n "r3  2x1000x4 : " >> gpurun_out/r4f/lean.txt; CC_LIB_PATH=scripts/ablate_build/libcc_r3.so C=2 F=1000 M=4 REPS=20 one >> gpurun_out/r4f/lean.txt
echo -n "cur 4x400x300 : " >> gpurun_out/r4f/lean.txt; C=4 F=400 M=300 REPS=20 one >> gpurun_out/r4f/lean.txt
echo -n "r3  4x400x300 : " >> gpurun_out/r4f/lean.txt; CC_LIB_PATH=scripts/ablate_build/libcc_r3.so C=4 F=400 M=300 REPS=20 one >> gpurun_out/r4f/lean.txt
done
python scripts/r4_status.py 2>&1 | grep -v amdgpu >> gpurun_out/r4f/lean.txt
cat gpurun_out/r4f/lean.txt
