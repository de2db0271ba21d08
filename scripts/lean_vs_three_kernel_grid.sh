export REPS=5
for C in 2 4; do for F in 256 400 510 600 800 1020; do for M in 4 50 125 300 500; do
  a=$(CC_RIG_PERSIST=1 C=$C F=$F M=$M python scripts/bench_rig.py 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(round(d['gpu_us_per_iteration'],1), d['iterations'])")
  b=$(CC_RIG_PERSIST=0 C=$C F=$F M=$M python scripts/bench_rig.py 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(round(d['gpu_us_per_iteration'],1), d['iterations'])")
  echo "C=$C F=$F M=$M lean: $a three: $b"
done; done; done
