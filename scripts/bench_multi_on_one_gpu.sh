#!/bin/bash
# Rehearsal of `bench.py --gpus N` on a ONE-GPU box: N ranks that all use GPU 0 (LOCAL_RANK is forced
# to 0 for device selection via CC_BENCH_DEVICE). Exercises the control flow and the mailbox exchange;
# the timing is meaningless (N ranks share one GPU). N <= 4.
N=${1:-2}
export CC_BENCH_DEVICE=0
export CC_EXCHANGE=mailbox   # RCCL refuses two ranks on one device
exec python -m torch.distributed.run --nnodes=1 --nproc-per-node $N --master-addr 127.0.0.1 --master-port 29533 \
  bench.py --gpus $N --steps 40 --warmup 8 --frames ${FRAMES:-200}
