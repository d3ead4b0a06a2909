#!/bin/bash
# Weak-scaling sweep of the ADMM hot path on ONE node: bench.py at N = 1, 2, 4, 8 GPUs (one process per GPU,
# torch.distributed over RCCL/xGMI; the batch is sharded, no collective before the final gather of x).  Prints one JSON
# line per N.  usage: tools/scale.sh [workload=cfg4] [steps=3] [warmup=1] ["1 2 4 8"]
# (the driver runs the same commands itself at round end; this script is for a node where all 8 GPUs are visible)
set -u
cd "$(dirname "$0")/.."
W=${1:-cfg4}; K=${2:-3}; WU=${3:-1}; NS=${4:-"1 2 4 8"}
export HSA_ENABLE_IPC_MODE_LEGACY=0
for N in $NS; do
  if [ "$N" = 1 ]; then
    python bench.py --gpus 1 --steps "$K" --warmup "$WU" --workload "$W" --no-cpu-baseline --no-cfg3-leg
  else
    PORT=$((29500 + RANDOM % 2000))
    python -m torch.distributed.run --nnodes=1 --nproc-per-node "$N" --master-addr 127.0.0.1 --master-port "$PORT" \
      bench.py --gpus "$N" --steps "$K" --warmup "$WU" --workload "$W" --no-cpu-baseline --no-cfg3-leg
  fi || { echo "{\"n_gpus\": $N, \"error\": \"bench.py failed\"}"; exit 1; }
done
