#!/bin/bash
# Control-flow check of bench.py's multi-rank path on a single-GPU box: two ranks share cuda:0, collectives over gloo.
# (The numbers are meaningless - both ranks compete for one device; the real path is RCCL, one rank per GPU.)
cd "$(dirname "$0")/.."
export DCREG_BENCH_BACKEND=gloo DCREG_BENCH_LOCAL_RANK=0
python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29517 bench.py --gpus 2 --steps 60 --warmup 20 "$@"
