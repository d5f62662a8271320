#!/bin/bash
# 2 ranks on the ONE GPU of a gpurun box over gloo: exercises the multi-rank bench path with device
# tensors (the RCCL path itself needs one GPU per rank and is run by the driver's scaling bench)
export TMPDIR=/tmp DCA_AMD_DIST_BACKEND=gloo
timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus 2 --steps 5 --warmup 2 --cells 20000 --genes 5000 --batch-size 1024 2>&1 | tail -5 | cut -c1-700
