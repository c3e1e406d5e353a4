#!/bin/bash
# round 6, call H: the whole GPU suite (row kernel removed, distributed Cholesky in the LM loop and in bench --gpus 2), smoke
ulimit -c 0
cd /root/repo
O=gpurun_out/r6h; mkdir -p $O
timeout 3000 python -m pytest tests -q -m gpu 2>&1 | tail -15 | tee $O/pytest_gpu.txt
timeout 600 python __graft_entry__.py smoke 2>&1 | tail -3 | tee $O/smoke.txt
