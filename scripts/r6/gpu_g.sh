#!/bin/bash
# round 6, call G: the distributed Cholesky between 2 and 4 processes on one GPU; the Cholesky and multi-GPU suites
ulimit -c 0
cd /root/repo
O=gpurun_out/r6g; mkdir -p $O
timeout 1500 python -m pytest tests/test_multi_gpu.py -q -m gpu -x -k "distributed" 2>&1 | tail -25 | tee $O/pytest_dist.txt
timeout 1500 python -m pytest tests/test_chol_gpu.py tests/test_multi_gpu.py tests/test_ba_gpu.py tests/test_boundary_link.py -q -m gpu 2>&1 | tail -8 | tee $O/pytest_rest.txt
