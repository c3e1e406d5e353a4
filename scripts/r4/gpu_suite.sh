#!/bin/bash
ulimit -c 0
cd /root/repo
mkdir -p gpurun_out/r4s
timeout 2400 python -m pytest tests -x -q -m gpu 2>&1 | tail -15 | tee gpurun_out/r4s/pytest_gpu.txt
