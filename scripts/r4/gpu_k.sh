#!/bin/bash
ulimit -c 0
cd /root/repo
timeout 600 python -m pytest tests/test_multi_gpu.py -x -q 2>&1 | tail -15
