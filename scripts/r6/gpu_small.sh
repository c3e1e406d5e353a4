#!/bin/bash
# round 6: one LM iteration of a 14 / 50-camera problem launch by launch (kernel trace), and the latency table
ulimit -c 0
cd /root/repo
O=gpurun_out/r6small; mkdir -p $O
TAG=${1:-base}
for cfg in "14 1500" "50 10000"; do
  set -- $cfg
  (cd /tmp && export TMPDIR=/tmp && rm -rf /tmp/p_it && timeout 300 rocprofv3 --kernel-trace -d /tmp/p_it -o t --output-format csv -- python /root/repo/scripts/small_iter_trace.py run $1 $2 > /dev/null 2>/tmp/it.err; python /root/repo/scripts/small_iter_trace.py show $(find /tmp/p_it -name "*kernel_trace.csv" | head -1)) | tee $O/iter_${1}cams_$TAG.txt
done
SMALL_NO_REF=1 timeout 600 python scripts/small_problem_latency.py 2>&1 | cut -c1-260 | tee $O/latency_$TAG.txt
