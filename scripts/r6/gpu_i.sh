#!/bin/bash
# round 6, call I: distributed Cholesky timing on one GPU; headline bench with speculation at every size, A/B; iteration trace at the headline size
ulimit -c 0
cd /root/repo
O=gpurun_out/r6i; mkdir -p $O
timeout 900 python scripts/r6/dist_chol_timing.py 9000 2 2>&1 | tee $O/dist_chol_timing.txt
timeout 900 python scripts/r6/dist_chol_timing.py 9000 4 2>&1 | tail -6 | tee -a $O/dist_chol_timing.txt
B="python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-matcher --no-end-to-end --no-dense-valued --no-structure-aware --no-connected"
for cfg in "spec_all:X=1" "spec_off:BSFM_SPECULATE=0"; do
  tag=${cfg%%:*}; envs=${cfg#*:}
  env $envs timeout 600 $B 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$tag', 'ms/step', d['ms_per_step'], d['phases_ms'])"
done | tee $O/bench_speculate.txt
(cd /tmp && export TMPDIR=/tmp && rm -rf /tmp/p_it && timeout 300 rocprofv3 --kernel-trace -d /tmp/p_it -o t --output-format csv -- python /root/repo/bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-matcher --no-end-to-end --no-dense-valued --no-structure-aware --no-connected > /dev/null 2>&1; python /root/repo/scripts/trace_iter.py $(find /tmp/p_it -name "*kernel_trace.csv" | head -1)) | tee $O/iter_trace_headline.txt
timeout 300 python -m pytest tests/test_chol_gpu.py -q -m gpu -k "starved" 2>&1 | tail -3
