#!/bin/bash
# Round 2, GPU call B: full -m gpu suite (no -x), then the bulk-kernel NT A/B at the headline config.
set -u
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/r02b
mkdir -p $OUT
cd $ROOT
timeout 1800 python -m pytest tests -m gpu -q -p no:cacheprovider > $OUT/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> $OUT/pytest_gpu.log
tail -15 $OUT/pytest_gpu.log
for nt in 1 0; do
  BSFM_SYRK_NT=$nt timeout 300 python bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-matcher --no-connected --no-structure-aware > $OUT/bench_nt$nt.json 2> $OUT/bench_nt$nt.err
  python -c "import json;d=json.load(open('$OUT/bench_nt$nt.json'));print('NT=$nt', d['ms_per_step'], d['phases_ms']['solve'], d['roofline']['achieved'])"
done
