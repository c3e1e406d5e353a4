#!/bin/bash
set -u
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/r3i
mkdir -p $OUT
cd $ROOT
for f in tests/test_*.py; do
  timeout 300 python -m pytest $f -m gpu -q -p no:cacheprovider -x --tb=line > $OUT/$(basename $f).log 2>&1
  echo "$f rc=$? $(tail -1 $OUT/$(basename $f).log | cut -c1-100)"
  grep -E "^FAILED|^/.*Error|Error" $OUT/$(basename $f).log | head -5 | cut -c1-250
done
