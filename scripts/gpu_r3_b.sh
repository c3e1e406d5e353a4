#!/bin/bash
# round 3, call B: 16-byte C traffic in the bulk kernel -- kernel alone, Cholesky parity, and the bench line with and without it.
set -u
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/r3b
mkdir -p $OUT
cd $ROOT
UB_ONLY_C16=1 timeout 120 scripts/_bin/ubench_syrk_stagger > $OUT/c16_ubench.txt 2>&1; echo "ubench rc=$?"
grep -v "^census\|examples\|workgroups of" $OUT/c16_ubench.txt
timeout 600 python -m pytest tests/test_chol_gpu.py tests/test_cfg3_gpu.py -m gpu -q -p no:cacheprovider -x > $OUT/pytest.log 2>&1; echo "pytest rc=$?" >> $OUT/pytest.log
tail -5 $OUT/pytest.log
for v in 0 3 2 1 3 0; do
  BSFM_SYRK_C16=$v timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-matcher --no-structure-aware --no-connected --no-end-to-end > $OUT/bench_c16_$v.json 2> $OUT/bench.err
  python - <<PY
import json
d = json.load(open("$OUT/bench_c16_$v.json"))
print("C16=$v value", d["value"], "ms/step", d["ms_per_step"], "solve", d["phases_ms"]["solve"], "syrk", d["roofline"]["achieved"], d["roofline"]["avg_launch_ms"])
PY
done
