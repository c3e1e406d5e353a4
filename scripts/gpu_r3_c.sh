#!/bin/bash
# round 3, call C: camera-major layout + new Schur task kernel: full parity suite, then the bench line
set -u
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/r3c
mkdir -p $OUT
cd $ROOT
timeout 900 python -m pytest tests -m gpu -q -p no:cacheprovider -x > $OUT/pytest.log 2>&1; echo "pytest rc=$?" >> $OUT/pytest.log
tail -25 $OUT/pytest.log
timeout 600 python bench.py --steps 20 --warmup 5 --no-matcher --no-end-to-end > $OUT/bench.json 2> $OUT/bench.err; echo "bench rc=$?"
python - <<PY
import json
d = json.load(open("$OUT/bench.json"))
print("value", d["value"], "ms/step", d["ms_per_step"], "phases", d["phases_ms"])
print("hbm", {k: (v["ms"], v["frac_of_8TBps"]) for k, v in d["hbm_kernels"].items() if isinstance(v, dict)})
print("schur", d["schur"])
print("structure_aware", {k: d["structure_aware"].get(k) for k in ("ms_per_step", "schur_ms", "final_cost_rel_diff_vs_dense")})
print("connected", {k: d.get("connected_scene", {}).get(k) for k in ("ms_per_step", "phases_ms")})
PY
tail -5 $OUT/bench.err
