#!/bin/bash
# Round 2, final evidence run (ON THE GPU BOX through gpurun): full -m gpu suite, smoke, the default bench line, rocprofv3 kernel
# stats + PMC traffic / MFMA counters of the BA command, small-problem latencies, the panel-engine timeline.
set -u
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/r02f
mkdir -p $OUT $ROOT/gpurun_out/prof
cd $ROOT
export TMPDIR=/tmp
timeout 1500 python -m pytest tests -m gpu -q -p no:cacheprovider > $OUT/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> $OUT/pytest_gpu.log
tail -4 $OUT/pytest_gpu.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $OUT/smoke.log 2>&1; echo "smoke rc=$?" >> $OUT/smoke.log; tail -2 $OUT/smoke.log
timeout 900 python bench.py > $OUT/bench_default.json 2> $OUT/bench_default.err; echo "bench rc=$?"
timeout 900 python bench.py --steps 20 --warmup 2 --no-cpu-baseline --no-matcher > $OUT/bench_steps20.json 2> $OUT/bench_steps20.err
timeout 900 bash scripts/profile_round.sh r02_cfg3_fd_final > $OUT/profile_round.log 2>&1
cp gpurun_out/prof/r02_cfg3_fd_final_* $OUT/ 2>/dev/null
timeout 600 python scripts/small_problem_latency.py > $OUT/small_problem_latency.txt 2>&1
BSFM_CHOL=engine BSFM_CHOL_WORKERS=240 BSFM_DEBUG_ENGINE=1 timeout 300 python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-matcher --no-connected --no-structure-aware > $OUT/bench_engine.json 2> $OUT/engine_timeline.txt
python - <<PY
import json
d = json.load(open("$OUT/bench_default.json"))
print("value", d["value"], "ms", d["ms_per_step"], "roof", d["roofline"]["achieved"], d["roofline"]["frac"], "phases", d["phases_ms"])
print("cpu", d["cpu_baseline"]["value"], "matcher", d["matcher"]["value"], d["matcher"]["roofline"]["frac"], "connected", d["connected_scene"]["ms_per_step"])
PY
