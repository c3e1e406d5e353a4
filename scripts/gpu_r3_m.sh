#!/bin/bash
# round 3, call M: full suite on the final code, the driver's bench command, the round's rocprofv3 evidence
set -u
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/r3m
mkdir -p $OUT
cd $ROOT
timeout 900 python -m pytest tests -m gpu -q -p no:cacheprovider > $OUT/pytest.log 2>&1; echo "pytest rc=$?" >> $OUT/pytest.log
tail -6 $OUT/pytest.log | cut -c1-220
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 > $OUT/bench.json 2> $OUT/bench.err; echo "bench rc=$?"
python - <<PY
import json
d = json.load(open("$OUT/bench.json"))
print("value", d["value"], "ms/step", d["ms_per_step"], "attempts", d["config"]["solve_attempts_per_step"], "phases", d["phases_ms"])
print("roofline", d["roofline"]["achieved"], d["roofline"]["frac"], d["roofline"]["whole_factorisation"])
print("hbm", {k: (v["ms"], v["frac_of_8TBps"]) for k, v in d["hbm_kernels"].items() if isinstance(v, dict)})
print("connected", d["connected_scene"]["ms_per_step"], d["connected_scene"]["envelope_solver"]["ms_per_step"], d["connected_scene"]["envelope_solver"]["solve_ms"])
print("e2e", d["end_to_end_run_sfm"]["warm_call"]["wall_s"], "matcher", d["matcher"]["value"], "cpu", d["cpu_baseline"]["value"])
PY
bash scripts/profile_round.sh r03_cfg3_fd_final 2>&1 | tail -4
