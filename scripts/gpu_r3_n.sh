#!/bin/bash
# matcher: parity tests (both kernels), A/B of the kernels on three key sets, the bench's matcher line
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_matcher.py tests/test_tracks.py -m gpu -x -q > gpurun_out/r3n_pytest_matcher.log 2>&1; tail -3 gpurun_out/r3n_pytest_matcher.log
AB_IMAGES=160 timeout 900 python scripts/match_rescan_ab.py 2>&1 | tee gpurun_out/r3n_match_ab.txt
for k in top2 auto; do BSFM_MATCH_KERNEL=$k timeout 600 python bench.py --workload match > gpurun_out/r3n_bench_match_$k.json 2> gpurun_out/r3n_bench_match_$k.err; python - <<PY
import json; d=json.loads(open("gpurun_out/r3n_bench_match_$k.json").read().strip().splitlines()[-1]); print("$k", d["value"], d["ms_per_step"], d["roofline"]["frac"], d["roofline"]["us_per_image_pair"], d["roofline"]["kernel_ms_per_pass"])
PY
done
