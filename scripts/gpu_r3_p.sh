#!/bin/bash
# stream priorities in the dense Cholesky schedule: bulk stream lowest / side stream highest
mkdir -p gpurun_out
for cfg in "0 0" "1 0" "0 1" "1 1" "0 0"; do set -- $cfg
  BSFM_BULK_PRIO=$1 BSFM_SIDE_PRIO=$2 timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-structure-aware --no-end-to-end --no-matcher --no-connected > /tmp/b.json 2>/tmp/b.err
  python - "$1" "$2" <<'PY'
import json,sys
d=json.loads(open("/tmp/b.json").read().strip().splitlines()[-1])
print(f"bulk_low={sys.argv[1]} side_high={sys.argv[2]}: {d['value']} it/s, {d['ms_per_step']} ms/step, solve {d['phases_ms']['solve']} ms, syrk {d['roofline']['achieved']} TF, attempts {d['config']['solve_attempts_per_step']}")
PY
done | tee gpurun_out/r3p_stream_prio.txt
