#!/bin/bash
mkdir -p gpurun_out
for cus in 0 8 16 24 32 48; do for ev in 1 0; do
  BSFM_PANEL_CUS=$cus BSFM_SYRK_EVENTS=$ev timeout 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-structure-aware --no-end-to-end --no-matcher --no-connected > /tmp/b.json 2>/tmp/b.err
  python - "$cus" "$ev" <<'PY'
import json,sys
try:
    d=json.loads(open("/tmp/b.json").read().strip().splitlines()[-1])
    print(f"panel_cus={sys.argv[1]} syrk_events={sys.argv[2]!r}: solve {d['phases_ms']['solve']} ms, {d['ms_per_step']} ms/step, attempts {d['config']['solve_attempts_per_step']}")
except Exception as e:
    print(sys.argv[1:], "failed", e, open("/tmp/b.err").read()[-300:])
PY
done; done | tee gpurun_out/r3p_cus_events.txt
