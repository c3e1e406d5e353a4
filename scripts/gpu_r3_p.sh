#!/bin/bash
mkdir -p gpurun_out
for cfg in "1000 0 1" "1000 1 1" "36 1 1" "36 1 4" "36 2 1" "36 2 4" "36 4 4" "36 4 0"; do set -- $cfg
  BSFM_WIDE_PANEL=$1 BSFM_PANEL_CUS=$2 BSFM_SYRK_EVENTS=$3 timeout 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-structure-aware --no-end-to-end --no-matcher --no-connected > /tmp/b.json 2>/tmp/b.err
  python - $1 $2 $3 <<'PY'
import json,sys
try:
    d=json.loads(open("/tmp/b.json").read().strip().splitlines()[-1])
    r=d.get('roofline') or {}
    print(f"wide_panel>{sys.argv[1]} reserved_cus={sys.argv[2]} syrk_events={sys.argv[3]}: solve {d['phases_ms']['solve']} ms, {d['ms_per_step']} ms/step, syrk {r.get('achieved')} TF")
except Exception as ex:
    print(sys.argv[1:], "failed", ex, open("/tmp/b.err").read()[-300:])
PY
done | tee gpurun_out/r3p_wide_panel_cus.txt
