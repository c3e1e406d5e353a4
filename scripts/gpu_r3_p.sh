#!/bin/bash
mkdir -p gpurun_out
for lds in 0 90000; do for ev in 1 0; do
  BSFM_X_BULK_LDS=$lds BSFM_SYRK_EVENTS=$ev timeout 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-structure-aware --no-end-to-end --no-matcher --no-connected > /tmp/b.json 2>/tmp/b.err
  python - "$lds" "$ev" <<'PY'
import json,sys
try:
    d=json.loads(open("/tmp/b.json").read().strip().splitlines()[-1])
    r=d.get('roofline') or {}
    print(f"bulk_lds={sys.argv[1]} syrk_events={sys.argv[2]}: solve {d['phases_ms']['solve']} ms, {d['ms_per_step']} ms/step, syrk {r.get('achieved')} TF, final cost {d['final_cost']:.3f}")
except Exception as e:
    print(sys.argv[1:], "failed", e, open("/tmp/b.err").read()[-300:])
PY
done; done | tee gpurun_out/r3p_bulk_one_per_cu.txt
