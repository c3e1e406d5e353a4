#!/bin/bash
mkdir -p gpurun_out
for u in "" 1 ""; do
  BSFM_UNSAFE_NOWAIT=$u timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-structure-aware --no-end-to-end --no-matcher > /tmp/b.json 2>/tmp/b.err
  python - "$u" <<'PY'
import json,sys
d=json.loads(open("/tmp/b.json").read().strip().splitlines()[-1])
c=d.get("connected_scene",{}); e=c.get("envelope_solver",{})
print(f"nowait={sys.argv[1]!r}: {d['value']} it/s, {d['ms_per_step']} ms/step, solve {d['phases_ms']['solve']} ms, attempts {d['config']['solve_attempts_per_step']}, final cost {d['final_cost']:.6f}; connected dense solve {c.get('phases_ms',{}).get('solve')} envelope solve {e.get('solve_ms')}")
PY
done | tee gpurun_out/r3p_nowait.txt
