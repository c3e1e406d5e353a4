#!/bin/bash
ulimit -c 0
cd /root/repo
for c in ${CHUNKS:-160 96 64 48 32 16}; do
  BSFM_SCHUR_CHUNK=$c timeout 200 python bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-matcher --no-structure-aware --no-end-to-end 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); c=d['connected_scene']
print('chunk $c: clique schur %.3f ms (iter %.2f) | connected schur %.3f ms (iter %.2f, tasks %s) envelope iter %.2f' % (d['phases_ms']['schur'], d['ms_per_step'], c['phases_ms']['schur'], c['ms_per_step'], c.get('schur_tasks'), c['envelope_solver']['ms_per_step']))"
done
