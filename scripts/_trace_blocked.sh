cd /tmp; export TMPDIR=/tmp
GPU_MAX_HW_QUEUES=${Q:-4} BSFM_CHOL=blocked timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/p_bl$Q -o st --output-format csv -- python $GRAFT_REPO_ROOT/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-matcher --no-connected --no-structure-aware > /tmp/bl.json 2> /tmp/bl.err
python -c "
import json; d=json.load(open('/tmp/bl.json')); print('Q=${Q:-4}', d['value'], d['phases_ms']['solve'])"
python - <<'PY'
import csv,glob,os
f=glob.glob('/tmp/p_bl%s/**/*kernel_stats.csv' % os.environ.get('Q','4'),recursive=True)[0]
rows=list(csv.DictReader(open(f)))
for r in sorted(rows,key=lambda r:-float(r['TotalDurationNs']))[:9]:
    print(f"{r['Name'][:60]:60s} calls {r['Calls']:>6s} avg {float(r['AverageNs'])/1e3:9.1f} us total {float(r['TotalDurationNs'])/1e6:8.2f} ms")
PY
python $GRAFT_REPO_ROOT/scripts/trace_chain.py $(find /tmp/p_bl$Q -name "*kernel_trace.csv" | head -1) 7 10 | cut -c1-150 | head -50
