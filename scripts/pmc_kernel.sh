#!/bin/bash
# Runs ON THE GPU BOX: SQ counters of one kernel (substring match) under bench.py, a few counters per pass.
# usage: scripts/pmc_kernel.sh <kernel substring> <pass1 counters, space separated> [-- <pass2 counters> ...]
set -u
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
KERN=$1; shift
cd /tmp && export TMPDIR=/tmp
i=0
pass=()
run_pass() {
    [ ${#pass[@]} -eq 0 ] && return
    i=$((i+1))
    timeout 200 rocprofv3 --pmc "${pass[@]}" -d /tmp/pk_$i -o c --output-format csv -- python $ROOT/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-structure-aware > /dev/null 2> /tmp/pk_$i.err || tail -3 /tmp/pk_$i.err
    python - "$KERN" /tmp/pk_$i <<'PY'
import csv, glob, collections, sys
kern, d = sys.argv[1], sys.argv[2]
for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
    acc = collections.defaultdict(lambda: [set(), 0.0])
    for r in csv.DictReader(open(f)):
        if kern not in r["Kernel_Name"]:
            continue
        a = acc[r["Counter_Name"]]; a[0].add(r["Dispatch_Id"]); a[1] += float(r["Counter_Value"])
    for c, (ids, t) in sorted(acc.items()):
        print(f"{kern} {c} per launch: {t / max(len(ids), 1):.4g} ({len(ids)} launches)")
PY
    pass=()
}
for a in "$@"; do
    if [ "$a" == "--" ]; then run_pass; else pass+=("$a"); fi
done
run_pass
