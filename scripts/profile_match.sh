#!/bin/bash
# Matcher evidence (ON THE GPU BOX through gpurun): bench line of config 5, rocprofv3 kernel stats of the same command, SQ counters
# of the scan kernels (k_match_bound / k_match_l2) in their own passes (120 images).   usage: bash scripts/profile_match.sh <tag>   -> gpurun_out/prof/<tag>_*
set -u
TAG=${1:-match}
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/prof
mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
timeout 600 python $ROOT/bench.py --workload match > $OUT/${TAG}_bench.json 2> /tmp/mb.err
# per-launch durations: ONE stream (BSFM_MATCH_STREAMS=1), otherwise consecutive launches overlap and every one of them looks twice as long
BSFM_MATCH_STREAMS=1 timeout 600 python $ROOT/bench.py --workload match > $OUT/${TAG}_bench_one_stream.json 2> /tmp/mb1.err
BSFM_MATCH_STREAMS=1 timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/p_m -o st --output-format csv -- python $ROOT/bench.py --workload match --steps 4 --no-cpu-baseline > $OUT/${TAG}_bench_under_profiler.json 2> /tmp/stm.err
cp $(find /tmp/p_m -name "*kernel_stats.csv" | head -1) $OUT/${TAG}_kernel_stats.csv
: > $OUT/${TAG}_pmc_counters.txt
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES" \
           "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVES GRBM_GUI_ACTIVE SQ_INST_CYCLES_VMEM"; do
  tag=$(echo $set | cut -d' ' -f1)
  BSFM_MATCH_STREAMS=1 timeout 300 rocprofv3 --pmc $set -d /tmp/p_mc_$tag -o c --output-format csv -- python $ROOT/bench.py --workload match --steps 4 --match-images 120 --no-cpu-baseline > /dev/null 2> /tmp/mc_$tag.err
  f=$(find /tmp/p_mc_$tag -name "*counter_collection.csv" | head -1)
  [ -n "$f" ] && python - "$f" >> $OUT/${TAG}_pmc_counters.txt <<'PY'
import csv, sys, collections, re
acc = collections.defaultdict(lambda: [set(), 0.0])
for r in csv.DictReader(open(sys.argv[1])):
    m = re.search(r"k_match_\w+", r["Kernel_Name"])
    if not m: continue
    a = acc[m.group(0) + " " + r["Counter_Name"]]; a[0].add(r["Dispatch_Id"]); a[1] += float(r["Counter_Value"])
for k, (ids, tot) in acc.items():
    print(k, "launches", len(ids), "mean_per_launch", tot / max(len(ids), 1))
PY
done
head -c 400 $OUT/${TAG}_bench.json; echo; cat $OUT/${TAG}_pmc_counters.txt; grep -E "k_match_|k_pair_" $OUT/${TAG}_kernel_stats.csv | cut -c1-200
