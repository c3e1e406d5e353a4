#!/bin/bash
# Round 2, GPU call A (runs ON THE GPU BOX through gpurun): parity suite, CPU baseline at the headline config (host cores, in the
# background while the GPU works), default bench line, kernel stats for the BA path and the matcher, SQ counters of k_match_l2.
set -u
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/r02a
mkdir -p $OUT
cd $ROOT
export TMPDIR=/tmp
nproc > $OUT/host.txt; grep -m1 "model name" /proc/cpuinfo >> $OUT/host.txt
( timeout 1500 python scripts/cpu_baseline_cfg3.py $OUT/cpu_baseline_cfg3.json 3 > $OUT/cpu_baseline.log 2>&1 ) &
CPU_PID=$!
timeout 1500 python -m pytest tests -m gpu -x -q -p no:cacheprovider > $OUT/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> $OUT/pytest_gpu.log
tail -5 $OUT/pytest_gpu.log
timeout 600 python bench.py > $OUT/bench_default.json 2> $OUT/bench_default.err; echo "bench rc=$?"
cd /tmp
timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/p_ba -o st --output-format csv -- python $ROOT/bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-matcher --no-connected --no-structure-aware > $OUT/bench_under_profiler.json 2> /tmp/st.err
cp $(find /tmp/p_ba -name "*kernel_stats.csv" | head -1) $OUT/r02_cfg3_fd_a_kernel_stats.csv
python $ROOT/scripts/trace_chain.py $(find /tmp/p_ba -name "*kernel_trace.csv" | head -1) 40 42 > $OUT/r02_cfg3_fd_a_chain_timeline.txt 2>&1
timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/p_m -o st --output-format csv -- python $ROOT/bench.py --workload match --steps 4 > $OUT/match_under_profiler.json 2> /tmp/stm.err
cp $(find /tmp/p_m -name "*kernel_stats.csv" | head -1) $OUT/r02_match_a_kernel_stats.csv
rocprofv3 -L > $OUT/counters_list.txt 2>&1
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES" \
           "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVES GRBM_GUI_ACTIVE SQ_INST_CYCLES_VMEM"; do
  tag=$(echo $set | cut -d' ' -f1)
  timeout 300 rocprofv3 --pmc $set -d /tmp/p_mc_$tag -o c --output-format csv -- python $ROOT/bench.py --workload match --steps 4 --match-images 120 > /dev/null 2> /tmp/mc_$tag.err
  f=$(find /tmp/p_mc_$tag -name "*counter_collection.csv" | head -1)
  [ -n "$f" ] && python - "$f" >> $OUT/r02_match_a_pmc.txt <<'PY'
import csv, sys, collections
acc = collections.defaultdict(lambda: [set(), 0.0])
for r in csv.DictReader(open(sys.argv[1])):
    if "k_match_l2" not in r["Kernel_Name"]: continue
    a = acc[r["Counter_Name"]]; a[0].add(r["Dispatch_Id"]); a[1] += float(r["Counter_Value"])
for k, (ids, tot) in acc.items():
    print(k, "launches", len(ids), "mean_per_launch", tot / max(len(ids), 1))
PY
  tail -2 /tmp/mc_$tag.err | cut -c1-200 >> $OUT/r02_match_a_pmc.txt
done
wait $CPU_PID
tail -c 600 $OUT/cpu_baseline.log
head -c 1500 $OUT/bench_default.json
