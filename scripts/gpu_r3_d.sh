#!/bin/bash
# round 3, call D: where does the new Schur task kernel spend its time?  kernel stats + SQ / cache counters
set -u
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/r3d
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
B="python $ROOT/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-matcher --no-connected --no-structure-aware --no-end-to-end"
timeout 250 rocprofv3 --kernel-trace --stats -d /tmp/p_stats -o st --output-format csv -- $B > $OUT/bench_under_profiler.json 2> /tmp/st.err
cp $(find /tmp/p_stats -name "*kernel_stats.csv" | head -1) $OUT/kernel_stats.csv
python - <<PY
import csv
for r in list(csv.DictReader(open("$OUT/kernel_stats.csv")))[:16]:
    print(r["Name"][:70], r["Calls"], r["AverageNs"], r["Percentage"])
PY
bash $ROOT/scripts/pmc_kernel.sh k_schur_tasks SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT -- SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_MFMA SQ_INSTS_SALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_VALU -- SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES GRBM_GUI_ACTIVE SQ_INST_CYCLES_VMEM SQ_WAVES -- TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCC_HIT_sum TCC_MISS_sum -- FETCH_SIZE > $OUT/schur_counters.txt 2>&1
cat $OUT/schur_counters.txt
