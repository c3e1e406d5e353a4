#!/bin/bash
# Runs ON THE GPU BOX (through gpurun): the round's rocprofv3 evidence for bench.py's default command.
#   1. kernel trace + stats        -> gpurun_out/prof/<tag>_kernel_stats.csv
#   2. PMC pass FETCH_SIZE         (separate runs: --pmc never together with trace flags other than kernel-trace,
#   3. PMC pass WRITE_SIZE          and the two TCC counters do not fit one pass)
#   4. scripts/pmc_summary.py      -> gpurun_out/prof/<tag>_pmc_traffic.json
# usage: scripts/profile_round.sh <tag>      e.g. r01_cfg3_fd_v3
set -u
TAG=${1:-r01}
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/prof
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout 250 rocprofv3 --kernel-trace --stats -d /tmp/p_stats -o st --output-format csv -- python $ROOT/bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-matcher --no-connected --no-structure-aware --no-end-to-end --no-dense-valued > $OUT/${TAG}_bench_under_profiler.json 2> /tmp/st.err
cp $(find /tmp/p_stats -name "*kernel_stats.csv" | head -1) $OUT/${TAG}_kernel_stats.csv
BSFM_FLOW_TRACE=1 BSFM_FLOW_TRACE_FILE=/tmp/flow_trace.txt timeout 200 python $ROOT/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-matcher --no-connected --no-structure-aware --no-end-to-end --no-dense-valued > /dev/null 2>&1; python $ROOT/scripts/r4/trace_stats.py /tmp/flow_trace.txt > $OUT/${TAG}_chain_timeline.txt 2>&1
timeout 250 rocprofv3 --pmc FETCH_SIZE -d /tmp/p_f -o f --output-format csv -- python $ROOT/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-matcher --no-connected --no-structure-aware --no-end-to-end --no-dense-valued > /dev/null 2> /tmp/f.err
timeout 250 rocprofv3 --pmc WRITE_SIZE -d /tmp/p_w -o w --output-format csv -- python $ROOT/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-matcher --no-connected --no-structure-aware --no-end-to-end --no-dense-valued > /dev/null 2> /tmp/w.err
timeout 250 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES GRBM_GUI_ACTIVE -d /tmp/p_m -o m --output-format csv -- python $ROOT/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-matcher --no-connected --no-structure-aware --no-end-to-end --no-dense-valued > /dev/null 2> /tmp/m.err
timeout 250 rocprofv3 --pmc MfmaUtil -d /tmp/p_u -o u --output-format csv -- python $ROOT/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-matcher --no-connected --no-structure-aware --no-end-to-end --no-dense-valued > /dev/null 2> /tmp/u.err
python $ROOT/scripts/pmc_summary.py $(find /tmp/p_f -name "*counter_collection.csv" | head -1) $(find /tmp/p_w -name "*counter_collection.csv" | head -1) $OUT/${TAG}_pmc_traffic.json $(find /tmp/p_m /tmp/p_u -name "*counter_collection.csv")
tail -3 /tmp/m.err /tmp/u.err | cut -c1-200
head -12 $OUT/${TAG}_kernel_stats.csv | cut -c1-150
# which summary bench.py's roofline.traffic quotes from now on (bench.py:pmc_summary reads profiles/LATEST; copy both into profiles/)
echo "{\"pmc_traffic\": \"${TAG}_pmc_traffic.json\", \"kernel_stats\": \"${TAG}_kernel_stats.csv\"}" > $OUT/LATEST
