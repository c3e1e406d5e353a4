#!/bin/bash
# SQ / cache counters of selected kernels of the BA iteration (ON THE GPU BOX through gpurun), one rocprofv3 --pmc pass per counter set.
#   usage: bash scripts/pmc_kernels.sh "<kernel name substring> [...]"      -> mean per launch of every counter
set -u
NAMES=${1:-"k_syrk_update k_potrf_diag k_trsm_panel64 k_backsub k_cam_blocks k_jacobian"}
cd /tmp; export TMPDIR=/tmp
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS" \
           "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" \
           "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCC_HIT_sum TCC_MISS_sum GRBM_GUI_ACTIVE SQ_BUSY_CU_CYCLES SQ_VALU_MFMA_BUSY_CYCLES"; do
  tag=$(echo $set | cut -d' ' -f1)
  timeout 300 rocprofv3 --pmc $set -d /tmp/p_k_$tag -o c --output-format csv -- python $GRAFT_REPO_ROOT/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-matcher --no-connected --no-structure-aware > /dev/null 2> /tmp/k_$tag.err
  f=$(find /tmp/p_k_$tag -name "*counter_collection.csv" | head -1)
  [ -n "$f" ] && python - "$f" "$NAMES" <<'PY'
import csv, sys, collections
names = sys.argv[2].split()
acc = collections.defaultdict(lambda: [set(), 0.0])
for r in csv.DictReader(open(sys.argv[1])):
    for nm in names:
        if nm in r["Kernel_Name"]:
            a = acc[(nm, r["Counter_Name"])]; a[0].add(r["Dispatch_Id"]); a[1] += float(r["Counter_Value"])
for (nm, c), (ids, tot) in sorted(acc.items()):
    print("%-16s %-28s launches %5d  mean %.4g" % (nm, c, len(ids), tot / max(len(ids), 1)))
PY
done
