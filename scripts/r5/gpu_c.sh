#!/bin/bash
# counters of the two Schur kernels
ulimit -c 0
cd /root/repo
O=gpurun_out/r5c; mkdir -p $O
export GRAFT_REPO_ROOT=${GRAFT_REPO_ROOT:-/root/repo}
sed -i 's/--no-structure-aware > \/dev\/null/--no-structure-aware --no-end-to-end --no-dense-valued > \/dev\/null/' scripts/pmc_kernels.sh
bash scripts/pmc_kernels.sh "k_schur_rows k_schur_tasks" > $O/counters_rows.txt 2>&1
cat $O/counters_rows.txt | tail -60
