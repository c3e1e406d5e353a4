#!/bin/bash
ulimit -c 0
cd /root/repo
for b in flow_dbg; do for n in 200 384; do echo "=== $b $n"; timeout 40 scripts/r4/_build/$b $n 1 2>&1 | cut -c1-300 | head -48; done; echo "=== one launch 384"; timeout 40 scripts/r4/_build/$b 384 0 2>&1 | cut -c1-300 | head; echo "=== one launch 1799"; timeout 40 scripts/r4/_build/$b 1799 0 2>&1 | cut -c1-300 | head; done
