#!/bin/bash
# run the kermit GPU test repeatedly; on a hang dump native stacks with rocgdb
for i in 1 2 3 4 5 6; do
  python -m pytest tests/test_ba_gpu.py -x -q -m gpu -k kermit > /tmp/hh.log 2>&1 &
  pid=$!
  for s in $(seq 1 12); do sleep 1; kill -0 $pid 2>/dev/null || break; done
  if kill -0 $pid 2>/dev/null; then
    echo "HANG on try $i (pid $pid)"
    which rocgdb gdb 2>/dev/null
    (rocgdb -p $pid -batch -ex "thread apply all bt 12" 2>/dev/null || gdb -p $pid -batch -ex "thread apply all bt 12" 2>/dev/null) | grep -E "^#|Thread" | head -60
    kill -9 $pid; exit 0
  else
    tail -1 /tmp/hh.log
  fi
done
echo "no hang in 6 tries"
