cd /tmp && export TMPDIR=/tmp
for N in 9000 3712 1350; do
  rm -rf /tmp/prof_o
  BSFM_CHOL_REPS=6 timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/prof_o -o run --output-format csv -- python /root/repo/scripts/r4/chol_reps.py $N > /tmp/reps.txt 2>&1
  f=$(find /tmp/prof_o -name '*kernel_stats.csv' | head -1)
  echo "== n = $N: $(grep residual /tmp/reps.txt | cut -c1-50)"; python /root/repo/scripts/kstats.py $f 4 | grep bwd
done
