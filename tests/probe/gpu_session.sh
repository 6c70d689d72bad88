#!/bin/bash
# scratch driver for one gpurun call (rewritten per session)
ROOT=$GRAFT_REPO_ROOT; OUT=$ROOT/gpurun_out; mkdir -p $OUT; export TMPDIR=/tmp
for t in 4 2 8 4 8; do
  cd /tmp
  if [ $t = 4 ]; then unset DEDF_LIB; else export DEDF_LIB=$ROOT/diffusion_edf_amd/csrc/libdedf_u$t.so; fi
  rocprofv3 --kernel-trace --stats -d $OUT/ft_$t -- python $ROOT/tests/probe/fill_timing.py 1000 > $OUT/r05zl_u$t.log 2>&1
  cd $ROOT
  python - <<PY >> $OUT/r05zl_aggregate_u.txt
import sqlite3, glob
f = sorted(glob.glob("$OUT/ft_$t/*/*_results.db"))[-1]
con = sqlite3.connect(f)
for r in con.execute("select name, total_calls, total_duration, average, percentage from top_kernels"):
    if "k_aggregate" in r[0] or "k_node" in r[0]: print("U = $t", f"{r[0][:60]:60s} {r[1]:6d} {r[3]:10.2f}")
PY
  rm -rf $OUT/ft_$t
done
cat $OUT/r05zl_aggregate_u.txt
