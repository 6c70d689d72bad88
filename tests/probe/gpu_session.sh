#!/bin/bash
# scratch driver for one gpurun call (rewritten per session)
ROOT=$GRAFT_REPO_ROOT; OUT=$ROOT/gpurun_out; mkdir -p $OUT; export TMPDIR=/tmp
cd $ROOT
python -m pytest tests -x -q -m gpu -k "tiny_batches or other_scale_counts or max_neighbors or sampler_parity or c2_bi_equi or workspace or eight_scales or ragged or c0_plumbing or medium or c3_total or sharded or zero_edge" > $OUT/r05zf_tests.log 2>&1; tail -3 $OUT/r05zf_tests.log
python tests/probe/small_batch.py 2 200 > $OUT/r05zf_small_batch.log 2>&1
for t in 16 1000; do
cd /tmp
rocprofv3 --kernel-trace --stats -d $OUT/ft_$t -- python $ROOT/tests/probe/fill_timing.py $t > $OUT/r05zf_fill_$t.log 2>&1
cd $ROOT
python - <<PY >> $OUT/r05zf_fill_timing.txt
import sqlite3, glob
f = sorted(glob.glob("$OUT/ft_$t/*/*_results.db"))[-1]
con = sqlite3.connect(f)
for r in con.execute("select name, total_calls, total_duration, average, percentage from top_kernels"):
    if "k_neighbors" in r[0] or "k_nbr" in r[0]: print("poses $t", f"{r[0][:60]:60s} {r[1]:6d} {r[3]:10.2f}")
PY
rm -rf $OUT/ft_$t
done
cat $OUT/r05zf_fill_timing.txt
python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-extractors > $OUT/r05zf_bench.json 2>/dev/null
