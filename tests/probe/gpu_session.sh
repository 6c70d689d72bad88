#!/bin/bash
# scratch driver for one gpurun call (rewritten per session)
ROOT=$GRAFT_REPO_ROOT; OUT=$ROOT/gpurun_out; mkdir -p $OUT; cd $ROOT; export TMPDIR=/tmp
python tests/probe/small_batch.py 2 200 > $OUT/r05y_small_batch.log 2>&1
python -m pytest tests -x -q -m gpu -k "tiny_batches or other_scale_counts or max_neighbors or sampler_parity or c2_bi_equi or workspace or eight_scales" > $OUT/r05y_tests.log 2>&1; tail -3 $OUT/r05y_tests.log
COMMON="--no-cpu-baseline --no-extractors --no-small-batches --no-score-fwd"
cd /tmp
rocprofv3 --kernel-trace --stats -d $OUT/r05y_p16_trace -- python $ROOT/bench.py --poses-per-gpu 16 --steps 200 --warmup 5 $COMMON > $OUT/r05y_p16_trace_bench.json 2> $OUT/r05y_p16_trace.log
cd $ROOT
python - <<PY > $OUT/r05y_p16_kernel_stats.txt
import sqlite3, glob
f = sorted(glob.glob("$OUT/r05y_p16_trace/*/*_results.db"))[-1]
con = sqlite3.connect(f)
for r in con.execute("select name, total_calls, total_duration, average, percentage from top_kernels"):
    print(f"{r[0][:100]:100s} {r[1]:6d} {r[2]:14.1f} {r[3]:12.2f} {r[4]:6.2f}")
PY
rm -rf $OUT/r05y_p16_trace
