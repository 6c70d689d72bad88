#!/bin/bash
# scratch driver for one gpurun call (rewritten per session)
ROOT=$GRAFT_REPO_ROOT; OUT=$ROOT/gpurun_out; mkdir -p $OUT; export TMPDIR=/tmp
cd $ROOT
python -m pytest tests -x -q -m gpu -k "radial_table or c1_anchored or half_precision_mode or fake_input or tiny or forward_with_one_time or sampler_parity" > $OUT/r05zp_tests.log 2>&1; tail -4 $OUT/r05zp_tests.log
run() { python bench.py --lmax 1 --scene 2048 --grasp 512 --poses-per-gpu 256 --steps 50 --warmup 5 --no-cpu-baseline --no-extractors --no-small-batches --no-score-fwd $2 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', round(d['value']), d['ms_per_step'], d['roofline'].get('frac'))" >> $OUT/r05zp_c1_ab.log; }
for i in 1 2; do
run table_auto
run per_edge --no-radial-table
done
cat $OUT/r05zp_c1_ab.log
python tests/probe/small_batch.py 2 200 > $OUT/r05zp_small_batch.log 2>&1; grep "lmax 1" $OUT/r05zp_small_batch.log
