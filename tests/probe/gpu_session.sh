#!/bin/bash
# scratch driver for one gpurun call (rewritten per session)
ROOT=$GRAFT_REPO_ROOT; OUT=$ROOT/gpurun_out; mkdir -p $OUT; export TMPDIR=/tmp
cd $ROOT
run() { python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-extractors --no-small-batches --no-score-fwd 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', round(d['value']), d['ms_per_step'])" >> $OUT/r05zn_ab.log; }
for i in 1 2; do
run base
DEDF_FILL_G=2 run fill_g2
DEDF_SMALL_BATCH_MAX=200000 run small_path_at_c2
done
cat $OUT/r05zn_ab.log
