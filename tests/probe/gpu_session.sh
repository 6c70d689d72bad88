#!/bin/bash
# scratch driver for one gpurun call (rewritten per session)
ROOT=$GRAFT_REPO_ROOT; OUT=$ROOT/gpurun_out; mkdir -p $OUT; export TMPDIR=/tmp
cd $ROOT
python -m pytest tests -x -q -m gpu -k "not (unet or keypoint or config5 or fp16_operand or randomised)" > $OUT/r05zr_tests.log 2>&1; tail -4 $OUT/r05zr_tests.log
run() { python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-extractors --no-small-batches --no-score-fwd 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', round(d['value']), d['ms_per_step'])" >> $OUT/r05zr_ab.log; }
for i in 1 2; do
run fused
DEDF_FUSE_AGG=0 run separate
done
cat $OUT/r05zr_ab.log
python tests/probe/small_batch.py 2 200 > $OUT/r05zr_small_batch.log 2>&1; grep lmax $OUT/r05zr_small_batch.log | cut -c1-330
