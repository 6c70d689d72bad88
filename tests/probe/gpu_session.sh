#!/bin/bash
# scratch driver for one gpurun call (rewritten per session)
ROOT=$GRAFT_REPO_ROOT; OUT=$ROOT/gpurun_out; mkdir -p $OUT; export TMPDIR=/tmp
cd $ROOT
DEDF_SMALL_BATCH=0 python -m pytest tests -x -q -m gpu -k "tiny_batches or fake_input or sampler_parity or other_scale_counts or c0_plumbing or ragged or medium" > $OUT/r05zk_tests_big_path.log 2>&1; tail -2 $OUT/r05zk_tests_big_path.log
python -m pytest tests -x -q -m gpu -k "c2_bi_equi or c3_total or sharded or c2_timed or lmax3 or ebm or max_neighbors or eight_scales or forward_with_one_time" > $OUT/r05zk_tests.log 2>&1; tail -2 $OUT/r05zk_tests.log
for i in 1 2; do
for v in 1 0; do
DEDF_PREP_FUSED=$v python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-extractors --no-small-batches --no-score-fwd 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('PREP_FUSED=$v', round(d['value']), d['ms_per_step'])" >> $OUT/r05zk_ab.log
done; done
cat $OUT/r05zk_ab.log
