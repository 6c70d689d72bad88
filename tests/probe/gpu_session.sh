#!/bin/bash
# scratch driver for one gpurun call (rewritten per session)
ROOT=$GRAFT_REPO_ROOT; OUT=$ROOT/gpurun_out; mkdir -p $OUT; export TMPDIR=/tmp
cd $ROOT
python -m pytest tests -x -q -m gpu -k "tiny_batches or fake_input or sampler or other_scale_counts or query_time or sharded or overflow or workspace or philox or c1_anchored or agent or randomised" > $OUT/r05zj_tests.log 2>&1; tail -5 $OUT/r05zj_tests.log
python tests/probe/small_batch.py 2 200 > $OUT/r05zj_small_batch.log 2>&1
DEDF_FUSE_MASKS=0 python tests/probe/small_batch.py 2 200 > $OUT/r05zj_small_batch_nofuse.log 2>&1
python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-extractors > $OUT/r05zj_bench.json 2>/dev/null
