#!/bin/bash
# scratch driver for one gpurun call (rewritten per session)
ROOT=$GRAFT_REPO_ROOT; OUT=$ROOT/gpurun_out; mkdir -p $OUT; cd $ROOT; export TMPDIR=/tmp
python -m pytest tests -x -q -m gpu --durations=5 > $OUT/r05zu_gpu_suite.log 2>&1; tail -3 $OUT/r05zu_gpu_suite.log
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $OUT/r05zu_smoke.log 2>&1; tail -1 $OUT/r05zu_smoke.log
python bench.py --gpus 1 --steps 20 --warmup 5 > $OUT/r05zu_bench.json 2> $OUT/r05zu_bench.err
