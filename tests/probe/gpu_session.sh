#!/bin/bash
# scratch driver for one gpurun call (rewritten per session)
ROOT=$GRAFT_REPO_ROOT; OUT=$ROOT/gpurun_out; mkdir -p $OUT; cd $ROOT; export TMPDIR=/tmp
python -m pytest tests -x -q -m gpu -k "query_time" -s > $OUT/r05zt_qt_tests.log 2>&1; tail -8 $OUT/r05zt_qt_tests.log
