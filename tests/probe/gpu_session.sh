#!/bin/bash
# scratch driver for one gpurun call (rewritten per session)
ROOT=$GRAFT_REPO_ROOT; OUT=$ROOT/gpurun_out; mkdir -p $OUT; cd $ROOT; export TMPDIR=/tmp
export DEDF_STRESS_LMAX3=1
TAG=case23 python tests/probe/stress_case.py 505 23 > $OUT/r05zw_case23_resolved.log 2>&1; grep -E "RESULT|binding|fp32" $OUT/r05zw_case23_resolved.log | cut -c1-400
