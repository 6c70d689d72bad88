#!/bin/bash
# scratch driver for one gpurun call (rewritten per session)
ROOT=$GRAFT_REPO_ROOT; OUT=$ROOT/gpurun_out; mkdir -p $OUT; cd $ROOT; export TMPDIR=/tmp
export DEDF_STRESS_LMAX3=1
timeout 900 python tests/stress_parity.py 22 606 sample > $OUT/r05zy_stress_sample.log 2>&1; grep -A1 "sample  17\|sample  21" $OUT/r05zy_stress_sample.log | cut -c1-400
