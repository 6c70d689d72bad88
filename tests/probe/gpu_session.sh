#!/bin/bash
# scratch driver for one gpurun call (rewritten per session)
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out; export TMPDIR=/tmp
python -m pytest tests -x -q -m gpu -k "query_time or fake_input or sampler_parity" -s > gpurun_out/r05w_qt_tests.log 2>&1; tail -30 gpurun_out/r05w_qt_tests.log
