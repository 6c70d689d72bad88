#!/bin/bash
# scratch driver for one gpurun call (rewritten per session)
ROOT=$GRAFT_REPO_ROOT; OUT=$ROOT/gpurun_out; mkdir -p $OUT; cd $ROOT; export TMPDIR=/tmp
python -m pytest tests -x -q -m gpu --durations=8 > $OUT/r05zi_gpu_suite.log 2>&1; tail -3 $OUT/r05zi_gpu_suite.log
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $OUT/r05zi_smoke.log 2>&1; tail -1 $OUT/r05zi_smoke.log
bash profiles/collect.sh r05zi "trace fetch write sq sq2 sq3 bench" > $OUT/r05zi_collect.log 2>&1
cd $ROOT
DEDF_SUMMARY_DIR=gpurun_out python profiles/summarize.py r05zi > $OUT/r05zi_summarize.log 2>&1
rm -rf $OUT/r05zi_trace $OUT/r05zi_pmc_fetch $OUT/r05zi_pmc_write $OUT/r05zi_pmc_sq $OUT/r05zi_pmc_sq2 $OUT/r05zi_pmc_sq3
python bench.py --gpus 1 --steps 20 --warmup 5 --lmax 3 > $OUT/r05zi_lmax3_bench.json 2>/dev/null
python bench.py --gpus 1 --steps 20 --warmup 5 --half > $OUT/r05zi_half_bench.json 2>/dev/null
python bench.py --gpus 1 --steps 20 --warmup 5 --config5 > $OUT/r05zi_config5_bench.json 2>/dev/null
python bench.py --lmax 1 --scene 2048 --grasp 512 --poses-per-gpu 256 --steps 50 --warmup 5 --no-extractors > $OUT/r05zi_c1_bench.json 2>/dev/null
python bench.py --gpus 1 --steps 20 --warmup 5 --poses-per-gpu 8000 --no-extractors --no-small-batches > $OUT/r05zi_poses8000_bench.json 2>/dev/null
python tests/probe/small_batch.py 2 200 > $OUT/r05zi_small_batch.log 2>&1
du -sh $OUT | tail -1
