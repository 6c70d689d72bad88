# scratch driver of one gpurun call (edited per session): results under gpurun_out/
mkdir -p gpurun_out
T=r04d
python tests/probe/call_overhead.py > gpurun_out/${T}_call_overhead.log 2>&1; grep lmax gpurun_out/${T}_call_overhead.log
bash profiles/collect.sh ${T} "trace fetch write sq sq2 sq3 bench" > gpurun_out/${T}_collect.log 2>&1
DEDF_SUMMARY_DIR=gpurun_out python profiles/summarize.py ${T} > gpurun_out/${T}_summarize.log 2>&1; grep -v per_kernel gpurun_out/${T}_summarize.log | tail -40
CMD="python $GRAFT_REPO_ROOT/bench.py --lmax 3 --steps 5 --warmup 1 --no-cpu-baseline --no-extractors --no-small-batches" bash profiles/collect.sh ${T}_lmax3 "trace sq2" > gpurun_out/${T}_lmax3_collect.log 2>&1
DEDF_SUMMARY_DIR=gpurun_out python profiles/summarize.py ${T}_lmax3 > gpurun_out/${T}_lmax3_summarize.log 2>&1; grep -v per_kernel gpurun_out/${T}_lmax3_summarize.log | tail -12
CMD="python $GRAFT_REPO_ROOT/bench.py --config5 --steps 5 --warmup 1 --no-cpu-baseline --no-small-batches" bash profiles/collect.sh ${T}_config5 "trace fetch write sq2" > gpurun_out/${T}_config5_collect.log 2>&1
DEDF_SUMMARY_DIR=gpurun_out python profiles/summarize.py ${T}_config5 > gpurun_out/${T}_config5_summarize.log 2>&1; grep -v per_kernel gpurun_out/${T}_config5_summarize.log | tail -12
find gpurun_out -maxdepth 1 -type d -name "${T}*" -exec rm -rf {} +      # the rocpd databases (hundreds of MB) stay on the box: the summaries travel
grep lmax gpurun_out/${T}_call_overhead.log
du -sh gpurun_out
