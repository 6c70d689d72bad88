# scratch driver of one gpurun call (edited per session): results under gpurun_out/
mkdir -p gpurun_out
R=${GRAFT_REPO_ROOT:-/root/repo}
bash profiles/collect.sh r03v "trace fetch write sq sq2 sq3 tcp"
CMD="python $R/bench.py --lmax 3 --steps 5 --warmup 1 --no-cpu-baseline --no-extractors --no-small-batches" bash profiles/collect.sh r03v_lmax3 "trace fetch write sq sq2"
cd $R
python profiles/summarize.py r03v > gpurun_out/r03v_summarize.log 2>&1
python profiles/summarize.py r03v_lmax3 > gpurun_out/r03v_lmax3_summarize.log 2>&1
cp profiles/r03v_kernel_stats.txt profiles/r03v_pmc_summary.json profiles/r03v_lmax3_kernel_stats.txt profiles/r03v_lmax3_pmc_summary.json gpurun_out/ 2>/dev/null
rm -rf gpurun_out/*_trace gpurun_out/*_pmc_fetch gpurun_out/*_pmc_write gpurun_out/*_pmc_sq gpurun_out/*_pmc_sq2 gpurun_out/*_pmc_sq3 gpurun_out/*_pmc_tcp
ls gpurun_out | head -30
