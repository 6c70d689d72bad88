# scratch driver of one gpurun call (edited per session): results under gpurun_out/
mkdir -p gpurun_out
T=r04zy
CMD="python $GRAFT_REPO_ROOT/bench.py --config5 --steps 5 --warmup 1 --no-cpu-baseline --no-small-batches --no-score-fwd" bash profiles/collect.sh ${T}_config5 "trace fetch write sq2" > gpurun_out/${T}_config5_collect.log 2>&1
DEDF_SUMMARY_DIR=gpurun_out python profiles/summarize.py ${T}_config5 > gpurun_out/${T}_config5_summarize.log 2>&1
find gpurun_out -maxdepth 1 -type d -name "${T}*" -exec rm -rf {} +
python - <<'PY'
import json
d=json.load(open("gpurun_out/r04zy_config5_pmc_summary.json"))
print({k:(round(v,4) if isinstance(v,float) else v) for k,v in d.items() if k!="per_kernel" and not k.endswith("definition")})
PY
python bench.py --no-cpu-baseline --no-extractors --no-small-batches | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']; print(round(d['value']), r['frac'], r['traffic'], r['traffic_source'], r.get('frac_mfma_issued'))"
du -sh gpurun_out
