# scratch driver of one gpurun call (edited per session): results under gpurun_out/
mkdir -p gpurun_out
T=r04zz
timeout 2400 python -m pytest tests -m gpu -x -q > gpurun_out/${T}_gpu_suite.log 2>&1; tail -3 gpurun_out/${T}_gpu_suite.log
python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/${T}_smoke.log 2>&1; tail -2 gpurun_out/${T}_smoke.log
python bench.py > gpurun_out/${T}_bench.json 2> gpurun_out/${T}_bench.err
python bench.py --lmax 3 --no-cpu-baseline > gpurun_out/${T}_lmax3_bench.json 2> gpurun_out/${T}_lmax3_bench.err
python bench.py --config5 --no-cpu-baseline > gpurun_out/${T}_config5_bench.json 2> gpurun_out/${T}_config5_bench.err
python bench.py --poses-per-gpu 8000 --no-cpu-baseline --no-extractors --no-small-batches --no-score-fwd --steps 10 > gpurun_out/${T}_poses8000_bench.json 2>/dev/null
python - <<'PY'
import json
for f in ("", "lmax3_", "config5_", "poses8000_"):
    try:
        d=json.loads(open(f"gpurun_out/r04zz_{f}bench.json").read().strip().splitlines()[-1]); r=d["roofline"]
        print(f or "C2", round(d["value"]), round(d["ms_per_step"],4), "edge", round(r["avg_launch_ms"],4), "edges", round(d["config"]["edges_per_step_rank0"]), "frac", round(r["frac"],4), "fwd", d["config"].get("score_fwd_ms_at_t0.5"), "small", {k:round(v["ms_per_step"],4) for k,v in (d["config"].get("small_batches_50_steps") or {}).items()})
    except Exception as e: print(f, "ERR", e)
PY
bash profiles/collect.sh ${T} "trace fetch write sq sq2 sq3" > gpurun_out/${T}_collect.log 2>&1
DEDF_SUMMARY_DIR=gpurun_out python profiles/summarize.py ${T} > gpurun_out/${T}_summarize.log 2>&1
CMD="python $GRAFT_REPO_ROOT/bench.py --lmax 3 --steps 5 --warmup 1 --no-cpu-baseline --no-extractors --no-small-batches --no-score-fwd" bash profiles/collect.sh ${T}_lmax3 "trace fetch write sq2" > gpurun_out/${T}_lmax3_collect.log 2>&1
DEDF_SUMMARY_DIR=gpurun_out python profiles/summarize.py ${T}_lmax3 > gpurun_out/${T}_lmax3_summarize.log 2>&1
find gpurun_out -maxdepth 1 -type d -name "${T}*" -exec rm -rf {} +
python - <<'PY'
import json
for t in ("r04zz","r04zz_lmax3"):
    d=json.load(open(f"gpurun_out/{t}_pmc_summary.json"))
    print(t, {k:(round(v,4) if isinstance(v,float) else v) for k,v in d.items() if k!="per_kernel" and not k.endswith("definition")})
PY
du -sh gpurun_out
