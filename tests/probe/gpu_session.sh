# scratch driver of one gpurun call (edited per session): results under gpurun_out/
mkdir -p gpurun_out
T=r04zy
timeout 2400 python -m pytest tests -m gpu -x -q > gpurun_out/${T}_gpu_suite.log 2>&1; tail -3 gpurun_out/${T}_gpu_suite.log
python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/${T}_smoke.log 2>&1; tail -1 gpurun_out/${T}_smoke.log
python bench.py > gpurun_out/${T}_bench.json 2> gpurun_out/${T}_bench.err
python bench.py --lmax 3 --no-cpu-baseline > gpurun_out/${T}_lmax3_bench.json 2> /dev/null
python bench.py --config5 --no-cpu-baseline > gpurun_out/${T}_config5_bench.json 2> /dev/null
python - <<'PY'
import json
for f in ("", "lmax3_", "config5_"):
    try:
        d=json.loads(open(f"gpurun_out/r04zy_{f}bench.json").read().strip().splitlines()[-1]); r=d["roofline"]
        print(f or "C2", round(d["value"]), round(d["ms_per_step"],4), "edge", round(r["avg_launch_ms"],4), "edges", round(d["config"]["edges_per_step_rank0"]), "frac", round(r["frac"],4), "fwd", d["config"].get("score_fwd_ms_at_t0.5"), "ext", d["config"].get("feature_extractors_ms"))
    except Exception as e: print(f, "ERR", e)
PY
for r in 1 2; do python tests/probe/unet_time.py 16384 10 2>&1 | grep "Extractor"; done > gpurun_out/${T}_unet_time.log; cat gpurun_out/${T}_unet_time.log
