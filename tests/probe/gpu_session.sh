# scratch driver of one gpurun call (edited per session): results under gpurun_out/
mkdir -p gpurun_out
T=r04t
python -m pytest tests -m gpu -q -x > gpurun_out/${T}_gpu_suite.log 2>&1; tail -3 gpurun_out/${T}_gpu_suite.log
python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/${T}_smoke.log 2>&1; tail -1 gpurun_out/${T}_smoke.log
python bench.py > gpurun_out/${T}_bench.json 2> gpurun_out/${T}_bench.err
python - <<'PY'
import json
d=json.loads(open("gpurun_out/r04t_bench.json").read().strip().splitlines()[-1]); r=d["roofline"]
print("C2", round(d["value"]), round(d["ms_per_step"],4), "edge", round(r["avg_launch_ms"],4), "frac", round(r["frac"],4), "fwd", d["config"]["score_fwd_ms_at_t0.5"], {k:round(v["ms_per_step"],4) for k,v in d["config"]["small_batches_50_steps"].items()}, "cpu", d["cpu_baseline"]["value"], "traffic", r["traffic"], r["traffic_source"])
PY
