# scratch driver of one gpurun call (edited per session): results under gpurun_out/
mkdir -p gpurun_out
T=r04b
python bench.py --no-cpu-baseline > gpurun_out/${T}_bench.json 2> gpurun_out/${T}_bench.err; python - <<'PY'
import json
d=json.loads(open("gpurun_out/r04b_bench.json").read().strip().splitlines()[-1]); r=d["roofline"]
print("C2", d["value"], d["ms_per_step"], "edge", r["avg_launch_ms"], "edges", d["config"]["edges_per_step_rank0"], "frac", r["frac"], "fwd", d["config"]["score_fwd_ms_at_t0.5"])
PY
python tests/probe/unet_torchprof.py 16384 panda_lowres_lmax3 > gpurun_out/${T}_unet_torchprof.log 2>&1; head -45 gpurun_out/${T}_unet_torchprof.log | cut -c1-200
python -m pytest tests -m gpu -q -x > gpurun_out/${T}_gpu_suite.log 2>&1; tail -5 gpurun_out/${T}_gpu_suite.log
