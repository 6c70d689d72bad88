mkdir -p gpurun_out
T=r04u
for n in 2000 4000 8000; do python bench.py --poses-per-gpu $n --no-cpu-baseline --no-extractors --no-small-batches --no-score-fwd --steps 10 > gpurun_out/${T}_poses${n}_bench.json 2>/dev/null; done
python bench.py --config5 --half --no-cpu-baseline --no-small-batches > gpurun_out/${T}_config5_half_bench.json 2> gpurun_out/${T}_config5_half.err
python - <<'PY'
import json,glob
for f in sorted(glob.glob("gpurun_out/r04u_*bench.json")):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1]); r=d["roofline"]
        print(f.split("/")[-1], round(d["value"]), round(d["ms_per_step"],3), "edge", round(r["avg_launch_ms"],3), "edges", round(d["config"]["edges_per_step_rank0"]), "frac", round(r["frac"],4))
    except Exception as e: print(f, "ERR", e)
PY
