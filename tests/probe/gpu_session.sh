mkdir -p gpurun_out
T=r05s
python -m pytest tests -m gpu -q --durations=5 > gpurun_out/${T}_gpu_suite.log 2>&1
python __graft_entry__.py --smoke > gpurun_out/${T}_smoke.log 2>&1
bash profiles/collect.sh $T "trace fetch write sq sq2 sq3 bench" > gpurun_out/${T}_collect.log 2>&1
DEDF_SUMMARY_DIR=gpurun_out python profiles/summarize.py $T > gpurun_out/${T}_summarize.log 2>&1
CMD="python $GRAFT_REPO_ROOT/bench.py --lmax 3 --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-extractors --no-small-batches --no-score-fwd" bash profiles/collect.sh ${T}_lmax3 "trace fetch write sq2" > gpurun_out/${T}_lmax3_collect.log 2>&1
DEDF_SUMMARY_DIR=gpurun_out python profiles/summarize.py ${T}_lmax3 > gpurun_out/${T}_lmax3_summarize.log 2>&1
CMD="python $GRAFT_REPO_ROOT/bench.py --config5 --gpus 1 --steps 10 --warmup 2 --no-cpu-baseline --no-small-batches --no-score-fwd" bash profiles/collect.sh ${T}_config5 "trace fetch write sq2" > gpurun_out/${T}_config5_collect.log 2>&1
DEDF_SUMMARY_DIR=gpurun_out python profiles/summarize.py ${T}_config5 > gpurun_out/${T}_config5_summarize.log 2>&1
find gpurun_out -maxdepth 1 -type d -name "${T}*" -exec rm -rf {} +
python bench.py --lmax 3 --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/${T}_lmax3_bench.json 2>/dev/null
python bench.py --config5 --steps 10 --warmup 2 --no-cpu-baseline > gpurun_out/${T}_config5_bench.json 2>/dev/null
python bench.py --poses-per-gpu 8000 --steps 10 --warmup 2 --no-cpu-baseline --no-extractors --no-small-batches --no-score-fwd > gpurun_out/${T}_poses8000_bench.json 2>/dev/null
python bench.py --half --steps 20 --warmup 5 --no-cpu-baseline --no-extractors --no-small-batches --no-score-fwd > gpurun_out/${T}_half_bench.json 2>/dev/null
python bench.py --lmax 1 --scene 2048 --grasp 512 --poses-per-gpu 256 --steps 50 --warmup 5 --no-cpu-baseline --no-extractors --no-small-batches --no-score-fwd > gpurun_out/${T}_c1_bench.json 2>/dev/null
python tests/probe/unet_time.py > gpurun_out/${T}_unet_time.log 2>&1
tail -4 gpurun_out/${T}_gpu_suite.log; tail -2 gpurun_out/${T}_smoke.log
python - <<'PY'
import json
for tag in ("r05s", "r05s_lmax3", "r05s_config5"):
    try:
        d=json.load(open(f"gpurun_out/{tag}_pmc_summary.json"))
        print(tag, {k:(round(v,4) if isinstance(v,float) else v) for k,v in d.items() if k!="per_kernel" and not k.endswith("definition") and not isinstance(v, dict)})
    except Exception as e: print(tag, e)
for f in ("r05s_bench","r05s_lmax3_bench","r05s_config5_bench","r05s_poses8000_bench","r05s_half_bench","r05s_c1_bench"):
    try:
        d=json.loads(open(f"gpurun_out/{f}.json").read().strip().splitlines()[-1]); r=d["roofline"]
        print(f, round(d["value"]), round(d["ms_per_step"],3), round(r["avg_launch_ms"],3), round(r["frac"],4), r.get("frac_gemm_executed"), d["config"].get("small_batches_50_steps"), d["config"].get("score_fwd_ms_at_t0.5"), d.get("cpu_baseline"))
    except Exception as e: print(f, e)
PY
tail -5 gpurun_out/${T}_unet_time.log
