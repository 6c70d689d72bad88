# scratch driver of one gpurun call (edited per session): results under gpurun_out/
mkdir -p gpurun_out
run() { DEDF_NODE_BALANCED=$1 python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-extractors --no-small-batches 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); k=d['roofline'].get('kernel_ms_per_step') or {}; print('balanced=$1', round(d['value']), round(d['ms_per_step'],4), {a: round(b,4) for a,b in k.items()})"; }
for i in 1 2 3; do run 0; run 1; done 2>&1 | tee gpurun_out/r03t_node_balanced_ab.log
run3() { DEDF_NODE_BALANCED=$1 python bench.py --lmax 3 --steps 20 --warmup 3 --no-cpu-baseline --no-extractors --no-small-batches 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); k=d['roofline'].get('kernel_ms_per_step') or {}; print('lmax3 balanced=$1', round(d['value']), round(d['ms_per_step'],4), {a: round(b,4) for a,b in k.items()})"; }
for i in 1 2; do run3 0; run3 1; done 2>&1 | tee -a gpurun_out/r03t_node_balanced_ab.log
