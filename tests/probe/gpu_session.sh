# scratch driver of one gpurun call (edited per session): results under gpurun_out/
mkdir -p gpurun_out
T=r05b
{
echo "== stage check, SO2 (default) =="; python tests/stage_check.py 2 4 512 60
echo "== stage check, general form (DEDF_SO2=0) =="; DEDF_SO2=0 python tests/stage_check.py 2 4 512 60
echo "== smoke =="; python __graft_entry__.py --smoke
} > gpurun_out/${T}_so2_stage.log 2>&1
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "score_parity or sampler_parity or c2_timed or radial_table or one_time or anchored or tiny or empty or point_attent or overflow or workspace" > gpurun_out/${T}_so2_tests.log 2>&1
for s in 1 0 1 0; do DEDF_SO2=$s python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extractors --no-small-batches --no-score-fwd | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']; print('SO2=$s', round(d['value']), d['ms_per_step'], r['avg_launch_ms'], r['frac'])"; done > gpurun_out/${T}_so2_ab.log 2>&1
python - > gpurun_out/${T}_oracle_threads.log 2>&1 <<'PY'
import time, torch, numpy as np, sys, os
sys.path.insert(0, "tests")
from diffusion_edf_amd import synthetic
from oracle import restatement as R, unet_oracle as U
from diffusion_edf_amd.so3 import parse_irreps
from diffusion_edf_amd.unet import UnetFeatureExtractor
from test_lmax3 import _randomized, SH3
print("default threads", torch.get_num_threads(), "cpus", os.cpu_count())
kwu = synthetic.unet_kwargs("panda_lowres_lmax3")
m = UnetFeatureExtractor(**kwu, deterministic=True)
sd = _randomized(m, seed=5)
kw = m._ctor
ocfg = U.UnetConfig(irreps_input=parse_irreps(kw["irreps_input"]), irreps_output=parse_irreps(kw["irreps_output"]),
                    irreps_emb=[parse_irreps(i) for i in kw["irreps_emb"]], fc_neurons=[list(f) for f in kw["fc_neurons"]],
                    n_layers=list(kw["n_layers"]), pool_ratio=list(kw["pool_ratio"]), radius=list(m.radius),
                    n_layers_midstream=kw["n_layers_midstream"], irreps_sh=SH3)
n = 4096
x = torch.from_numpy(synthetic.make_scene(n, seed=0).astype(np.float32))
f = torch.rand(n, 3, generator=torch.Generator().manual_seed(1))
P = R.cast_params(sd, torch.float64)
for thr in (None, 64, 32, 16, 8):
    if thr: torch.set_num_threads(thr)
    t0 = time.time(); U.unet_forward(ocfg, P, x, f.double()); print("threads", thr or "default", round(time.time() - t0, 2), "s")
PY
tail -3 gpurun_out/${T}_so2_stage.log; tail -3 gpurun_out/${T}_so2_tests.log; cat gpurun_out/${T}_so2_ab.log; cat gpurun_out/${T}_oracle_threads.log | tail -6
