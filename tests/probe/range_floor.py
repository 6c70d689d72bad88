"""fp32 floor of the scaled-weight cases of tests/test_gpu_parity.py::test_fp16_operand_range_scaled_weights_and_features: the error of the fp32
restatement against the fp64 one on the same scaled inputs, next to the HIP path's."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), "tests"))
import stage_check as SC
from test_gpu_parity import _RANGE_GROUPS
torch.set_num_threads(16)
cases = [(g, f) for g in list(_RANGE_GROUPS) + ['all_of_them'] for f in (30.0, 100.0, 1000.0, 0.01)]
if len(sys.argv) > 1:
    cases = [(a.split(':')[0], float(a.split(':')[1])) for a in sys.argv[1:]]
for group, factor in cases:
    kw, cfg, P, keys, query, Ts, time = SC.build_case(2, 6, 512, 60)
    names = [g for g in _RANGE_GROUPS if group in (g, 'all_of_them')]
    P = dict(P)
    for g in names:
        pre = _RANGE_GROUPS[g]
        if pre is None:
            continue
        for k in [k for k in P if k.startswith(pre)]:
            P[k] = P[k] * factor
    if 'key_features' in names:
        keys = [k._replace(f=k.f * factor) for k in keys]
    if 'query_features' in names:
        query = query._replace(f=query.f * factor)
    a64, l64, _, _ = SC.oracle_run(kw, P, keys, query, Ts, time, torch.float64)
    a32, l32, _, _ = SC.oracle_run(kw, P, keys, query, Ts, time, torch.float32)
    scale = float(max(a64.abs().max(), l64.abs().max()))
    e32 = max(float((a32.double() - a64).abs().max()), float((l32.double() - l64).abs().max())) / scale
    head, ang, lin = SC.gpu_run(kw, P, keys, query, Ts, time, debug=False)
    st = head.stats()
    eg = max(float((ang.double().cpu() - a64).abs().max()), float((lin.double().cpu() - l64).abs().max())) / scale
    print(f"{group:18s} x{factor:<7g} score scale {scale:9.3g}  fp32 restatement {e32:.2e}  HIP {eg:.2e}  nonfinite {st['nonfinite']}", flush=True)
