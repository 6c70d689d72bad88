import sys, torch
sys.path.insert(0,'/root/repo'); sys.path.insert(0,'/root/repo/tests')
import stage_check as SC
import test_gpu_parity as T
for factor in (30.,100.,1000.):
  for group in list(T._RANGE_GROUPS)+['all_of_them']:
    kw, cfg, P, keys, query, Ts, time = SC.build_case(2, 6, 512, 60)
    names = [g for g in T._RANGE_GROUPS if group in (g, 'all_of_them')]
    P = dict(P)
    for g in names:
        pre = T._RANGE_GROUPS[g]
        if pre is None: continue
        for k in [k for k in P if k.startswith(pre)]: P[k] = P[k]*factor
    if 'key_features' in names: keys = [k._replace(f=k.f*factor) for k in keys]
    if 'query_features' in names: query = query._replace(f=query.f*factor)
    ang64, lin64, d64, _ = SC.oracle_run(kw, P, keys, query, Ts, time, torch.float64)
    head, ang, lin = SC.gpu_run(kw, P, keys, query, Ts, time, debug=False)
    st = head.stats()
    scale = float(max(ang64.abs().max(), lin64.abs().max()))
    err = max(float((ang.double() - ang64).abs().max()), float((lin.double() - lin64).abs().max())) / scale
    print(factor, group, 'nonfinite', st['nonfinite'], 'err', err, 'scale', scale, flush=True)
