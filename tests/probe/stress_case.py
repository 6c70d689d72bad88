"""Re-run ONE case of tests/stress_parity.py:  python tests/probe/stress_case.py <seed> <index> [stage]   (environment switches apply)"""
import os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import stress_parity as S
import stage_check as SC
seed, idx = int(sys.argv[1]), int(sys.argv[2])
rng = np.random.default_rng(seed)
for i in range(idx):
    S.draw_case(rng)
if len(sys.argv) > 3 and sys.argv[3] == "stage":
    kw, cfg, P, keys, query, Ts, time = S.draw_case(rng)
    import oracle.restatement as R
    orig = R.config_from_kwargs
    R.config_from_kwargs = lambda k: orig(k)._replace(max_neighbors=cfg.max_neighbors)      # (the oracle takes the cap from its Config)
    from diffusion_edf_amd import score_head as SH
    init = SH.ScoreModelHead.__init__
    def patched(self, *a, **k2):
        init(self, *a, **k2); self.cfg.max_neighbors = cfg.max_neighbors
    SH.ScoreModelHead.__init__ = patched
    rep = SC.stage_report_case(kw, cfg, P, keys, query, Ts, time, verbose=False)
    print({k: (f"{v:.2e}" if isinstance(v, float) else v) for k, v in rep.items()})
else:
    err, ok, desc = S.run_case(idx, rng)
    print("RESULT", os.environ.get("TAG", ""), f"{err:.3e}", ok, desc)
