"""``ScoreModelBase.sample`` — drop-in for reference ``diffusion_edf/score_model_base.py:110-204``.

Same signature and return value (``(1 + sum(N_steps) + 1, nT, 7)`` float64 trajectory: seed, every step, final pose
again).  The whole Langevin loop (score evaluation in f32, SE(3) update in f64) runs inside ``dedf_sample``; the host only
builds the per-step schedule exactly as the reference does (torch.logspace/linspace in float64).

Extra keyword arguments (not in the reference): ``noise`` — injected standard normals ``(sum N_steps, 2, nT, 3)`` f64 for
parity runs; ``seed`` / ``first_pose_index`` — counter-based Philox stream keyed by the *global* pose index so that a
pose-sharded multi-GPU run draws the same noise as a single-GPU run.
"""
from __future__ import annotations

import ctypes as C
import warnings
from typing import List, Optional, Sequence, Tuple, Union

import numpy as np
import torch

from . import _lib
from .gnn_data import FeaturedPoints
from .score_head import ScoreModelHead


_SCHEDULE_CACHE: dict = {}


def build_schedule(ang_mult: float, lin_mult: float, diffusion_schedules, N_steps, timesteps, temperatures=1.0,
                   log_t_schedule: bool = True, time_exponent_temp: float = 0.5, time_exponent_alpha: float = 0.5):
    """Per-step (t, alpha_ang, alpha_lin, temperature) in float64 — reference score_model_base.py:131-171.
    The powers are torch's own scalar float64 ``pow`` (its vectorised form and libm's differ from it in the last bit for a few per cent of
    the arguments); the products around them are plain IEEE double multiplications in the reference's order, done on Python floats.  A
    deployment calls ``sample`` with the same schedule over and over: results are memoised."""
    if isinstance(temperatures, (int, float)):
        temperatures = [float(temperatures) for _ in range(len(diffusion_schedules))]
    key = (float(ang_mult), float(lin_mult), tuple(tuple(float(v) for v in s_) for s_ in diffusion_schedules), tuple(int(n) for n in N_steps),
           tuple(float(v) for v in timesteps), tuple(float(v) for v in temperatures), bool(log_t_schedule), float(time_exponent_temp), float(time_exponent_alpha))
    hit = _SCHEDULE_CACHE.get(key)
    if hit is not None:
        return hit
    temperatures = torch.tensor(temperatures, dtype=torch.float64)
    sched = torch.tensor(diffusion_schedules, dtype=torch.float64)
    ts, aa, al, tt = [], [], [], []
    for n, schedule in enumerate(sched):
        if log_t_schedule:
            t_schedule = torch.logspace(start=torch.log(schedule[0]), end=torch.log(schedule[1]), steps=N_steps[n],
                                        base=torch.e, dtype=torch.float64)
        else:
            t_schedule = torch.linspace(start=schedule[0], end=schedule[1], steps=N_steps[n], dtype=torch.float64)
        temp_n, dt_n = float(temperatures[n]), timesteps[n]
        for t in t_schedule.unbind(0):
            p_alpha = float(torch.pow(t, time_exponent_alpha))
            p_temp = p_alpha if time_exponent_temp == time_exponent_alpha else float(torch.pow(t, time_exponent_temp))
            ts.append(float(t))
            tt.append(temp_n * p_temp)
            aa.append((ang_mult ** 2) * p_alpha * dt_n)
            al.append((lin_mult ** 2) * p_alpha * dt_n)
    out = (np.asarray(ts, dtype=np.float64), np.asarray(aa, dtype=np.float64), np.asarray(al, dtype=np.float64), np.asarray(tt, dtype=np.float64))
    for a in out:
        a.setflags(write=False)          # the memoised arrays are shared by every later caller: an in-place edit must raise, not corrupt them
    if len(_SCHEDULE_CACHE) < 64:
        _SCHEDULE_CACHE[key] = out
    return out


class ScoreModelBase(torch.nn.Module):
    """Holds a ``score_head``; the subclasses in ``agent.py`` add the key and query feature extractors (``unet.py``, ``keypoint_extractor.py``)."""

    def __init__(self, score_head: Optional[ScoreModelHead] = None):
        super().__init__()
        if score_head is not None:
            self.score_head = score_head
            self.lin_mult, self.ang_mult = score_head.lin_mult, score_head.ang_mult

    @torch.no_grad()
    def sample(self, T_seed: torch.Tensor,
               scene_pcd_multiscale: List[FeaturedPoints],
               grasp_pcd: FeaturedPoints,
               diffusion_schedules: List[Union[List[float], Tuple[float, float]]],
               N_steps: List[int],
               timesteps: List[float],
               temperatures: Union[Union[int, float], Sequence[Union[int, float]]] = 1.0,
               log_t_schedule: bool = True,
               time_exponent_temp: float = 0.5,
               time_exponent_alpha: float = 0.5,
               noise: Optional[torch.Tensor] = None,
               seed: int = 0,
               first_pose_index: int = 0) -> torch.Tensor:
        head: ScoreModelHead = self.score_head
        assert T_seed.ndim == 2 and T_seed.shape[-1] == 7, f"{T_seed.shape}"
        device = T_seed.device
        head._sync_inputs(scene_pcd_multiscale, grasp_pcd)
        lib = _lib.load()
        t, aa, al, tt = build_schedule(head.ang_mult, head.lin_mult, diffusion_schedules, N_steps, timesteps, temperatures,
                                       log_t_schedule, time_exponent_temp, time_exponent_alpha)
        n_steps, nT = len(t), len(T_seed)
        if nT == 0:          # an empty pose block (a rank with no poses in a sharded run): nothing to launch, same output shape
            return torch.empty(n_steps + 2, 0, 7, device=device, dtype=torch.float64)
        sch = _lib.DedfSchedule()
        sch.n_steps = n_steps
        dp = C.POINTER(C.c_double)
        sch.t, sch.alpha_ang, sch.alpha_lin, sch.temperature = (a.ctypes.data_as(dp) for a in (t, aa, al, tt))
        T64 = T_seed.detach().to(torch.float64).contiguous()
        out = torch.empty(n_steps + 2, nT, 7, device=device, dtype=torch.float64)
        nz = None
        if noise is not None:
            assert tuple(noise.shape) == (n_steps, 2, nT, 3), f"{noise.shape}"
            nz = noise.detach().to(device=device, dtype=torch.float64).contiguous()
        rc = lib.dedf_sample(head._handle, nT, T64.data_ptr(), C.byref(sch), C.c_uint64(seed), C.c_int64(first_pose_index),
                             nz.data_ptr() if nz is not None else None, out.data_ptr(), head._stream())
        _lib.raise_for(lib, head._handle, rc, "dedf_sample")
        # dedf_sample has synchronised: the reference's zero-edge warning (multiscale_tensor_field.py:249-250) costs nothing here.
        # (`forward` stays free of host round trips and does not warn.)
        if n_steps > 0 and head.stats()['n_edges_total'] == 0:
            warnings.warn("Multiscale Tensor Field: zero edges detected!")
        return out

    @torch.no_grad()
    def forward(self, Ts: torch.Tensor, time: torch.Tensor, key_pcd_multiscale: List[FeaturedPoints], query_pcd: FeaturedPoints):
        """score only (reference score_model_base.py:206-225 without the out-of-scope feature extractors)"""
        return self.score_head(Ts=Ts, key_pcd_multiscale=key_pcd_multiscale, query_pcd=query_pcd, time=time)
