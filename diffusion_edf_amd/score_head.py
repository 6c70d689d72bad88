"""``ScoreModelHead`` — drop-in for reference ``diffusion_edf/score_head.py:18-252`` on MI355X.

Same constructor keywords (the ``score_head_kwargs`` block of the reference YAML configs), same parameter names
(``state_dict`` keys equal the reference's below ``score_head.``), same ``forward(Ts, key_pcd_multiscale, query_pcd,
time) -> (ang_vel, lin_vel)`` contract, ``warmup``, ``lin_mult`` / ``ang_mult`` / ``n_scales`` attributes and
``jittable = False`` so that reference ``agent.py:53-55`` skips TorchScript.  The arithmetic runs in libdedf.so
(hand-written HIP for gfx950); there is no PyTorch/CPU fallback — construction fails loudly without the library/GPU.
"""
from __future__ import annotations

import ctypes as C
from typing import Dict, List, Optional, Sequence, Tuple, Union

import torch

from . import _lib
from .gnn_data import FeaturedPoints
from .params import HeadConfig, init_params, param_spec


class _Node(torch.nn.Module):
    """anonymous container so that parameter names follow the reference module tree"""


def _register(root: torch.nn.Module, dotted: str, value: torch.Tensor):
    parts = dotted.split('.')
    mod = root
    for p in parts[:-1]:
        if not hasattr(mod, p):
            mod.add_module(p, _Node())
        mod = getattr(mod, p)
    mod.register_parameter(parts[-1], torch.nn.Parameter(value, requires_grad=False))


class ScoreModelHead(torch.nn.Module):
    jittable: bool = False

    def __init__(self,
                 max_time: float,
                 time_emb_mlp: List[int],
                 key_tensor_field_kwargs: Dict,
                 irreps_query_edf,
                 lin_mult: float,
                 ang_mult: float,
                 time_enc_n: float = 10000.,
                 edge_time_encoding: bool = False,
                 query_time_encoding: bool = True,
                 device: Union[str, torch.device, None] = None,
                 init_seed: int = 2,
                 max_edges: int = 0,
                 ebm: bool = False,
                 half_precision: bool = False):
        super().__init__()
        kw = dict(ebm=ebm, max_time=max_time, time_emb_mlp=list(time_emb_mlp), key_tensor_field_kwargs=dict(key_tensor_field_kwargs),
                  irreps_query_edf=irreps_query_edf, lin_mult=lin_mult, ang_mult=ang_mult, time_enc_n=time_enc_n,
                  edge_time_encoding=edge_time_encoding, query_time_encoding=query_time_encoding)
        if not edge_time_encoding and not query_time_encoding and not ebm:
            raise NotImplementedError("No time encoding! Are you sure?")          # reference score_head.py:72-73
        self.cfg = HeadConfig.from_kwargs(kw)
        self.cfg.half_gemm = bool(half_precision)
        self.lin_mult, self.ang_mult = float(lin_mult), float(ang_mult)
        self.n_scales = self.cfg.n_scales
        self.max_time = float(max_time)
        self.time_emb_mlp = list(time_emb_mlp)
        self.key_edf_dim = self.query_edf_dim = self.cfg.dim
        self.n_irreps_prescore = self.cfg.muls[1]
        self.edge_time_encoding, self.query_time_encoding = edge_time_encoding, query_time_encoding
        self.irreps_key_edf = '+'.join(f"{m}x{l}e" for m, l in self.cfg.irreps)
        self._max_edges = int(max_edges)
        for name, t in init_params(self.cfg, seed=init_seed).items():
            _register(self, name, t)
        self.register_load_state_dict_post_hook(ScoreModelHead._after_load)
        self._handle = None
        self._handle_device: Optional[torch.device] = None
        self._scene_key = None
        self._query_key = None
        if device is not None:
            self.to(device)

    # ------------------------------------------------------------------------------------------------------------------
    def _state(self) -> Dict[str, torch.Tensor]:
        return {k: v for k, v in self.state_dict().items()}

    def _ensure_handle(self, device: torch.device):
        if self._handle is not None and self._handle_device == device:
            return
        self._release()
        if device.type != 'cuda':
            raise RuntimeError("diffusion_edf_amd.ScoreModelHead runs on an MI355X (torch device 'cuda') only; "
                               "there is no CPU path (the CPU restatement lives under oracle/ and is test infrastructure).")
        lib = _lib.load()
        idx = device.index if device.index is not None else torch.cuda.current_device()
        ccfg = _lib.make_config(self.cfg, idx, self._max_edges)
        blob = _lib.pack_params(ccfg, self._state())
        h = C.c_void_p()
        rc = lib.dedf_create(C.byref(ccfg), blob.ctypes.data_as(C.POINTER(C.c_float)), blob.size, C.byref(h))
        if rc != _lib.OK:
            _lib.raise_for(lib, None, rc, "dedf_create failed (see stderr)")
        self._handle, self._handle_device = h, device
        self._scene_key = self._query_key = None
        if getattr(self, "_radial_table", None) is not None:
            lib.dedf_set_radial_table(h, int(self._radial_table))

    def _release(self):
        if self._handle is not None:
            _lib.load().dedf_destroy(self._handle)
            self._handle = None

    def half(self):
        """``model.half()`` is how the reference switches to half precision (agent.py:50-51).  Here it selects the half-precision
        GEMM mode of the kernels (every GEMM one fp16 MFMA product with fp32 accumulation instead of the 3-term split; tensor
        products, norms, softmax and the SE(3) update stay fp32).  Parameters are kept as fp32 master copies; fp16 inputs are
        accepted at the boundary."""
        self.cfg.half_gemm = True
        self._release()
        return self

    def float(self):
        self.cfg.half_gemm = False
        self._release()
        return super().float()

    def set_radial_table(self, on: bool):
        """the sampler's radial table (``dedf_set_radial_table``, on by default): when every pose of a step shares the diffusion time, the
        front of the radial network is tabulated once per step on a fine length grid and interpolated per edge; ``True`` = automatic (batches of at least 8 192 pose x query nodes), ``"always"`` regardless of the batch size, ``False`` evaluates it per
        edge everywhere (what ``forward`` always does: it takes one time per pose)"""
        self._radial_table = 2 if on == "always" else int(bool(on))          # "always": also for batches too small for it to pay (tests)
        if self._handle is not None:
            _lib.load().dedf_set_radial_table(self._handle, self._radial_table)

    def refresh_weights(self):
        """call after load_state_dict(): the packed device image is rebuilt on next use"""
        self._release()

    def _after_load(self, *_):          # post hook: fires for this module's own load_state_dict AND when a parent module loads a checkpoint
        self.refresh_weights()

    def __del__(self):
        try:
            self._release()
        except Exception:
            pass

    # ------------------------------------------------------------------------------------------------------------------
    def _stream(self) -> C.c_void_p:
        """torch's current stream ON THE HANDLE'S DEVICE (which need not be the current device)"""
        return C.c_void_p(torch.cuda.current_stream(self._handle_device).cuda_stream)

    def set_key_clouds(self, key_pcd_multiscale: Sequence[FeaturedPoints]):
        assert len(key_pcd_multiscale) == self.n_scales
        dev = key_pcd_multiscale[0].x.device
        self._ensure_handle(dev)
        lib = _lib.load()
        xs = [p.x.detach().to(torch.float32).contiguous() for p in key_pcd_multiscale]
        fs = [p.f.detach().to(torch.float32).contiguous() for p in key_pcd_multiscale]
        for x, f in zip(xs, fs):
            assert x.ndim == 2 and x.shape[-1] == 3, f"{x.shape}"
            assert f.ndim == 2 and f.shape[-1] == self.key_edf_dim and len(f) == len(x), f"{f.shape}"
        n = len(xs)
        npts = (C.c_int * n)(*[len(x) for x in xs])
        xp = (C.c_void_p * n)(*[x.data_ptr() for x in xs])
        fp = (C.c_void_p * n)(*[f.data_ptr() for f in fs])
        rc = lib.dedf_set_key_clouds(self._handle, n, npts, xp, fp, self._stream())
        _lib.raise_for(lib, self._handle, rc, "dedf_set_key_clouds")
        if self.cfg.use_src_point_attn:              # PointAttentiveScoreModel: alpha *= src_points.w[edge_src]  (gnn_block.py:190-194)
            for p in key_pcd_multiscale:
                assert isinstance(p.w, torch.Tensor)                                     # gnn_block.py:192
                assert p.w.ndim == 1 and len(p.w) == len(p.x), f"{p.w.shape}"
            ws = [p.w.detach().to(torch.float32).contiguous() for p in key_pcd_multiscale]
            wp = (C.c_void_p * n)(*[w.data_ptr() for w in ws])
            rc = lib.dedf_set_key_weights(self._handle, n, npts, wp, self._stream())
            _lib.raise_for(lib, self._handle, rc, "dedf_set_key_weights")
        self._scene_key = self._refs(self._key_of(key_pcd_multiscale))

    # Upload cache.  An entry holds STRONG references to the caller's tensors next to their version counters: while a cloud is
    # cached its storage cannot be handed to another tensor by the caching allocator, so "same object, same version" really
    # means "same contents" (a key made of data_ptr()s alone can collide with the next scene's freshly allocated tensors).
    # In-place writes that bypass the version counter (.data, custom kernels, dlpack) are not seen: call set_key_clouds /
    # set_query explicitly after such writes.
    @staticmethod
    def _refs(tensors):
        return tuple((t, t._version) for t in tensors)

    @staticmethod
    def _same(refs, tensors) -> bool:
        return refs is not None and len(refs) == len(tensors) and all(r[0] is t and r[1] == t._version for r, t in zip(refs, tensors))

    def _key_of(self, key_pcd_multiscale):
        ts = []
        for p in key_pcd_multiscale:
            ts += [p.x, p.f]
            if self.cfg.use_src_point_attn and isinstance(p.w, torch.Tensor):
                ts.append(p.w)
        return ts

    @staticmethod
    def _query_tensors(query_pcd):
        return [query_pcd.x, query_pcd.f, query_pcd.w]

    def set_query(self, query_pcd: FeaturedPoints):
        dev = query_pcd.x.device
        self._ensure_handle(dev)
        lib = _lib.load()
        assert query_pcd.f.ndim == 2 and query_pcd.f.shape[-1] == self.query_edf_dim, f"{query_pcd.f.shape}"
        w = query_pcd.w
        assert isinstance(w, torch.Tensor)
        x = query_pcd.x.detach().to(torch.float32).contiguous()
        f = query_pcd.f.detach().to(torch.float32).contiguous()
        w = w.detach().to(torch.float32).contiguous()
        rc = lib.dedf_set_query(self._handle, len(x), x.data_ptr(), f.data_ptr(), w.data_ptr(), self._stream())
        _lib.raise_for(lib, self._handle, rc, "dedf_set_query")
        self._query_key = self._refs(self._query_tensors(query_pcd))

    def _sync_inputs(self, key_pcd_multiscale, query_pcd):
        if self._handle is None or not self._same(self._scene_key, self._key_of(key_pcd_multiscale)):
            self.set_key_clouds(key_pcd_multiscale)
        if not isinstance(query_pcd.w, torch.Tensor) or not self._same(self._query_key, self._query_tensors(query_pcd)):
            self.set_query(query_pcd)

    @torch.no_grad()
    def forward(self, Ts: torch.Tensor, key_pcd_multiscale: List[FeaturedPoints], query_pcd: FeaturedPoints,
                time: torch.Tensor) -> Tuple[torch.Tensor, torch.Tensor]:
        """``(ang, lin)`` like reference ``score_head.py:142-211``.  The call is enqueued on the current stream and never synchronises.
        If an evaluation overflows its edge workspace or an operand leaves the fp16 window of the split GEMMs, its outputs are NaN
        and the NEXT call on this module that finds the evaluation complete (``forward``, ``set_key_clouds`` / ``set_query``,
        ``sample``) raises ``RuntimeError`` for it -- unless ``stats()`` was read in between (then the caller has seen
        ``overflow`` / ``nonfinite`` himself).  A batch whose poses all share one time takes the sampler's radial table
        (scores differ from the mixed-time evaluation of the same pose by ~1e-5 of the score scale, both within the tolerance)."""
        assert Ts.ndim == 2 and Ts.shape[-1] == 7, f"{Ts.shape}"                      # reference score_head.py:149
        assert time.ndim == 1 and len(time) == len(Ts), f"{time.shape}"               # :150
        assert query_pcd.f.ndim == 2 and query_pcd.f.shape[-1] == self.query_edf_dim, f"{query_pcd.f.shape}"   # :151
        self._sync_inputs(key_pcd_multiscale, query_pcd)
        lib = _lib.load()
        nT = len(Ts)
        Ts32 = Ts.detach().to(torch.float32).contiguous()
        t32 = time.detach().to(torch.float32).contiguous()
        ang = torch.empty(nT, 3, device=Ts.device, dtype=torch.float32)
        lin = torch.empty(nT, 3, device=Ts.device, dtype=torch.float32)
        rc = lib.dedf_score(self._handle, nT, Ts32.data_ptr(), t32.data_ptr(), ang.data_ptr(), lin.data_ptr(), self._stream())
        _lib.raise_for(lib, self._handle, rc, "dedf_score")
        return ang.to(Ts.dtype), lin.to(Ts.dtype)

    def warmup(self, Ts, key_pcd_multiscale, query_pcd, time):                         # reference score_head.py:213-218
        return self.forward(Ts=Ts, key_pcd_multiscale=key_pcd_multiscale, query_pcd=query_pcd, time=time)

    def _get_fake_input(self):
        """Random inputs of the reference's shapes (score_head.py:220-246: 5 poses, 100 key points per scale, 10 query points);
        `Irreps.randn` draws N(0,1) per component, `random_quaternions` a normalised N(0,1)^4 with w >= 0 (transforms.py:349-355)."""
        device = next(iter(self.parameters())).device
        nT, nP, nQ = 5, 100, 10
        q = torch.randn(nT, 4, device=device)
        q = q / q.norm(dim=-1, keepdim=True)
        q = torch.where(q[:, :1] < 0, -q, q)
        Ts = torch.cat([q, torch.randn(nT, 3, device=device)], dim=-1)
        time = torch.rand(nT, device=device)
        key_pcd_multiscale = [FeaturedPoints(x=torch.randn(nP, 3, device=device), f=torch.randn(nP, self.key_edf_dim, device=device),
                                             b=torch.zeros(nP, device=device, dtype=torch.long),
                                             w=torch.ones(nP, device=device) if self.cfg.use_src_point_attn else None)
                              for _ in range(self.n_scales)]
        query_pcd = FeaturedPoints(x=torch.randn(nQ, 3, device=device), f=torch.randn(nQ, self.query_edf_dim, device=device),
                                   b=torch.zeros(nQ, device=device, dtype=torch.long), w=torch.ones(nQ, device=device))
        return Ts, key_pcd_multiscale, query_pcd, time

    # ------------------------------------------------------------------------------------------------------------------
    def stats(self) -> dict:
        lib = _lib.load()
        st = _lib.DedfStats()
        rc = lib.dedf_get_stats(self._handle, C.byref(st))
        _lib.raise_for(lib, self._handle, rc, "dedf_get_stats")
        return dict(n_dst=st.n_dst, n_edges=[st.n_edges[i] for i in range(self.n_scales)], n_edges_total=st.n_edges_total,
                    overflow=bool(st.overflow), nonfinite=bool(st.nonfinite),
                    sample_retries=int(st.sample_retries), edges_per_dst_capacity=int(st.edges_per_dst_capacity),
                    # accuracy guard of the sampler's radial table: largest interpolation error per scale, scales that fell back to per-edge
                    rtab_err=[float(st.rtab_err[i]) for i in range(self.n_scales)],
                    rtab_fallback=[bool(st.rtab_fallback >> i & 1) for i in range(self.n_scales)])

    def profile_enable(self, on: bool = True):
        lib = _lib.load()
        _lib.raise_for(lib, self._handle, lib.dedf_profile_enable(self._handle, int(on)), "dedf_profile_enable")

    def profile_read(self) -> dict:
        """summed HIP-event kernel times [ms] per kernel class since the last read (resets the counters)"""
        lib = _lib.load()
        pr = _lib.DedfProfile()
        _lib.raise_for(lib, self._handle, lib.dedf_profile_read(self._handle, C.byref(pr)), "dedf_profile_read")
        return dict(n_evals=pr.n_evals, n_edges=pr.n_edges, n_dst=pr.n_dst, ms={k: pr.ms[i] for i, k in enumerate(_lib.PROF_CLASSES)})

    def debug_enable(self, on: bool = True):
        _lib.load().dedf_debug_enable(self._handle, int(on))

    def debug_buffer(self, name: str, dtype=torch.float32) -> torch.Tensor:
        lib = _lib.load()
        n = C.c_size_t(0)
        rc = lib.dedf_debug_copy(self._handle, name.encode(), None, 0, C.byref(n))
        _lib.raise_for(lib, self._handle, rc, "dedf_debug_copy")
        out = torch.empty(n.value // 4, dtype=torch.float32 if dtype == torch.float32 else torch.int32)
        if n.value:
            rc = lib.dedf_debug_copy(self._handle, name.encode(), out.data_ptr(), n.value, C.byref(n))
            _lib.raise_for(lib, self._handle, rc, "dedf_debug_copy")
        return out


class EbmScoreModelHead(ScoreModelHead):
    """Energy-based critic head — drop-in for ``compute_energy`` of reference ``diffusion_edf/score_head_ebm.py:18-255``
    (what ``agent.py:163-174`` calls to rank the sampled poses).  Same constructor keywords and parameter names (no
    ``lin_vel_tp`` / ``ang_vel_tp``); ``forward`` (score = autograd of the energy, ``score_head_ebm.py:192-222``) needs a
    backward pass and is not on the accelerated path."""
    jittable: bool = False

    def __init__(self, max_time: float, time_emb_mlp: List[int], key_tensor_field_kwargs: Dict, irreps_query_edf,
                 lin_mult: float, ang_mult: float, time_enc_n: float = 10000., edge_time_encoding: bool = False,
                 query_time_encoding: bool = True, **kw):
        kw.pop('ebm', None)
        super().__init__(max_time=max_time, time_emb_mlp=time_emb_mlp, key_tensor_field_kwargs=key_tensor_field_kwargs,
                         irreps_query_edf=irreps_query_edf, lin_mult=lin_mult, ang_mult=ang_mult, time_enc_n=time_enc_n,
                         edge_time_encoding=edge_time_encoding, query_time_encoding=query_time_encoding, ebm=True, **kw)
        self.energy_rescale_factor = 1. / float(self.key_edf_dim)

    @torch.no_grad()
    def compute_energy(self, Ts: torch.Tensor, key_pcd_multiscale: List[FeaturedPoints], query_pcd: FeaturedPoints,
                       time: torch.Tensor) -> torch.Tensor:
        assert Ts.ndim == 2 and Ts.shape[-1] == 7, f"{Ts.shape}"                      # reference score_head_ebm.py:129
        assert time.ndim == 1 and len(time) == len(Ts), f"{time.shape}"               # :130
        assert query_pcd.f.ndim == 2 and query_pcd.f.shape[-1] == self.query_edf_dim, f"{query_pcd.f.shape}"   # :131
        self._sync_inputs(key_pcd_multiscale, query_pcd)
        lib = _lib.load()
        nT = len(Ts)
        Ts32 = Ts.detach().to(torch.float32).contiguous()
        t32 = time.detach().to(torch.float32).contiguous()
        energy = torch.empty(nT, device=Ts.device, dtype=torch.float32)
        rc = lib.dedf_energy(self._handle, nT, Ts32.data_ptr(), t32.data_ptr(), energy.data_ptr(), self._stream())
        _lib.raise_for(lib, self._handle, rc, "dedf_energy")
        return energy.to(Ts.dtype)

    def warmup(self, Ts, key_pcd_multiscale, query_pcd, time):                         # reference score_head_ebm.py:176-181
        return self.compute_energy(Ts=Ts, key_pcd_multiscale=key_pcd_multiscale, query_pcd=query_pcd, time=time)

    def forward(self, Ts, key_pcd_multiscale, query_pcd, time):
        raise NotImplementedError("EbmScoreModelHead.forward is the autograd of compute_energy (reference score_head_ebm.py:"
                                  "203-217); only compute_energy (the inference-time critic) is on the accelerated path")
