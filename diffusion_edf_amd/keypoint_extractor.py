"""Query-side model of the place tasks: ``KeypointExtractor`` and the context-free ``MultiscaleTensorField`` it is made of.

Reference ``diffusion_edf/keypoint_extractor.py:50-197``: the grasped object's cloud goes through a ``UnetFeatureExtractor``; key points are
picked from the input cloud by farthest point sampling (inside an optional bounding box); two ``MultiscaleTensorField`` s evaluated at the key
points give their equivariant descriptors (``tensor_field``, output irreps = the EDF irreps) and a scalar embedding (``weight_field``, output
``64x0e``) that ``weight_post`` (LayerNorm, SiLU, Linear, Sigmoid) turns into the point weights ``w`` of the query EDF.

Everything numeric runs in ``libdedf.so``: the UNet (``unet.py``), FPS (``dedf_fps``), the fields (``dedf_field`` on an EBM-type handle: same
fused edge / aggregation / node kernels as the score head, no time encoding) and the weight head (``dedf_keypoint_weight``).  GPU only.

``StaticKeypointModel`` (``keypoint_extractor.py:22-47``, the pick tasks' query model) holds learned key points and has no arithmetic beyond a
sigmoid; it is here so that both ``query_model`` names of the YAML files resolve.
"""
from __future__ import annotations

import ctypes as C
import math
from typing import Dict, List, Optional, Sequence, Union

import torch

from . import _lib
from .gnn_data import FeaturedPoints
from .params import HeadConfig, init_params
from .score_head import _register
from .so3 import irreps_dim, parse_irreps

_KTF = "key_tensor_field."


class MultiscaleTensorField(torch.nn.Module):
    """reference ``multiscale_tensor_field.py:19-260`` for ``irreps_query=None`` and ``edge_context_emb_dim=None`` (how KeypointExtractor builds its
    two fields, ``keypoint_extractor.py:97-112``): same constructor keywords, same parameter names (``graph_parsers.N.length_enc.param_module.*``,
    ``edge_scalars_pre_linears.N.0.*``, ``gnn_block_init.*``), ``forward(query_points, input_points_multiscale) -> FeaturedPoints``.

    ``irreps_output`` is either the input irreps (features out) or ``64x0e`` (the weight field: the FFN ends in 64 scalars and ``skip_2`` is a
    LinearRS with bias, ``gnn_block.py:112``); in the second case ``forward`` returns the 64 scalars and keeps ``emb`` for the caller."""

    def __init__(self, irreps_input, irreps_output, irreps_sh, num_heads: int, fc_neurons: List[int], length_emb_dim: int, irreps_query,
                 r_cluster_multiscale: List[Optional[float]], edge_context_emb_dim: Optional[int], r_mincut_nonscalar_sh: Optional[float] = None,
                 length_enc_max_r: Optional[float] = None, n_scales: Optional[int] = None, n_layers: int = 1, irreps_mlp_mid=3,
                 attn_type: str = 'mlp', alpha_drop: float = 0.1, proj_drop: float = 0.1, drop_path_rate: float = 0.0,
                 use_src_point_attn: bool = False, use_dst_point_attn: bool = False, cutoff_method: str = 'edge_attn', init_seed: int = 2):
        super().__init__()
        if irreps_query is not None or edge_context_emb_dim is not None:
            raise NotImplementedError("MultiscaleTensorField with query features / context encoding is the score head's field: use ScoreModelHead")
        if use_src_point_attn or use_dst_point_attn or attn_type != 'mlp':
            raise NotImplementedError
        self.irreps_input, self.irreps_output = parse_irreps(str(irreps_input)), parse_irreps(str(irreps_output))
        self.scalar_out = self.irreps_output != self.irreps_input
        if self.scalar_out and self.irreps_output != [(64, 0)]:
            raise NotImplementedError(f"irreps_output must be the input irreps or 64x0e, not {irreps_output}")
        tf = dict(irreps_input=str(irreps_input), irreps_output=str(irreps_input), irreps_sh=str(irreps_sh), num_heads=num_heads,
                  fc_neurons=list(fc_neurons), length_emb_dim=length_emb_dim, r_cluster_multiscale=list(r_cluster_multiscale),
                  r_mincut_nonscalar_sh=r_mincut_nonscalar_sh, length_enc_max_r=length_enc_max_r, n_scales=n_scales, n_layers=n_layers,
                  irreps_mlp_mid=irreps_mlp_mid, cutoff_method=cutoff_method)
        self.cfg = HeadConfig.from_kwargs(dict(ebm=True, max_time=1.0, time_emb_mlp=[256, 128, 64], key_tensor_field_kwargs=tf,
                                               irreps_query_edf=str(irreps_input), lin_mult=1.0, ang_mult=1.0, edge_time_encoding=False,
                                               query_time_encoding=False))
        if self.cfg.lmax not in (2, 3):
            raise NotImplementedError("context-free fields are instantiated for lmax 2 and 3")
        self.n_scales, self.dim = self.cfg.n_scales, self.cfg.dim
        full = init_params(self.cfg, seed=init_seed)
        self._unused = {k: torch.zeros_like(v) for k, v in full.items() if not k.startswith(_KTF)}      # the head's time MLPs: not part of a field
        g = torch.Generator().manual_seed(init_seed + 1)
        for k, v in full.items():
            if not k.startswith(_KTF):
                continue
            name = k[len(_KTF):]
            if self.scalar_out and name == "gnn_block_init.ffn.fctp_2.tp.weight":
                v = v[:3 * 64 * 64].clone()                                            # 192x0e -> 64x0e is the only path left
            _register(self, name, v)
        if self.scalar_out:
            _register(self, "gnn_block_init.skip_2.skip.tp.weight", torch.randn(64 * 64, generator=g) / 8.0)
            _register(self, "gnn_block_init.skip_2.skip.bias.0", torch.zeros(64))
        self._handle, self._handle_device, self._keys = None, None, None
        self.register_load_state_dict_post_hook(MultiscaleTensorField._after_load)

    # ------------------------------------------------------------------------------------------------------------------
    def _c_state(self) -> Dict[str, torch.Tensor]:
        sd = {k: v.detach().float().cpu() for k, v in self.state_dict().items()}
        out = dict(self._unused)
        for k, v in sd.items():
            if k.startswith("gnn_block_init.skip_2."):
                continue
            if self.scalar_out and k == "gnn_block_init.ffn.fctp_2.tp.weight":
                muls = self.cfg.muls
                v = torch.cat([v.reshape(-1), torch.zeros(sum(3 * m * m for m in muls[1:]))])   # the rows of the other degrees: zero
            out[_KTF + k] = v
        return out

    def _release(self):
        if self._handle is not None:
            _lib.load().dedf_destroy(self._handle)
            self._handle = None
        self._keys = None

    def _after_load(self, *_):          # also when a parent module loads a checkpoint
        self._release()

    def half(self):
        """half-precision GEMM mode of the field's two fused kernels (``dedf_config.half_gemm``); parameters stay fp32 master copies"""
        self.cfg.half_gemm = True
        self._release()
        return self

    def float(self):
        self.cfg.half_gemm = False
        self._release()
        return self

    def __del__(self):
        try:
            self._release()
        except Exception:
            pass

    def _ensure_handle(self, device: torch.device):
        if self._handle is not None and self._handle_device == device:
            return
        self._release()
        if device.type != 'cuda':
            raise RuntimeError("diffusion_edf_amd.MultiscaleTensorField runs on an MI355X (torch device 'cuda') only; there is no CPU path")
        lib = _lib.load()
        idx = device.index if device.index is not None else torch.cuda.current_device()
        ccfg = _lib.make_config(self.cfg, idx, 0x7fffffff - 64)      # edge workspace: the exact worst case (few points, wide radii)
        blob = _lib.pack_params(ccfg, self._c_state())
        h = C.c_void_p()
        rc = lib.dedf_create(C.byref(ccfg), blob.ctypes.data_as(C.POINTER(C.c_float)), blob.size, C.byref(h))
        if rc != _lib.OK:
            _lib.raise_for(lib, None, rc, "dedf_create failed (see stderr)")
        self._handle, self._handle_device = h, device

    @torch.no_grad()
    def field_and_emb(self, query_points: FeaturedPoints, input_points_multiscale: Sequence[FeaturedPoints], max_neighbors: int = 1000):
        """the block's output and its ``emb`` (kernel-side ``dedf_field``), without the ``64x0e`` post-processing of ``forward``: what
        ``KeypointExtractor`` hands to ``dedf_keypoint_weight`` (which forms ``FFN + skip_2(emb)`` itself)"""
        assert len(input_points_multiscale) == self.n_scales
        assert query_points.x.ndim == 2 and query_points.x.shape[-1] == 3
        if max_neighbors != self.cfg.max_neighbors:
            raise NotImplementedError("max_neighbors is fixed at 1000 (every call site of the reference)")
        dev = query_points.x.device
        self._ensure_handle(dev)
        lib = _lib.load()
        stream = C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)
        with torch.cuda.device(dev):
            xs = [p.x.detach().float().contiguous() for p in input_points_multiscale]
            fs = [p.f.detach().float().contiguous() for p in input_points_multiscale]
            for x, f in zip(xs, fs):
                assert x.ndim == 2 and x.shape[-1] == 3 and f.shape == (len(x), self.dim), f"{x.shape} {f.shape}"
            n = len(xs)
            rc = lib.dedf_set_key_clouds(self._handle, n, (C.c_int * n)(*[len(x) for x in xs]), (C.c_void_p * n)(*[x.data_ptr() for x in xs]),
                                         (C.c_void_p * n)(*[f.data_ptr() for f in fs]), stream)
            _lib.raise_for(lib, self._handle, rc, "dedf_set_key_clouds")
            xq = query_points.x.detach().float().contiguous()
            field = torch.empty(len(xq), self.dim, device=dev, dtype=torch.float32)
            emb = torch.empty_like(field)
            rc = lib.dedf_field(self._handle, len(xq), xq.data_ptr(), field.data_ptr(), emb.data_ptr(), stream)
            _lib.raise_for(lib, self._handle, rc, "dedf_field")
        return field, emb

    @torch.no_grad()
    def forward(self, query_points: FeaturedPoints, input_points_multiscale: Sequence[FeaturedPoints], context_emb=None,
                max_neighbors: int = 1000) -> FeaturedPoints:
        assert context_emb is None
        field, emb = self.field_and_emb(query_points, input_points_multiscale, max_neighbors)
        dev = query_points.x.device
        lib = _lib.load()
        stream = C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)
        with torch.cuda.device(dev):
            if self.scalar_out:                       # FFN output + skip_2(emb), skip_2 = LinearRS(emb -> 64x0e, bias) on the HIP per-node linear
                sd = self.state_dict()
                muls = self.cfg.muls
                from . import unet_pad
                kmuls = unet_pad.wide_of(muls)                 # the per-node linear works in the kernel layout (lmax 3: 8x3e padded to 16x3e)
                W = torch.cat([sd["gnn_block_init.skip_2.skip.tp.weight"].detach().float().reshape(-1).cpu(),
                               torch.zeros(sum(m * m for m in kmuls[1:]))]).to(dev).contiguous()
                b = sd["gnn_block_init.skip_2.skip.bias.0"].detach().float().to(dev).contiguous()
                emb_k = unet_pad.pad_features(emb, muls).contiguous()
                skip = torch.empty_like(emb_k)
                rc = lib.dedf_linear_rs_lmax(len(muls) - 1, emb_k.data_ptr(), len(emb_k), None, None, W.data_ptr(), b.data_ptr(), None, skip.data_ptr(), stream)
                if rc != _lib.OK:
                    raise RuntimeError(f"dedf_linear_rs failed ({rc})")
                f_out = field[:, :64] - emb[:, :64] + skip[:, :64]
            else:
                f_out = field
        return FeaturedPoints(x=query_points.x, f=f_out.to(query_points.x.dtype), b=query_points.b, w=query_points.w)


class StaticKeypointModel(torch.nn.Module):
    """reference ``keypoint_extractor.py:22-47``: learned key points, features and weight logits; ``forward`` repeats them per batch index."""

    def __init__(self, keypoint_coords: Union[torch.Tensor, List], irreps_output):
        super().__init__()
        keypoint_coords = torch.as_tensor(keypoint_coords, dtype=torch.float32)
        assert keypoint_coords.ndim == 2 and keypoint_coords.shape[-1] == 3, f"{keypoint_coords.shape}"
        self.irreps_output = parse_irreps(str(irreps_output))
        self.register_buffer("keypoint_coords", keypoint_coords)
        g = torch.Generator().manual_seed(0)
        self.keypoint_features = torch.nn.Parameter(torch.randn(len(keypoint_coords), irreps_dim(self.irreps_output), generator=g))
        self.keypoint_weights = torch.nn.Parameter(torch.randn(len(keypoint_coords), generator=g))

    @torch.no_grad()
    def forward(self, input_points: FeaturedPoints) -> FeaturedPoints:
        b = input_points.b
        assert b.ndim == 1
        batch_unique = torch.unique(b)
        nb = len(batch_unique)
        return FeaturedPoints(x=self.keypoint_coords.repeat(nb, 1), f=self.keypoint_features.repeat(nb, 1),
                              b=batch_unique.repeat(len(self.keypoint_coords)), w=torch.sigmoid(self.keypoint_weights).repeat(nb))


class KeypointExtractor(torch.nn.Module):
    """reference ``keypoint_extractor.py:50-197``; constructor keywords = the ``query_kwargs`` block of the place tasks' YAML files."""

    def __init__(self, feature_extractor_kwargs: Dict, tensor_field_kwargs: Dict, keypoint_kwargs: Dict,
                 feature_extractor_name: str = 'UnetFeatureExtractor', weight_activation: str = 'sigmoid',
                 weight_mult: Optional[Union[float, int]] = None, deterministic: bool = False):
        super().__init__()
        from .unet import ForwardOnlyFeatureExtractor, UnetFeatureExtractor
        self.deterministic = deterministic
        self.pool_ratio = float(keypoint_kwargs['pool_ratio'])
        self.keypoint_bbox = keypoint_kwargs.get('bbox', None)
        wdim = keypoint_kwargs['weight_pre_emb_dim']
        if not wdim:                                                                   # :67-76: default = the number of scalars of the features
            wdim = sum(m for m, l in parse_irreps(str(feature_extractor_kwargs['irreps_output'])) if l == 0)
        self.weight_pre_emb_dim = int(wdim)
        if self.weight_pre_emb_dim != 64:
            raise NotImplementedError("weight_pre_emb_dim must be 64 (every shipped config)")
        if weight_activation not in ('sigmoid', 'none'):
            # the reference's test `weight_activation == 'sigmoid' or 'none'` (:121) is always true, so 'softmax' silently means "no extra
            # activation after an Identity": only the two meaningful values are accepted here
            raise NotImplementedError(f"weight_activation {weight_activation!r}")
        self.weight_sigmoid = weight_activation == 'sigmoid'
        if weight_mult is None:
            self.weight_mult_logit = None
        else:
            self.weight_mult_logit = torch.nn.Parameter(torch.log(torch.exp(torch.tensor(float(weight_mult))) - 1), requires_grad=False)
        if feature_extractor_name == 'UnetFeatureExtractor':
            self.feature_extractor = UnetFeatureExtractor(**feature_extractor_kwargs, deterministic=deterministic)
        elif feature_extractor_name == 'ForwardOnlyFeatureExtractor':
            self.feature_extractor = ForwardOnlyFeatureExtractor(**feature_extractor_kwargs, deterministic=deterministic)
        else:
            raise ValueError(f"Unknown feature extractor name: {feature_extractor_name}")
        tf = dict(tensor_field_kwargs)
        for k in ('irreps_input', 'irreps_query', 'edge_context_emb_dim'):
            assert k not in tf                                                         # :93-101
        tf.update(irreps_input=feature_extractor_kwargs['irreps_output'], irreps_query=None, edge_context_emb_dim=None)
        self.tensor_field = MultiscaleTensorField(**tf, init_seed=11)
        self.weight_field = MultiscaleTensorField(**dict(tf, irreps_output=f"{self.weight_pre_emb_dim}x0e"), init_seed=12)
        self.weight_post = torch.nn.Sequential(torch.nn.LayerNorm(self.weight_pre_emb_dim), torch.nn.SiLU(inplace=True),
                                               torch.nn.Linear(self.weight_pre_emb_dim, 1),
                                               torch.nn.Sigmoid() if self.weight_sigmoid else torch.nn.Identity())
        for p in self.weight_post.parameters():
            p.requires_grad_(False)
        self.irreps_output = str(tensor_field_kwargs['irreps_output'])

    def half(self):
        """``model.half()`` (reference agent.py:50-51): the UNet's layers and the two fields switch to single-product fp16 GEMMs (fp32 accumulate,
        everything else fp32); the per-node projections and the 64-channel weight head stay in full precision"""
        self.feature_extractor.half()
        self.tensor_field.half()
        self.weight_field.half()
        return self

    def float(self):
        self.feature_extractor.float()
        self.tensor_field.float()
        self.weight_field.float()
        return self

    @torch.no_grad()
    def init_query_points(self, src_points: FeaturedPoints, retain_feature: bool = False, retain_weight: bool = False) -> FeaturedPoints:
        """``:128-169``: points inside the bbox, then FPS with ``pool_ratio`` (start at the first point when deterministic)"""
        from .connectivity import fps
        assert src_points.x.ndim == 2 and src_points.x.shape[-1] == 3, f"{src_points.x.shape}"
        x, f, b, w = src_points.x, src_points.f, src_points.b, src_points.w
        if self.keypoint_bbox is not None:
            bbox = torch.tensor(self.keypoint_bbox, dtype=x.dtype, device=x.device)
            idx = ((x >= bbox[:, 0]) & (x <= bbox[:, 1])).all(dim=-1).nonzero().reshape(-1)
            x, f, b = x.index_select(0, idx), f.index_select(0, idx), b.index_select(0, idx)
            if w is not None:
                w = w.index_select(0, idx)
        picked = fps(x.detach(), b.detach(), ratio=self.pool_ratio, random_start=not self.deterministic)
        x, b = x.index_select(0, picked), b.index_select(0, picked)
        f = f.index_select(0, picked) if retain_feature else torch.empty_like(x)
        w = w.index_select(0, picked) if (retain_weight and w is not None) else None
        return FeaturedPoints(x=x, f=f, b=b, w=w)

    def get_query_points(self, src_points: FeaturedPoints) -> FeaturedPoints:
        return self.init_query_points(src_points, retain_feature=False, retain_weight=False)

    @torch.no_grad()
    def forward(self, input_points: FeaturedPoints, max_neighbors: Optional[int] = 1000) -> FeaturedPoints:
        multiscale = self.feature_extractor(input_points)
        query_points = self.get_query_points(input_points)
        out = self.tensor_field(query_points=query_points, input_points_multiscale=multiscale, context_emb=None, max_neighbors=max_neighbors)
        wf = self.weight_field
        w_field, w_emb = wf.field_and_emb(query_points, multiscale, max_neighbors)      # (not wf.forward: the weight head below forms FFN + skip_2(emb) itself)
        dev = query_points.x.device
        sd = {k: v.detach().float().to(dev).contiguous() for k, v in wf.state_dict().items() if k.startswith("gnn_block_init.skip_2.")}
        ln, lin = self.weight_post[0], self.weight_post[2]
        t = lambda v: v.detach().float().to(dev).contiguous()
        ln_w, ln_b, lin_w = t(ln.weight), t(ln.bias), t(lin.weight.reshape(-1))
        mult = 1.0 if self.weight_mult_logit is None else float(torch.nn.functional.softplus(self.weight_mult_logit))
        n = len(query_points.x)
        weights = torch.empty(n, device=dev, dtype=torch.float32)
        lib = _lib.load()
        with torch.cuda.device(dev):
            rc = lib.dedf_keypoint_weight(w_field.data_ptr(), w_emb.data_ptr(), n, wf.dim,
                                          sd["gnn_block_init.skip_2.skip.tp.weight"].data_ptr(), sd["gnn_block_init.skip_2.skip.bias.0"].data_ptr(),
                                          ln_w.data_ptr(), ln_b.data_ptr(), lin_w.data_ptr(), float(lin.bias), int(self.weight_sigmoid), mult,
                                          weights.data_ptr(), C.c_void_p(torch.cuda.current_stream(dev).cuda_stream))
        if rc != _lib.OK:
            raise RuntimeError(f"dedf_keypoint_weight failed ({rc})")
        return FeaturedPoints(x=out.x, f=out.f, b=out.b, w=weights.to(out.f.dtype))
