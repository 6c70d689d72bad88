"""First slice of the UNet feature extractor: ONE layer on the MI355X-native kernels.

The reference builds every layer of ``UnetFeatureExtractor`` — pool layers, radius-graph layers, mid block, unpool layers —
as the same pair (``unet_feature_extractor.py:141-156, 160-176, 187-202``)::

    layer['radial'] = GaussianRadialBasisLayerFiniteCutoff(num_basis=fc_neurons[0], cutoff=0.99 * radius)
    layer['gnn']    = block.EquiformerBlock(irreps_src, irreps_dst, irreps_edge_attr, irreps_head, num_heads, fc_neurons, ...)

and applies it to a bipartite graph ``edge_src -> edge_dst`` with ``edge_vec = x_src[edge_src] - x_dst[edge_dst]`` (``:289-302``,
``:316-324``).  ``UnetLayer`` is that pair behind one call: same parameter names as the reference's state dict below the layer
(``radial.mean``, ``gnn.linear_src.tp.weight``, ``gnn.ga.sep_act.dtp_rad.net.0.weight`` …), the arithmetic in ``libdedf.so``
(``dedf_layer_forward``: the fused edge kernel in its UNet mode — destination message added, radial basis read by the radial MLP
directly, no edge logits — the joint-softmax aggregation, and the node kernel with the block's two skip connections).

The kernels are instantiated for ``64x0e+32x1e+16x2e`` with ``fc_neurons [64, 32, 32]`` (levels 2 and 3 and the mid block of every UNet
the reference ships, ``configs/*/*/score_model_configs.yaml: irreps_emb[2:]``).  The two fine levels (``32x0e+16x1e+8x2e``,
``[32, 16, 16]``) and the layers between levels of different width run on the SAME kernels as zero-padded wide layers — an exact
embedding, see ``unet_pad.py`` — at the price of the padded arithmetic.  Graphs come from ``connectivity.FpsPool`` / ``RadiusGraph``
(HIP ``dedf_fps`` / ``dedf_radius``).  GPU only.
"""
from __future__ import annotations

import ctypes as C
import weakref
from typing import Dict, List, Optional, Sequence

import torch

from . import _lib, unet_pad
from .params import init_from_spec, unet_layer_param_spec
from .score_head import _register
from .so3 import irreps_dim, parse_irreps


_SHAPES = {(64, 32, 16): "wide", (32, 16, 8): "narrow", (64, 32, 16, 8): "wide, lmax 3", (32, 16, 8, 4): "narrow, lmax 3"}


class UnetLayer(torch.nn.Module):
    """{'radial', 'gnn'} of one UNet layer.  ``forward`` returns the new destination features ``(n_dst, D_dst)``.

    ``irreps`` is the block's ``irreps_dst`` (= its ``irreps_emb``), ``irreps_src`` the irreps of the source features (pool / unpool layers
    between levels of different width).  Both ``64x0e+32x1e+16x2e`` and ``32x0e+16x1e+8x2e`` are accepted, with ``fc_neurons``
    ``[64, 32, 32]`` or ``[32, 16, 16]``: the narrow shapes run zero-padded on the wide kernels (``unet_pad.py``; exact, not approximate)."""

    def __init__(self, irreps: str = '64x0e+32x1e+16x2e', irreps_edge_attr: str = '1x0e+1x1e+1x2e', num_heads: int = 4,
                 fc_neurons: Sequence[int] = (64, 32, 32), radius: float = 15.0, irreps_mlp_mid: int = 3, init_seed: int = 2,
                 irreps_src: Optional[str] = None):
        super().__init__()
        self.irreps = parse_irreps(irreps)
        self.irreps_src = self.irreps if irreps_src is None else parse_irreps(irreps_src)
        sh = parse_irreps(irreps_edge_attr)
        L = len(self.irreps) - 1
        for irr in (self.irreps, self.irreps_src):
            if tuple(m for m, _ in irr) not in _SHAPES or [l for _, l in irr] != list(range(L + 1)):
                raise NotImplementedError(f"UnetLayer: irreps must be 64x0e+32x1e+16x2e(+8x3e) or 32x0e+16x1e+8x2e(+4x3e), source and destination of one lmax, not {irr}")
        if list(fc_neurons) not in ([64, 32, 32], [32, 16, 16]):
            raise NotImplementedError(f"UnetLayer: fc_neurons must be [64, 32, 32] or [32, 16, 16], not {list(fc_neurons)}")
        if [l for _, l in sh] != list(range(L + 1)) or any(m != 1 for m, _ in sh) or num_heads != 4 or irreps_mlp_mid != 3:
            raise NotImplementedError("UnetLayer: irreps_edge_attr 1x0e+1x1e+...+1x{lmax}e with lmax that of the features, 4 heads, irreps_mlp_mid 3")
        self.lmax, self.kdim = L, unet_pad.wide_dim(L)
        self.fc_neurons, self.num_heads, self.radius = list(fc_neurons), num_heads, float(radius)
        self.muls, self.muls_src = [m for m, _ in self.irreps], [m for m, _ in self.irreps_src]
        self.dim, self.dim_src = irreps_dim(self.irreps), irreps_dim(self.irreps_src)
        spec = unet_layer_param_spec(self.irreps, self.fc_neurons, num_heads, lmax_sh=L, irreps_src=self.irreps_src)
        for name, t in init_from_spec(spec, seed=init_seed).items():
            _register(self, name, t)
        self._handle = None
        self._handle_device: Optional[torch.device] = None
        self._deferred = False
        self._half = False
        self._ws_box = [None]                      # share_workspace: [the layer whose per-call workspace this one runs in] (boxed: not a submodule) ...
        self._ws_link = None                       # ... and the owner's handle the link was made with
        self._ws_borrowers = weakref.WeakSet()     # layers that run in THIS layer's workspace
        self.register_load_state_dict_post_hook(UnetLayer._after_load)

    # ------------------------------------------------------------------------------------------------------------------
    def _release(self):
        if self._handle is not None:
            lib = _lib.load()
            for b in list(self._ws_borrowers):     # nobody may keep a pointer to a workspace that is about to go
                if b._handle is not None and b._ws_link is not None:
                    lib.dedf_layer_share_workspace(b._handle, None)
                b._ws_link = None
            lib.dedf_destroy(self._handle)
            self._handle = None
        self._ws_link = None

    def share_workspace(self, owner: Optional["UnetLayer"]):
        """run in ``owner``'s per-call workspace (``dedf_layer_share_workspace``: messages, edge lists, segment records, aggregate, verdict word)
        instead of an own one — the layers of an extractor run one after the other on one stream.  ``None`` detaches.  The link is (re)made
        lazily, whenever either handle is (re)built."""
        old = self._ws_box[0]
        if old is not None and old is not owner:
            old._ws_borrowers.discard(self)
            if self._handle is not None and self._ws_link is not None:
                _lib.load().dedf_layer_share_workspace(self._handle, None)
            self._ws_link = None
        self._ws_box[0] = owner if owner is not self else None
        if self._ws_box[0] is not None:
            self._ws_box[0]._ws_borrowers.add(self)

    def _link_workspace(self, device):
        o = self._ws_box[0]
        if o is None:
            return
        o._ensure_handle(device)
        if self._ws_link != o._handle.value:
            lib = _lib.load()
            rc = lib.dedf_layer_share_workspace(self._handle, o._handle)
            _lib.raise_for(lib, self._handle, rc, "dedf_layer_share_workspace")
            self._ws_link = o._handle.value

    # (the packed device image is rebuilt on next use after ANY load: a post hook also fires when a parent module loads a checkpoint,
    #  which never calls the children's load_state_dict)
    def _after_load(self, *_):
        self._release()

    def half(self):
        """the reference's ``model.half()`` (agent.py:50-51) for this layer: every GEMM of its two fused kernels as ONE fp16 MFMA product (fp16
        operands, fp32 accumulation) instead of the 3-term split; parameters stay fp32 master copies, everything else stays fp32"""
        self._half = True
        self._release()
        return self

    def float(self):
        self._half = False
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
            raise RuntimeError("diffusion_edf_amd.UnetLayer runs on an MI355X (torch device 'cuda') only; there is no CPU path "
                               "(the CPU restatement lives under oracle/ and is test infrastructure).")
        lib = _lib.load()
        idx = device.index if device.index is not None else torch.cuda.current_device()
        W = unet_pad.wide_of(self.muls)          # (at lmax 3 every shape is "narrow": 8x3e itself runs padded to 16x3e)
        narrow = self.muls != W or self.muls_src != W or self.fc_neurons != unet_pad.WIDE_FC
        state = {k: v for k, v in self.state_dict().items()}
        if narrow:
            state = unet_pad.expand_layer_params(state, self.muls, self.fc_neurons, self.muls_src)
        # both node sets in the narrow level shape (levels 0-1 of the panda UNets): the narrow instantiations of the layer kernels, which skip the
        # lane-local work on the structurally zero channels (`unet_pad.place` puts the true ones where the kernels expect them)
        nw = list(self.muls) == list(self.muls_src) == unet_pad.NARROW3[:len(self.muls)]
        ccfg = _lib.make_unet_layer_config(self.radius, idx, unet_pad.WIDE_FC, W, self.num_heads,
                                           valid=self.muls if self.muls != W else None,
                                           fc_valid=self.fc_neurons if self.fc_neurons != unet_pad.WIDE_FC else None, half_gemm=self._half, narrow=nw)
        blob = _lib.pack_params(ccfg, state)
        h = C.c_void_p()
        rc = lib.dedf_create(C.byref(ccfg), blob.ctypes.data_as(C.POINTER(C.c_float)), blob.size, C.byref(h))
        if rc != _lib.OK:
            _lib.raise_for(lib, None, rc, "dedf_create failed (see stderr)")
        self._handle, self._handle_device, self._deferred = h, device, False

    @torch.no_grad()
    def forward(self, node_coord_src: torch.Tensor, node_input_src: torch.Tensor, node_coord_dst: torch.Tensor,
                node_input_dst: torch.Tensor, edge_src: torch.Tensor, edge_dst: torch.Tensor) -> torch.Tensor:
        """Arguments as the reference has them at the call site (``unet_feature_extractor.py:289-302``): coordinates and features of
        both node sets and the edge lists (int64).  Edges sorted by ``edge_dst`` (as ``FpsPool`` / ``RadiusGraph`` return them) are used
        as they are; any other order (the reversed graphs of the up path) is sorted here — the sums do not depend on it."""
        assert node_input_src.shape == (len(node_coord_src), self.dim_src), f"{node_input_src.shape}"
        assert node_input_dst.shape == (len(node_coord_dst), self.dim), f"{node_input_dst.shape}"
        dev = node_coord_src.device
        f32 = lambda t: t.detach().to(device=dev, dtype=torch.float32)
        fs = unet_pad.pad_features(f32(node_input_src), self.muls_src)
        fd = unet_pad.pad_features(f32(node_input_dst), self.muls)
        ed = edge_dst.detach().to(device=dev, dtype=torch.int64)
        dst_sorted = not (len(ed) > 1 and bool((ed[1:] < ed[:-1]).any()))
        out = self.forward_wide(node_coord_src, fs, node_coord_dst, fd, edge_src, ed, dst_sorted=dst_sorted)
        return unet_pad.unpad_features(out, self.muls).to(node_input_dst.dtype)

    @torch.no_grad()
    def forward_wide(self, node_coord_src, f_src_wide, node_coord_dst, f_dst_wide, edge_src, edge_dst, dst_sorted: bool, deferred: bool = False):
        """the same on features already in the 240-wide kernel layout (``unet_pad.pad_features``) -> (n_dst, 240) in that layout: what the
        extractors chain, padding once at the input and un-padding once at the outputs.  ``dst_sorted=False`` sorts the edges by destination
        (no device read-back); ``deferred``: do not synchronise, the edge-list verdict is collected later (``check``)."""
        assert node_coord_src.ndim == 2 and node_coord_src.shape[-1] == 3 and node_coord_dst.ndim == 2 and node_coord_dst.shape[-1] == 3
        assert f_src_wide.shape == (len(node_coord_src), self.kdim) and f_dst_wide.shape == (len(node_coord_dst), self.kdim)
        assert edge_src.ndim == 1 and edge_src.shape == edge_dst.shape
        dev = node_coord_src.device
        self._ensure_handle(dev)
        self._link_workspace(dev)
        lib = _lib.load()
        f32 = lambda t: t.detach().to(device=dev, dtype=torch.float32).contiguous()
        xs, xd, fs, fd = f32(node_coord_src), f32(node_coord_dst), f32(f_src_wide), f32(f_dst_wide)
        es = edge_src.detach().to(device=dev, dtype=torch.int64)
        ed = edge_dst.detach().to(device=dev, dtype=torch.int64)
        if not dst_sorted:
            order = torch.sort(ed, stable=True).indices
            es, ed = es[order], ed[order]
        es, ed = es.contiguous(), ed.contiguous()
        out = torch.empty(len(xd), self.kdim, device=dev, dtype=torch.float32)
        if deferred != self._deferred:
            lib.dedf_layer_defer_check(self._handle, int(deferred))
            self._deferred = deferred
        stream = C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)
        rc = lib.dedf_layer_forward(self._handle, len(xs), xs.data_ptr(), fs.data_ptr(), len(xd), xd.data_ptr(), fd.data_ptr(),
                                    len(es), es.data_ptr(), ed.data_ptr(), out.data_ptr(), stream)
        _lib.raise_for(lib, self._handle, rc, "dedf_layer_forward")
        return out

    def check(self):
        """collect the edge-list verdict of the ``deferred`` calls since the last check (synchronises)"""
        if self._handle is not None and self._deferred:
            lib = _lib.load()
            rc = lib.dedf_layer_check(self._handle, C.c_void_p(torch.cuda.current_stream(self._handle_device).cuda_stream))
            _lib.raise_for(lib, self._handle, rc, "dedf_layer_forward")


# =====================================================================================================================================
# The whole extractor
# =====================================================================================================================================

def _muls(irreps) -> List[int]:
    return [m for m, _ in irreps]


class NodeLinear(torch.nn.Module):
    """[EquivariantLayerNormV2 +] LinearRS per node on the HIP kernel ``dedf_linear_rs`` — the reference's ``LinearRS`` (``input_emb``,
    unet_feature_extractor.py:64-66) and ``ProjectIfMismatch`` (skip.py:13-34: LayerNorm over irreps_in, then LinearRS with bias; identity when
    the irreps agree).  Parameter names as in the reference: ``tp.weight`` / ``bias.0`` (LinearRS), ``layernorm.affine_*`` + ``skip.tp.weight``
    / ``skip.bias.0`` (ProjectIfMismatch).  Narrow irreps run zero-padded like the layers (``unet_pad.py``)."""

    def __init__(self, irreps_in, irreps_out, layernorm: bool, prefix: str = "", init_seed: int = 2):
        super().__init__()
        self.irreps_in, self.irreps_out = parse_irreps(irreps_in) if isinstance(irreps_in, str) else list(irreps_in), \
            parse_irreps(irreps_out) if isinstance(irreps_out, str) else list(irreps_out)
        self.has_ln, self.prefix = layernorm, prefix
        self.m_in = {l: m for m, l in self.irreps_in}
        self.m_out = _muls(self.irreps_out)
        self.L = L = len(self.m_out) - 1
        self.W = unet_pad.wide_of(self.m_out)
        self.kdim = unet_pad.wide_dim(L)
        assert [l for _, l in self.irreps_out] == list(range(L + 1)) and tuple(self.m_out) in _SHAPES
        assert all(l in range(L + 1) for l in self.m_in) and all(self.m_in.get(l, 0) <= self.W[l] for l in range(L + 1))
        g = torch.Generator().manual_seed(init_seed)
        blocks = [(self.m_in.get(l, 0), self.m_out[l]) for l in range(L + 1)]
        w = torch.cat([torch.randn(a * b, generator=g) / max(a, 1) ** 0.5 for a, b in blocks if a > 0])
        lin = prefix + ("skip." if layernorm else "")
        _register(self, lin + "tp.weight", w)
        _register(self, lin + "bias.0", torch.zeros(self.m_out[0]))
        if layernorm:
            _register(self, prefix + "layernorm.affine_weight", torch.ones(sum(self.m_in.values())))
            _register(self, prefix + "layernorm.affine_bias", torch.zeros(self.m_in.get(0, 0)))
        self._dev = None
        self._pl_cache = {}
        self.register_load_state_dict_post_hook(NodeLinear._after_load)

    def _pl_in(self, l):      # placement of the input channels of degree l inside the wide block
        m = self.m_in.get(l, 0)
        return unet_pad.place(m, self.W[l]) if (m and m % 4 == 0 and tuple(self.m_in.get(k, 0) for k in range(self.L + 1)) in _SHAPES) else torch.arange(m)

    def _pl_dev(self, l, dev):
        key = (l, str(dev))
        if key not in self._pl_cache:
            self._pl_cache[key] = self._pl_in(l).to(dev)
        return self._pl_cache[key]

    def _wide(self, dev):
        if self._dev is not None and self._dev[0] == dev:
            return self._dev[1:]
        sd = {k: v.detach().float().cpu() for k, v in self.state_dict().items()}
        lin = self.prefix + ("skip." if self.has_ln else "")
        M = self.W
        W, o = [], 0
        for l in range(self.L + 1):
            a, b = self.m_in.get(l, 0), self.m_out[l]
            full = torch.zeros(M[l], M[l])
            if a:
                full[self._pl_in(l)[:, None], unet_pad.place(b, M[l])[None, :]] = sd[lin + "tp.weight"][o:o + a * b].reshape(a, b)
                o += a * b
            W.append(full.reshape(-1))
        bias = torch.zeros(M[0]); bias[unet_pad.place(self.m_out[0], M[0])] = sd[lin + "bias.0"]
        lnw = lnb = None
        valid = None
        if self.has_ln:
            lnw, lnb, ow = torch.zeros(sum(M)), torch.zeros(M[0]), 0
            it = 0
            for l in range(self.L + 1):
                a = self.m_in.get(l, 0)
                lnw[ow + self._pl_in(l)] = sd[self.prefix + "layernorm.affine_weight"][it:it + a]
                it += a; ow += M[l]
            lnb[self._pl_in(0)] = sd[self.prefix + "layernorm.affine_bias"]
            valid = (C.c_int * 4)(*([self.m_in.get(l, 0) or M[l] for l in range(self.L + 1)] + [0] * (3 - self.L)))
        t = lambda v: None if v is None else v.to(dev).contiguous()
        self._dev = (dev, t(torch.cat(W)), t(bias), t(lnw), t(lnb), valid)
        return self._dev[1:]

    def _after_load(self, *_):
        self._dev = None

    def pad_in(self, f: torch.Tensor) -> torch.Tensor:
        out = f.new_zeros(len(f), self.kdim)
        o_t = o_w = 0
        for l in range(self.L + 1):
            a, d, Mw = self.m_in.get(l, 0), 2 * l + 1, self.W[l]
            if a:
                out[:, o_w:o_w + Mw * d].view(-1, Mw, d)[:, self._pl_dev(l, f.device), :] = f[:, o_t:o_t + a * d].reshape(-1, a, d)
                o_t += a * d
            o_w += Mw * d
        return out

    @torch.no_grad()
    def forward(self, f: torch.Tensor) -> torch.Tensor:
        if not f.is_cuda:
            raise RuntimeError("diffusion_edf_amd.unet needs GPU tensors: the product has no CPU path")
        return unet_pad.unpad_features(self.forward_wide(self.pad_in(f.detach().float())), self.m_out).to(f.dtype)

    @torch.no_grad()
    def forward_wide(self, x: torch.Tensor) -> torch.Tensor:
        """(n, 240) in the kernel layout -> (n, 240) in the kernel layout"""
        dev = x.device
        W, bias, lnw, lnb, valid = self._wide(dev)
        x = x.contiguous()
        out = torch.empty_like(x)
        lib = _lib.load()
        with torch.cuda.device(dev):
            rc = lib.dedf_linear_rs_lmax(self.L, x.data_ptr(), len(x), None if lnw is None else lnw.data_ptr(), None if lnb is None else lnb.data_ptr(),
                                         W.data_ptr(), bias.data_ptr(), valid, out.data_ptr(), C.c_void_p(torch.cuda.current_stream(dev).cuda_stream))
        if rc != _lib.OK:
            raise RuntimeError(f"dedf_linear_rs failed ({rc})")
        return out


class _ParitySign(torch.nn.Module):
    """holds the ``sign`` buffer of the reference's ``ParityInversionSh`` (utils.py:26-47) so that this key of a reference state dict finds a home (the reference's other
    persistent buffers -- e3nn ``tp.output_mask`` etc. -- are tolerated by ``agent.get_models``, see ``agent.is_reference_only_buffer``); the
    kernels never read it (the harmonics of a swapped edge are recomputed, Y_l(-v) = (-1)^l Y_l(v))"""

    def __init__(self, irreps):
        super().__init__()
        self.register_buffer('sign', torch.cat([torch.full(((2 * l + 1) * m,), -1.0 if l % 2 else 1.0) for m, l in parse_irreps(str(irreps))]))


class UnetFeatureExtractor(torch.nn.Module):
    """reference ``unet_feature_extractor.py:19-417`` on the MI355X-native kernels: same constructor keywords (the
    ``feature_extractor_kwargs`` block of the score-model YAML files), same state-dict names (``input_emb.*``, ``down_blocks.N.pool_proj.*``,
    ``down_blocks.N.pool_layer.{radial,gnn}.*``, ``down_blocks.N.layer_stack.I.*``, ``mid_block.I.*``, ``up_blocks.K.layer_stack.I.*``,
    ``up_blocks.K.unpool_layer.*``, ``project_outputs.N.*``), same ``forward(pcd) -> List[FeaturedPoints]``.

    Graphs: ``connectivity.FpsPool`` / ``RadiusGraph`` (HIP FPS + radius search); layers: ``UnetLayer``; projections: ``NodeLinear``; the up
    path re-uses the down path's graphs with source and destination swapped — ``ParityInversionSh`` (utils.py:26-47) is implicit, the edge
    kernel evaluates the harmonics of the swapped edge vector.  Skip connections ``(a + b) / sqrt(3)`` as at ``:347,359``.
    Supported: lmax 2, irreps_emb per level 64x0e+32x1e+16x2e or 32x0e+16x1e+8x2e with fc_neurons [64,32,32] / [32,16,16], 4 heads,
    pool_method 'fps', attn_type 'mlp' — every UNet the reference ships.  Dropout / drop-path are identity in eval and not modelled."""

    _forward_only = False

    def half(self):
        """half-precision GEMM mode of every layer (``UnetLayer.half``); the per-node projections and the graph primitives stay fp32"""
        for m in self.modules():
            if isinstance(m, UnetLayer):
                m.half()
        return self

    def float(self):
        for m in self.modules():
            if isinstance(m, UnetLayer):
                m.float()
        return self

    def __init__(self, irreps_input, irreps_output, irreps_emb, irreps_edge_attr, num_heads, fc_neurons, n_layers, pool_ratio, radius,
                 deterministic: bool = False, pool_method='fps', irreps_mlp_mid=3, attn_type='mlp', alpha_drop=0.1, proj_drop=0.1,
                 drop_path_rate=0.0, n_layers_midstream: int = 2, n_scales: Optional[int] = None, output_scalespace=None):
        super().__init__()
        import math
        from . import connectivity as CN
        self._ctor = dict(irreps_input=irreps_input, irreps_output=irreps_output, irreps_emb=list(irreps_emb), fc_neurons=[list(f) for f in fc_neurons],
                          n_layers=list(n_layers), pool_ratio=list(pool_ratio), n_layers_midstream=n_layers_midstream)
        self.irreps_output = str(irreps_output)
        self.irreps_emb = [str(i) for i in irreps_emb]
        self.n_scales = len(self.irreps_emb) if n_scales is None else n_scales
        ns = self.n_scales
        assert ns == len(self.irreps_emb) == len(irreps_edge_attr) == len(num_heads) == len(fc_neurons) == len(radius) == len(pool_ratio) == len(n_layers)
        if irreps_input is None:
            raise NotImplementedError("irreps_input=None")
        if any(str(a).replace(' ', '') not in ('1x0e+1x1e+1x2e', '1x0e+1x1e+1x2e+1x3e') for a in irreps_edge_attr) or any(h != 4 for h in num_heads):
            raise NotImplementedError("UnetFeatureExtractor: irreps_edge_attr 1x0e+1x1e+1x2e(+1x3e) and 4 heads per level")
        self._edge_attr = [str(a).replace(' ', '') for a in irreps_edge_attr]
        if (pool_method if isinstance(pool_method, str) else pool_method[0]) != 'fps' or (attn_type if isinstance(attn_type, str) else attn_type[0]) != 'mlp':
            raise NotImplementedError
        if irreps_mlp_mid != 3:
            raise NotImplementedError
        self.output_scalespace = [ns + n if n < 0 else n for n in (list(range(ns)) if output_scalespace is None else output_scalespace)]
        self.deterministic, self.n_layers, self.n_layers_midstream = deterministic, list(n_layers), n_layers_midstream
        self.radius = [radius[0]]
        for n, r in enumerate(radius[1:]):                                    # :78-86 (note the reference's pool_ratio[n-1])
            self.radius.append(self.radius[-1] / math.sqrt(pool_ratio[n - 1]) if r is None else r)
        emb, fc = self.irreps_emb, [list(f) for f in fc_neurons]
        self.input_emb = NodeLinear(irreps_input, emb[0], layernorm=False, init_seed=7)
        seeds = iter(range(100, 10 ** 6))
        mk = lambda n, src, dst: UnetLayer(irreps=dst, irreps_src=src, irreps_edge_attr=self._edge_attr[n], fc_neurons=fc[n], radius=self.radius[n],
                                           init_seed=next(seeds))
        self.down_blocks = torch.nn.ModuleList()
        for n in range(ns):
            blk = torch.nn.ModuleDict()
            blk['pool'] = CN.FpsPool(ratio=pool_ratio[n], random_start=not deterministic, r=self.radius[n], max_num_neighbors=1000)
            prev = emb[max(n - 1, 0)]
            blk['pool_proj'] = NodeLinear(prev, emb[n], layernorm=True) if parse_irreps(prev) != parse_irreps(emb[n]) else torch.nn.Identity()
            blk['radius_graph'] = CN.RadiusGraph(r=self.radius[n], max_num_neighbors=1000)
            blk['pool_layer'] = mk(n, prev, emb[n])
            blk['layer_stack'] = torch.nn.ModuleList([mk(n, emb[n], emb[n]) for _ in range(n_layers[n] - 1)])
            self.down_blocks.append(blk)
        self.mid_block = torch.nn.ModuleList([mk(ns - 1, emb[-1], emb[-1]) for _ in range(0 if self._forward_only else n_layers_midstream)])
        self.up_blocks = torch.nn.ModuleList()
        for n in range(-1 if self._forward_only else ns - 1, -1, -1):
            blk = torch.nn.ModuleDict()
            blk['layer_stack'] = torch.nn.ModuleList([mk(n, emb[n], emb[n]) for _ in range(n_layers[n] - 1)])
            blk['unpool_layer'] = mk(n, emb[n], emb[max(n - 1, 0)])
            blk['parity_inversion'] = _ParitySign(irreps_edge_attr[n])
            self.up_blocks.append(blk)
        self.project_outputs = torch.nn.ModuleList(
            [NodeLinear(emb[n], irreps_output, layernorm=True) if parse_irreps(emb[n]) != parse_irreps(irreps_output) else torch.nn.Identity()
             for n in range(ns)])
        # the layers run one after the other: ONE per-call workspace (and one verdict word) for all of them, the first layer's
        layers = [m for m in self.modules() if isinstance(m, UnetLayer)]
        for m in layers[1:]:
            m.share_workspace(layers[0])

    @torch.no_grad()
    def forward(self, pcd):
        """Inside, features travel in the 240-wide kernel layout (padded once after ``input_emb``, un-padded once at the outputs), the layers
        run without host synchronisation (their edge-list verdicts are collected at the end) and the graphs' batch check is done once."""
        import math
        from . import connectivity as CN
        from .gnn_data import FeaturedPoints
        x, f, b = pcd.x, pcd.f, pcd.b
        assert f.ndim == 2 and x.ndim == 2 and b.ndim == 1 and len(f) == len(x) == len(b)
        if not f.is_cuda:
            raise RuntimeError("diffusion_edf_amd.unet needs GPU tensors: the product has no CPU path")
        single = CN._check_cloud(x, b) is None       # several clouds in one batch vector: the graphs are built cloud by cloud (connectivity.py)
        dt = f.dtype
        x = x.detach().float().contiguous()
        run = lambda layer, xs, fs, xd, fd, es, ed, srt: layer.forward_wide(xs, fs, xd, fd, es, ed, dst_sorted=srt, deferred=True)
        lin = lambda mod, v: v if isinstance(mod, torch.nn.Identity) else mod.forward_wide(v)
        f = self.input_emb.forward_wide(self.input_emb.pad_in(f.detach().float()))
        down_out, down_edges, scale_out, used = [(f, x, b)], [], [], []
        for n_blk, blk in enumerate(self.down_blocks):
            # (deterministic cascade: from the second level on the cloud is the previous level's FPS order, whose re-sampling is its own prefix)
            f_dst, x_dst, es, ed, _, b_dst = blk['pool'](x, f, b, _trusted=single, _fps_ordered=single and self.deterministic and n_blk > 0, _need_degree=False)      # :279-282
            f_dst = lin(blk['pool_proj'], f_dst)
            f = run(blk['pool_layer'], x, f, x_dst, f_dst, es, ed, True)
            used.append(blk['pool_layer'])
            x, b = x_dst, b_dst
            down_out.append((f, x, b)); down_edges.append((es, ed))
            _, _, es, ed, _, _ = blk['radius_graph'](x, f, b, _trusted=single, _need_degree=False)                          # :306-311
            for layer in blk['layer_stack']:
                f = run(layer, x, f, x, f, es, ed, True)
                used.append(layer)
                down_out.append((f, x, b)); down_edges.append((es, ed))
            scale_out.append((f, x, b))
        if not self._forward_only:
            for layer in self.mid_block:                                                              # :332-344 (the last radius graph)
                f = run(layer, x, f, x, f, es, ed, True)
                used.append(layer)
            f_skip, _, _ = down_out.pop()
            f = (f + f_skip) / math.sqrt(3)                                                           # :347
            up_out = []
            for k, blk in enumerate(self.up_blocks):
                for layer in blk['layer_stack']:
                    f_skip, x_dst, b_dst = down_out.pop()
                    es, ed = down_edges.pop()
                    f_dst = (f + f_skip) / math.sqrt(3)                                               # :359
                    f = run(layer, x, f, x_dst, f_dst, ed, es, False)                                 # source / destination swapped (:358)
                    used.append(layer)
                    x, b = x_dst, b_dst
                up_out.append((f, x, b))
                f_dst, x_dst, b_dst = down_out.pop()                                                  # :381-403
                es, ed = down_edges.pop()
                if k != self.n_scales - 1:
                    f = run(blk['unpool_layer'], x, f, x_dst, f_dst, ed, es, False)
                    used.append(blk['unpool_layer'])
                    x, b = x_dst, b_dst
            scale_out = up_out[::-1]                                                                  # else: forward_only_feature_extractor.py:258-274
        outs = []
        for s in range(self.n_scales):
            if s in self.output_scalespace:
                fo = unet_pad.unpad_features(lin(self.project_outputs[s], scale_out[s][0]), _muls(parse_irreps(self.irreps_output)))
                outs.append(FeaturedPoints(x=scale_out[s][1].to(pcd.x.dtype), f=fo.to(dt), b=scale_out[s][2], w=None))
        if used:
            used[0].check()          # the layers share one workspace and one deferred verdict word (UnetLayer.share_workspace): one read-back
        return outs


class ForwardOnlyFeatureExtractor(UnetFeatureExtractor):
    """reference ``forward_only_feature_extractor.py:19-275`` (the key model of the sapien high-res configs): the down path of the UNet only —
    per scale FpsPool -> pool layer -> radius-graph layer stack — and the features at the end of each scale, projected to ``irreps_output``.
    Same constructor keywords (``n_layers_midstream`` is accepted and unused, as in the reference); state-dict names ``input_emb.*``,
    ``down_blocks.N.*``, ``project_outputs.N.*``."""
    _forward_only = True
