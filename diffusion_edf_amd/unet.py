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

Instantiated for ``irreps_src = irreps_dst = 64x0e+32x1e+16x2e`` with ``fc_neurons [64, 32, 32]``: levels 2 and 3 and the mid block of
every UNet the reference ships (``configs/*/*/score_model_configs.yaml: irreps_emb[2:]``).  The two fine levels (``32x0e+16x1e+8x2e``,
``[32, 16, 16]``) need a kernel family with 8-channel blocks and are not built; ``irreps_src != irreps_dst`` (the level-2 pool layer)
likewise.  Graphs come from ``connectivity.FpsPool`` / ``RadiusGraph`` (HIP ``dedf_fps`` / ``dedf_radius``).  GPU only.
"""
from __future__ import annotations

import ctypes as C
from typing import Dict, List, Optional, Sequence

import torch

from . import _lib
from .params import init_from_spec, unet_layer_param_spec
from .score_head import _register
from .so3 import irreps_dim, parse_irreps


class UnetLayer(torch.nn.Module):
    """{'radial', 'gnn'} of one UNet layer.  ``forward`` returns the new destination features ``(n_dst, D)``."""

    def __init__(self, irreps: str = '64x0e+32x1e+16x2e', irreps_edge_attr: str = '1x0e+1x1e+1x2e', num_heads: int = 4,
                 fc_neurons: Sequence[int] = (64, 32, 32), radius: float = 15.0, irreps_mlp_mid: int = 3, init_seed: int = 2):
        super().__init__()
        self.irreps = parse_irreps(irreps)
        sh = parse_irreps(irreps_edge_attr)
        if [m for m, _ in self.irreps] != [64, 32, 16] or [l for _, l in self.irreps] != [0, 1, 2]:
            raise NotImplementedError(f"UnetLayer is instantiated for 64x0e+32x1e+16x2e (UNet levels 2, 3, mid block), not {irreps}")
        if list(fc_neurons) != [64, 32, 32]:
            raise NotImplementedError(f"UnetLayer is instantiated for fc_neurons [64, 32, 32], not {list(fc_neurons)}")
        if [l for _, l in sh] != [0, 1, 2] or any(m != 1 for m, _ in sh) or num_heads != 4 or irreps_mlp_mid != 3:
            raise NotImplementedError("UnetLayer: irreps_edge_attr 1x0e+1x1e+1x2e, 4 heads, irreps_mlp_mid 3")
        self.fc_neurons, self.num_heads, self.radius = list(fc_neurons), num_heads, float(radius)
        self.dim = irreps_dim(self.irreps)
        for name, t in init_from_spec(unet_layer_param_spec(self.irreps, self.fc_neurons, num_heads), seed=init_seed).items():
            _register(self, name, t)
        self._handle = None
        self._handle_device: Optional[torch.device] = None

    # ------------------------------------------------------------------------------------------------------------------
    def _release(self):
        if self._handle is not None:
            _lib.load().dedf_destroy(self._handle)
            self._handle = None

    def load_state_dict(self, *a, **k):
        r = super().load_state_dict(*a, **k)
        self._release()                       # the packed device image is rebuilt on next use
        return r

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
        ccfg = _lib.make_unet_layer_config(self.radius, idx, self.fc_neurons, [m for m, _ in self.irreps], self.num_heads)
        blob = _lib.pack_params(ccfg, {k: v for k, v in self.state_dict().items()})
        h = C.c_void_p()
        rc = lib.dedf_create(C.byref(ccfg), blob.ctypes.data_as(C.POINTER(C.c_float)), blob.size, C.byref(h))
        if rc != _lib.OK:
            _lib.raise_for(lib, None, rc, "dedf_create failed (see stderr)")
        self._handle, self._handle_device = h, device

    @torch.no_grad()
    def forward(self, node_coord_src: torch.Tensor, node_input_src: torch.Tensor, node_coord_dst: torch.Tensor,
                node_input_dst: torch.Tensor, edge_src: torch.Tensor, edge_dst: torch.Tensor) -> torch.Tensor:
        """Arguments as the reference has them at the call site (``unet_feature_extractor.py:289-302``): coordinates and features of
        both node sets and the edge lists (int64, sorted by ``edge_dst`` as ``FpsPool`` / ``RadiusGraph`` return them)."""
        assert node_coord_src.ndim == 2 and node_coord_src.shape[-1] == 3 and node_coord_dst.ndim == 2 and node_coord_dst.shape[-1] == 3
        assert node_input_src.shape == (len(node_coord_src), self.dim), f"{node_input_src.shape}"
        assert node_input_dst.shape == (len(node_coord_dst), self.dim), f"{node_input_dst.shape}"
        assert edge_src.ndim == 1 and edge_src.shape == edge_dst.shape
        dev = node_coord_src.device
        self._ensure_handle(dev)
        lib = _lib.load()
        f32 = lambda t: t.detach().to(device=dev, dtype=torch.float32).contiguous()
        xs, fs, xd, fd = f32(node_coord_src), f32(node_input_src), f32(node_coord_dst), f32(node_input_dst)
        es = edge_src.detach().to(device=dev, dtype=torch.int64).contiguous()
        ed = edge_dst.detach().to(device=dev, dtype=torch.int64).contiguous()
        out = torch.empty(len(xd), self.dim, device=dev, dtype=torch.float32)
        stream = C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)
        rc = lib.dedf_layer_forward(self._handle, len(xs), xs.data_ptr(), fs.data_ptr(), len(xd), xd.data_ptr(), fd.data_ptr(),
                                    len(es), es.data_ptr(), ed.data_ptr(), out.data_ptr(), stream)
        _lib.raise_for(lib, self._handle, rc, "dedf_layer_forward")
        return out.to(node_input_dst.dtype)
