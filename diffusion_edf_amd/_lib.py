"""ctypes binding of libdedf.so (C ABI in include/dedf.h).  There is NO fallback: if the HIP library is missing or
no gfx950 device is present the product raises."""
from __future__ import annotations

import ctypes as C
import os
from typing import Optional

import numpy as np

from .params import HeadConfig

_LIB_PATH = os.environ.get("DEDF_LIB") or os.path.join(os.path.dirname(os.path.abspath(__file__)), "csrc", "libdedf.so")
MAX_SCALES = 8

OK, ERR_INVALID, ERR_UNSUPPORTED, ERR_RUNTIME = 0, 1, 2, 3
ABI_VERSION = 6          # DEDF_ABI_VERSION of include/dedf.h this binding mirrors

SYMBOLS = [
    "dedf_version", "dedf_abi_version", "dedf_struct_size", "dedf_param_count", "dedf_param_name", "dedf_param_numel", "dedf_create", "dedf_destroy",
    "dedf_last_error", "dedf_set_key_clouds", "dedf_set_key_weights", "dedf_set_query", "dedf_score", "dedf_energy", "dedf_sample", "dedf_get_stats",
    "dedf_debug_enable", "dedf_debug_copy", "dedf_debug_packed", "dedf_profile_enable", "dedf_profile_read",
    "dedf_fps", "dedf_radius", "dedf_radius_scratch_bytes", "dedf_layer_forward", "dedf_linear_rs", "dedf_field", "dedf_keypoint_weight", "dedf_layer_defer_check", "dedf_layer_check", "dedf_layer_share_workspace", "dedf_set_radial_table", "dedf_linear_rs_lmax",
]


class DedfConfig(C.Structure):
    _fields_ = [
        ("lmax", C.c_int), ("mul", C.c_int * 4), ("num_heads", C.c_int), ("fc_neurons", C.c_int * 3),
        ("length_emb_dim", C.c_int), ("time_emb_mlp", C.c_int * 3), ("irreps_mlp_mid", C.c_int), ("n_scales", C.c_int),
        ("radii", C.c_float * MAX_SCALES), ("r_mincut_nonscalar_sh", C.c_float), ("length_enc_max_r", C.c_float),
        ("max_time", C.c_float), ("time_enc_n", C.c_float), ("lin_mult", C.c_float), ("ang_mult", C.c_float),
        ("max_neighbors", C.c_int), ("device", C.c_int), ("max_edges", C.c_int64), ("ebm", C.c_int), ("half_gemm", C.c_int), ("use_src_point_attn", C.c_int),
        ("unet_layer", C.c_int), ("unet_valid", C.c_int * 4), ("unet_fc_valid", C.c_int * 3), ("unet_narrow", C.c_int),
        ("query_time_encoding", C.c_int),
    ]


class DedfSchedule(C.Structure):
    _fields_ = [("n_steps", C.c_int), ("t", C.POINTER(C.c_double)), ("alpha_ang", C.POINTER(C.c_double)),
                ("alpha_lin", C.POINTER(C.c_double)), ("temperature", C.POINTER(C.c_double))]


class DedfStats(C.Structure):
    _fields_ = [("n_dst", C.c_int64), ("n_edges", C.c_int64 * MAX_SCALES), ("n_edges_total", C.c_int64), ("overflow", C.c_int), ("nonfinite", C.c_int),
                ("rtab_err", C.c_float * MAX_SCALES), ("rtab_fallback", C.c_int), ("sample_retries", C.c_int), ("edges_per_dst_capacity", C.c_int)]


class DedfProfile(C.Structure):
    _fields_ = [("n_evals", C.c_int64), ("n_edges", C.c_int64), ("n_dst", C.c_int64), ("ms", C.c_double * 6)]


PROF_CLASSES = ["pose_time", "neighbors", "edge", "aggregate", "node", "reduce"]

_lib: Optional[C.CDLL] = None


def lib_path() -> str:
    return _LIB_PATH


def load() -> C.CDLL:
    """Load libdedf.so and declare the prototypes.  Raises (no CPU path) if the library has not been built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(_LIB_PATH):
        raise RuntimeError(f"{_LIB_PATH} not found: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                           "(hipcc --offload-arch=gfx950). diffusion_edf_amd has no CPU fallback.")
    lib = C.CDLL(_LIB_PATH)
    P = C.POINTER
    lib.dedf_version.restype = C.c_char_p
    # the structs below are mirrored by hand: refuse a library built from another version of include/dedf.h (dedf_get_stats and
    # dedf_profile_read write sizeof(struct) bytes into the caller's memory)
    if not hasattr(lib, "dedf_abi_version"):
        raise RuntimeError(f"{_LIB_PATH} predates dedf_abi_version (ABI < 3); rebuild it")
    lib.dedf_abi_version.restype = C.c_int
    lib.dedf_struct_size.argtypes = [C.c_int]; lib.dedf_struct_size.restype = C.c_size_t
    if lib.dedf_abi_version() != ABI_VERSION:
        raise RuntimeError(f"{_LIB_PATH} has ABI version {lib.dedf_abi_version()}, this binding mirrors {ABI_VERSION}; rebuild the library")
    for which, cls in enumerate((DedfConfig, DedfSchedule, DedfStats, DedfProfile)):
        if lib.dedf_struct_size(which) != C.sizeof(cls):
            raise RuntimeError(f"{cls.__name__}: the library's struct has {lib.dedf_struct_size(which)} bytes, the ctypes mirror {C.sizeof(cls)}")
    lib.dedf_param_count.argtypes = [P(DedfConfig)]; lib.dedf_param_count.restype = C.c_int
    lib.dedf_param_name.argtypes = [P(DedfConfig), C.c_int]; lib.dedf_param_name.restype = C.c_char_p
    lib.dedf_param_numel.argtypes = [P(DedfConfig), C.c_int]; lib.dedf_param_numel.restype = C.c_size_t
    lib.dedf_create.argtypes = [P(DedfConfig), P(C.c_float), C.c_size_t, P(C.c_void_p)]; lib.dedf_create.restype = C.c_int
    lib.dedf_destroy.argtypes = [C.c_void_p]; lib.dedf_destroy.restype = None
    lib.dedf_last_error.argtypes = [C.c_void_p]; lib.dedf_last_error.restype = C.c_char_p
    lib.dedf_set_key_clouds.argtypes = [C.c_void_p, C.c_int, P(C.c_int), P(C.c_void_p), P(C.c_void_p), C.c_void_p]
    lib.dedf_set_key_clouds.restype = C.c_int
    lib.dedf_set_key_weights.argtypes = [C.c_void_p, C.c_int, P(C.c_int), P(C.c_void_p), C.c_void_p]; lib.dedf_set_key_weights.restype = C.c_int
    lib.dedf_set_query.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]; lib.dedf_set_query.restype = C.c_int
    lib.dedf_score.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]; lib.dedf_score.restype = C.c_int
    lib.dedf_energy.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]; lib.dedf_energy.restype = C.c_int
    lib.dedf_sample.argtypes = [C.c_void_p, C.c_int, C.c_void_p, P(DedfSchedule), C.c_uint64, C.c_int64, C.c_void_p, C.c_void_p, C.c_void_p]
    lib.dedf_sample.restype = C.c_int
    lib.dedf_fps.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p]; lib.dedf_fps.restype = C.c_int
    lib.dedf_radius.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_float, C.c_int, C.c_int, C.c_int64, C.c_void_p, C.c_void_p,
                                P(C.c_int64), C.c_void_p, C.c_size_t, C.c_void_p]
    lib.dedf_radius.restype = C.c_int
    lib.dedf_radius_scratch_bytes.argtypes = [C.c_int]; lib.dedf_radius_scratch_bytes.restype = C.c_size_t
    lib.dedf_layer_forward.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p,
                                       C.c_void_p, C.c_void_p]
    lib.dedf_layer_forward.restype = C.c_int
    lib.dedf_linear_rs.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, P(C.c_int), C.c_void_p, C.c_void_p]
    lib.dedf_linear_rs.restype = C.c_int
    lib.dedf_linear_rs_lmax.argtypes = [C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, P(C.c_int), C.c_void_p, C.c_void_p]
    lib.dedf_linear_rs_lmax.restype = C.c_int
    lib.dedf_set_radial_table.argtypes = [C.c_void_p, C.c_int]; lib.dedf_set_radial_table.restype = C.c_int
    lib.dedf_layer_defer_check.argtypes = [C.c_void_p, C.c_int]; lib.dedf_layer_defer_check.restype = C.c_int
    lib.dedf_layer_check.argtypes = [C.c_void_p, C.c_void_p]; lib.dedf_layer_check.restype = C.c_int
    lib.dedf_layer_share_workspace.argtypes = [C.c_void_p, C.c_void_p]; lib.dedf_layer_share_workspace.restype = C.c_int
    lib.dedf_field.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]; lib.dedf_field.restype = C.c_int
    lib.dedf_keypoint_weight.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                         C.c_float, C.c_int, C.c_float, C.c_void_p, C.c_void_p]
    lib.dedf_keypoint_weight.restype = C.c_int
    lib.dedf_get_stats.argtypes = [C.c_void_p, P(DedfStats)]; lib.dedf_get_stats.restype = C.c_int
    lib.dedf_debug_enable.argtypes = [C.c_void_p, C.c_int]; lib.dedf_debug_enable.restype = C.c_int
    lib.dedf_debug_copy.argtypes = [C.c_void_p, C.c_char_p, C.c_void_p, C.c_size_t, P(C.c_size_t)]; lib.dedf_debug_copy.restype = C.c_int
    lib.dedf_debug_packed.argtypes = [C.c_void_p, C.c_char_p, P(P(C.c_float)), P(C.c_size_t)]; lib.dedf_debug_packed.restype = C.c_int
    lib.dedf_profile_enable.argtypes = [C.c_void_p, C.c_int]; lib.dedf_profile_enable.restype = C.c_int
    lib.dedf_profile_read.argtypes = [C.c_void_p, P(DedfProfile)]; lib.dedf_profile_read.restype = C.c_int
    _lib = lib
    return lib


def make_config(cfg: HeadConfig, device: int, max_edges: int = 0) -> DedfConfig:
    if cfg.n_scales > MAX_SCALES:
        raise ValueError(f"{cfg.n_scales} scales: the library takes at most {MAX_SCALES} (dedf_config.radii)")
    c = DedfConfig()
    c.lmax = cfg.lmax
    for i, m in enumerate(cfg.muls):
        c.mul[i] = m
    c.num_heads = cfg.num_heads
    for i in range(3):
        c.fc_neurons[i] = cfg.fc_neurons[i]
        c.time_emb_mlp[i] = cfg.time_emb_mlp[i]
    c.length_emb_dim = cfg.length_emb_dim
    c.irreps_mlp_mid = cfg.irreps_mlp_mid
    c.n_scales = cfg.n_scales
    for i, r in enumerate(cfg.radii):
        c.radii[i] = -1.0 if r is None else float(r)
    c.r_mincut_nonscalar_sh = cfg.r_mincut_nonscalar_sh
    c.length_enc_max_r = cfg.length_enc_max_r
    c.max_time, c.time_enc_n = cfg.max_time, cfg.time_enc_n
    c.lin_mult, c.ang_mult = cfg.lin_mult, cfg.ang_mult
    c.max_neighbors = cfg.max_neighbors
    c.device = device
    c.max_edges = max_edges
    c.ebm = int(cfg.ebm)
    c.half_gemm = int(getattr(cfg, 'half_gemm', False))
    c.use_src_point_attn = int(getattr(cfg, 'use_src_point_attn', False))
    c.unet_layer = 0
    c.query_time_encoding = int(getattr(cfg, 'query_time_encoding', False))
    return c


def make_unet_layer_config(radius: float, device: int, fc_neurons=(64, 32, 32), muls=(64, 32, 16), num_heads: int = 4,
                           irreps_mlp_mid: int = 3, valid=None, fc_valid=None, half_gemm: bool = False, narrow: bool = False) -> DedfConfig:
    """dedf_config of ONE UNet layer (dedf_config.unet_layer = 1): irreps 64x0e+32x1e+16x2e, radial MLP [64,32,32]; `valid` / `fc_valid`:
    true multiplicities / radial widths of a narrower model that runs zero-padded (dedf.h: unet_valid, unet_fc_valid)"""
    c = DedfConfig()
    c.lmax = len(muls) - 1
    for i, m in enumerate(muls):
        c.mul[i] = m
    c.num_heads = num_heads
    for i in range(3):
        c.fc_neurons[i] = fc_neurons[i]
    c.length_emb_dim = fc_neurons[0]
    c.irreps_mlp_mid = irreps_mlp_mid
    c.n_scales = 1
    c.radii[0] = float(radius)
    c.max_neighbors = 1000
    c.device = device
    c.unet_layer = 1
    c.half_gemm = int(bool(half_gemm))
    c.unet_narrow = int(bool(narrow))
    for i in range(len(muls)):
        c.unet_valid[i] = 0 if valid is None else int(valid[i])
    for i in range(3):
        c.unet_fc_valid[i] = 0 if fc_valid is None else int(fc_valid[i])
    return c


def param_names(ccfg: DedfConfig):
    lib = load()
    n = lib.dedf_param_count(C.byref(ccfg))
    if n < 0:
        raise ValueError("configuration rejected by libdedf")
    return [(lib.dedf_param_name(C.byref(ccfg), i).decode(), int(lib.dedf_param_numel(C.byref(ccfg), i))) for i in range(n)]


def raise_for(lib, handle, rc: int, what: str):
    if rc == OK:
        return
    msg = lib.dedf_last_error(handle).decode() if handle else ""
    if rc == ERR_INVALID:
        raise ValueError(f"{what}: {msg}")
    if rc == ERR_UNSUPPORTED:
        raise NotImplementedError(f"{what}: {msg}")
    raise RuntimeError(f"{what}: {msg}")


def pack_params(ccfg: DedfConfig, state: dict) -> np.ndarray:
    """Concatenate `state` (reference state_dict names below `score_head.`) in the library's canonical order."""
    chunks = []
    for name, numel in param_names(ccfg):
        if name not in state:
            raise KeyError(f"missing parameter {name}")
        a = state[name].detach().cpu().to(dtype=__import__('torch').float32).reshape(-1).numpy()
        if a.size != numel:
            raise ValueError(f"{name}: expected {numel} elements, got {a.size}")
        chunks.append(a)
    return np.ascontiguousarray(np.concatenate(chunks), dtype=np.float32)
