"""Callers of the hot path: model assembly from the reference's YAML files and checkpoints, and the multi-model denoising
cascade with critic ranking.

Mirrors, for everything that touches the score head (same names, argument meaning, assertion messages):

  reference diffusion_edf/multiscale_score_model.py:27-135   -> ``MultiscaleScoreModel``
  reference diffusion_edf/point_attentive_score_model.py:23-110 -> ``PointAttentiveScoreModel``
  reference diffusion_edf/trainer.py:35-70, 124-147           -> ``load_configs`` / checkpoint loading in ``get_models``
  reference diffusion_edf/agent.py:20-64                      -> ``get_models``
  reference diffusion_edf/agent.py:66-186                     -> ``DiffusionEdfAgent`` (``compute_critic_energy``, ``sample``)

The feature extractors are built from the YAML blocks like the reference builds them — ``UnetFeatureExtractor``
/ ``ForwardOnlyFeatureExtractor`` (``unet.py``), ``KeypointExtractor`` and ``StaticKeypointModel`` (``keypoint_extractor.py``), all on the
HIP kernels — so a reference
``score_model_state_dict`` loads into ``key_model.*`` / ``query_model.*`` / ``score_head.*`` by name.  They can also be *injected*: every model
takes a ``key_extractor`` and a ``query_extractor`` — any callable ``FeaturedPoints -> List[FeaturedPoints]`` / ``FeaturedPoints ->
FeaturedPoints`` with an ``irreps_output`` attribute, e.g. ``PrecomputedFeatures`` below.  What is NOT here: ``edf_interface`` (PointCloud, SE3
types; the pre-processing steps its YAML names are in ``preprocess.py``, the task-level files agent.yaml / server.yaml in ``configs.py``),
and the Pyro service.
"""
from __future__ import annotations

import copy
import os
from typing import Any, Callable, Dict, List, Optional, Sequence, Tuple, Union

import torch
import yaml

from . import dist as ddist
from .gnn_data import FeaturedPoints
from .keypoint_extractor import KeypointExtractor, StaticKeypointModel
from .score_head import EbmScoreModelHead, ScoreModelHead
from .score_model_base import ScoreModelBase


class PrecomputedFeatures(torch.nn.Module):
    """Extractor stand-in for features computed elsewhere: returns what it was given, whatever the input cloud."""

    def __init__(self, output, irreps_output: str):
        super().__init__()
        self.output = output
        self.irreps_output = irreps_output

    def forward(self, pcd=None):
        return self.output


class MultiscaleScoreModel(ScoreModelBase):
    """reference multiscale_score_model.py:27-135 with the two extractors injected instead of built.

    ``score_head_kwargs`` / ``key_kwargs`` / ``query_kwargs`` are the blocks of ``score_model_configs.yaml`` unchanged; the
    keys the reference injects into ``key_tensor_field_kwargs`` (``irreps_input``, ``use_src_point_attn``,
    ``use_dst_point_attn``, :79-85) and ``irreps_query_edf`` (:93, :104) are injected here the same way."""
    _use_src_point_attn = False

    def __init__(self,
                 query_model: str,
                 score_head_kwargs: Dict,
                 key_kwargs: Dict,
                 query_kwargs: Dict,
                 deterministic: bool = False,
                 key_extractor: Optional[Callable] = None,
                 query_extractor: Optional[Callable] = None):
        super().__init__()
        key_name = key_kwargs['feature_extractor_name']
        if not self._use_src_point_attn and key_name not in ('UnetFeatureExtractor', 'ForwardOnlyFeatureExtractor'):
            raise ValueError(f"Unknown feature extractor name: {key_name}")                    # :51
        if query_model not in ('KeypointExtractor', 'StaticKeypointModel'):
            raise ValueError(f"Unknown query model: {query_model}")                            # :62
        if key_extractor is None and self._use_src_point_attn:
            key_extractor = KeypointExtractor(**copy.deepcopy(key_kwargs), deterministic=deterministic)     # point_attentive_score_model.py:34-37
        elif key_extractor is None:
            from .unet import ForwardOnlyFeatureExtractor, UnetFeatureExtractor
            cls = UnetFeatureExtractor if key_name == 'UnetFeatureExtractor' else ForwardOnlyFeatureExtractor
            key_extractor = cls(**key_kwargs['feature_extractor_kwargs'], deterministic=deterministic)     # :40-51
        self.key_model = key_extractor
        if query_extractor is None and query_model == 'StaticKeypointModel':
            query_extractor = StaticKeypointModel(**query_kwargs)                               # :58-60
        elif query_extractor is None and query_model == 'KeypointExtractor':
            query_extractor = KeypointExtractor(**copy.deepcopy(query_kwargs), deterministic=deterministic)                  # :54-57
        self.query_model = query_extractor
        key_irreps = getattr(key_extractor, 'irreps_output', None) or key_kwargs['feature_extractor_kwargs']['irreps_output']
        if getattr(query_extractor, 'irreps_output', None):
            query_irreps = query_extractor.irreps_output
        elif query_model == 'KeypointExtractor':
            query_irreps = query_kwargs['feature_extractor_kwargs']['irreps_output']
        else:
            query_irreps = query_kwargs['irreps_output']

        score_head_kwargs = copy.deepcopy(score_head_kwargs)       # the reference edits the caller's dict in place; the values are the same
        max_time = float(score_head_kwargs['max_time'])
        time_emb_mlp: List[int] = score_head_kwargs['time_emb_mlp']
        if 'lin_mult' not in score_head_kwargs or 'ang_mult' not in score_head_kwargs:
            raise NotImplementedError()                                                         # :68-75
        tf = score_head_kwargs['key_tensor_field_kwargs']
        assert 'irreps_input' not in tf.keys()
        tf['irreps_input'] = key_irreps
        assert 'use_src_point_attn' not in tf.keys()
        tf['use_src_point_attn'] = self._use_src_point_attn
        assert 'use_dst_point_attn' not in tf.keys()
        tf['use_dst_point_attn'] = False
        head_cls = EbmScoreModelHead if score_head_kwargs.get("ebm", False) else ScoreModelHead
        extra = {k: score_head_kwargs[k] for k in ('time_enc_n',) if k in score_head_kwargs}
        self.score_head = head_cls(max_time=max_time, time_emb_mlp=time_emb_mlp, key_tensor_field_kwargs=tf,
                                   irreps_query_edf=query_irreps, lin_mult=float(score_head_kwargs['lin_mult']),
                                   ang_mult=float(score_head_kwargs['ang_mult']),
                                   edge_time_encoding=score_head_kwargs['edge_time_encoding'],
                                   query_time_encoding=score_head_kwargs['query_time_encoding'], **extra)
        self.lin_mult = self.score_head.lin_mult
        self.ang_mult = self.score_head.ang_mult
        self.diffusion_schedules = None          # set by get_models from train_configs (agent.py:48)

    def _extract(self, which: str, fn, pcd):
        if fn is None:
            raise NotImplementedError(f"{which}: no extractor")
        return fn(pcd)

    def get_key_pcd_multiscale(self, pcd) -> List[FeaturedPoints]:                              # :130-131
        return self._extract("get_key_pcd_multiscale", self.key_model, pcd)

    def get_query_pcd(self, pcd) -> FeaturedPoints:                                            # :133-134
        return self._extract("get_query_pcd", self.query_model, pcd)

    def load_state_dict(self, state_dict, strict: bool = True):
        """Takes the reference's full ``score_model_state_dict``: ``score_head.*`` goes to the HIP head (names and shapes as
        the reference's modules register them), ``key_model.*`` / ``query_model.*`` to the injected extractors when they are
        modules; with ``strict=False`` (the reference agent's default, agent.py:28) anything else is skipped."""
        r = super().load_state_dict(state_dict, strict=strict)
        self.score_head.refresh_weights()
        return r


class PointAttentiveScoreModel(MultiscaleScoreModel):
    """reference point_attentive_score_model.py:23-110: the key model is ONE KeypointExtractor whose output is the single key
    cloud (``get_key_pcd_multiscale`` wraps it in a list, :106-107) and whose point weights ``w`` multiply the attention of every
    edge after the softmax (``use_src_point_attn=True``, :71-72; gnn_block.py:190-194)."""
    _use_src_point_attn = True

    def get_key_pcd_multiscale(self, pcd) -> List[FeaturedPoints]:
        out = self._extract("get_key_pcd_multiscale", self.key_model, pcd)
        return [out] if isinstance(out, FeaturedPoints) else list(out)       # (FeaturedPoints is itself a tuple)


# Persistent BUFFERS a reference ``score_model_state_dict`` carries and this build's schema does not need (they hold constants, no trained
# state): e3nn ``o3.TensorProduct.output_mask`` and the constants of its generated code (``_compiled_main_*``, ``_w3j_*``), the
# ``cutoff_eps`` of every graph parser (graph_parser.py:37), the Wigner ``J`` matrices of ``SliceAndTransform`` (wigner.py:215) and
# ``ParityInversionSh.sign`` (utils.py:44).
_REFERENCE_ONLY_SUFFIXES = ('.output_mask', '.cutoff_eps', '.J', '.sign')
_REFERENCE_ONLY_MARKERS = ('._compiled_main', '._w3j_', '.tp._w3j')


def is_reference_only_buffer(key: str) -> bool:
    return key.endswith(_REFERENCE_ONLY_SUFFIXES) or any(m in key for m in _REFERENCE_ONLY_MARKERS)


def load_configs(configs_root_dir: str, train_configs_file: str, task_configs_file: str) -> Dict[str, Dict]:
    """The three YAML files of one model directory, as reference trainer.py:41-48 reads them."""
    with open(os.path.join(configs_root_dir, train_configs_file)) as f:
        train = yaml.load(f, Loader=yaml.FullLoader)
    with open(os.path.join(configs_root_dir, task_configs_file)) as f:
        task = yaml.load(f, Loader=yaml.FullLoader)
    with open(os.path.join(configs_root_dir, train['model_config_file'])) as f:
        model = yaml.load(f, Loader=yaml.FullLoader)
    return dict(train=train, task=task, model=model)


def get_models(configs_root_dir: str,
               train_configs_file: str,
               task_configs_file: str,
               checkpoint_dir: Optional[str],
               device: str,
               n_warmups: int = 10,
               compile_score_head: bool = False,
               strict_load: bool = False,
               half_precision: bool = False,
               key_extractor: Optional[Callable] = None,
               query_extractor: Optional[Callable] = None) -> MultiscaleScoreModel:
    """reference agent.py:20-64.  ``compile_score_head`` is accepted and has no effect (the head is a HIP library, there is
    nothing to script); the warm-up runs the head's own fake input (score_head.py:213-218) instead of demo batches from the
    out-of-scope data loader."""
    cfgs = load_configs(configs_root_dir, train_configs_file, task_configs_file)
    name = cfgs['model']['model_name']
    if name == 'PointAttentiveScoreModel':                                                     # trainer.py:134-139
        model_cls = PointAttentiveScoreModel
    elif name == 'MultiscaleScoreModel':
        model_cls = MultiscaleScoreModel
    else:
        raise ValueError(f"Unknown score model name: {name}")
    model = model_cls(**cfgs['model']['model_kwargs'], deterministic=False, key_extractor=key_extractor, query_extractor=query_extractor)
    if checkpoint_dir is not None:                                                              # trainer.py:141-147
        checkpoint = torch.load(checkpoint_dir, map_location='cpu')
        sd = checkpoint['score_model_state_dict']
        if strict_load:             # strict means strict about PARAMETERS: the reference-only buffers below are not part of this build's schema
            own = set(model.state_dict())
            sd = {k: v for k, v in sd.items() if k in own or not is_reference_only_buffer(k)}
        res = model.load_state_dict(sd, strict=strict_load)
        # strict_load=False (the reference's default, agent.py:28) tolerates missing / extra keys of INJECTED feature extractors.
        # Everything this build constructs itself -- the score head always, the key / query models unless they were injected -- is
        # checked: a parameter that does not line up means the schema differs from the checkpoint's and that module would silently run
        # on its seeded init, which is never acceptable.  Reference-only persistent buffers (e3nn's `output_mask` / compiled-code
        # constants, `cutoff_eps` of graph_parser.py:37, the Wigner `J` of wigner.py:215, ParityInversionSh's `sign`) carry no trained
        # state and are tolerated.
        built = ['score_head.'] + (['key_model.'] if key_extractor is None else []) + (['query_model.'] if query_extractor is None else [])
        mine = lambda k: any(k.startswith(p) for p in built)
        missing = [k for k in res.missing_keys if mine(k) and not is_reference_only_buffer(k)]
        unexpected = [k for k in res.unexpected_keys if mine(k) and not is_reference_only_buffer(k)]
        if missing or unexpected:
            raise RuntimeError(f"checkpoint {checkpoint_dir}: parameters of {built} do not match the schema of this build: "
                               f"missing {missing[:8]}, unexpected {unexpected[:8]}")
        print(f"Successfully Loaded checkpoint @ epoch: {checkpoint['epoch']} (steps: {checkpoint['steps']})")
    model = model.to(device).eval()
    model.diffusion_schedules = cfgs['train']['diffusion_configs']['time_schedules']
    if half_precision:              # agent.py:50-51 `model.half()`: here the half-precision GEMM mode of the head and of the extractors' layers
        model.score_head.half()
        for m in (model.key_model, model.query_model):
            if isinstance(m, torch.nn.Module) and type(m).half is not torch.nn.Module.half:
                m.half()
    if n_warmups and torch.device(device).type == 'cuda':
        print(f"Warming up the model for {n_warmups} iterations", flush=True)
        fake = model.score_head._get_fake_input()
        for _ in range(n_warmups):
            model.score_head.warmup(*fake)
    return model


class DiffusionEdfAgent():
    """reference agent.py:66-186.  ``models`` / ``critic`` may be passed ready-made; otherwise they are built with
    ``get_models`` from ``model_kwargs_list`` / ``critic_kwargs`` (whose dicts may carry ``key_extractor`` / ``query_extractor``).
    ``proc_fn`` / ``unprocess_fn`` stand where the reference composes ``edf_interface`` pre-processing (identity by default).
    With ``torch.distributed`` initialised over more than one rank the poses of every stage are sharded over the ranks
    (``dist.sample_sharded``) and every rank returns the full result."""

    def __init__(self, model_kwargs_list: Optional[List[Dict]] = None,
                 preprocess_config=None,
                 unprocess_config=None,
                 device: str = 'cuda',
                 compile_score_head: bool = False,
                 half_precision: bool = False,
                 critic_kwargs: Optional[Dict] = None,
                 models: Optional[List[ScoreModelBase]] = None,
                 critic: Optional[ScoreModelBase] = None,
                 proc_fn: Optional[Callable] = None,
                 unprocess_fn: Optional[Callable] = None,
                 proc_registry: Optional[Dict[str, Callable]] = None):
        # reference agent.py:79-80: proc_fn = compose_proc_fn(preprocess_config).  The procs the shipped preprocess.yaml files name
        # (downsample, rescale; crop_bbox) are in preprocess.py; ``proc_registry`` adds / overrides names, explicit ``proc_fn`` /
        # ``unprocess_fn`` callables win over the configs.
        from .preprocess import compose_proc_fn
        if proc_fn is None and preprocess_config:
            proc_fn = compose_proc_fn(preprocess_config, proc_registry)
        if unprocess_fn is None and unprocess_config:
            unprocess_fn = compose_proc_fn(unprocess_config, proc_registry)
        if critic is not None:
            self.critic = critic
        elif critic_kwargs is not None:
            self.critic = get_models(**critic_kwargs, device=device, compile_score_head=compile_score_head, half_precision=half_precision)
        else:
            self.critic = None
        self.models = list(models) if models is not None else []
        for kwargs in (model_kwargs_list or []):
            self.models.append(get_models(**kwargs, device=device, compile_score_head=compile_score_head, half_precision=half_precision))
        self.proc_fn = proc_fn if proc_fn is not None else (lambda x: x)
        self.unprocess_fn = unprocess_fn if unprocess_fn is not None else (lambda x: x)

    @torch.no_grad()
    def compute_critic_energy(self, key_pcd, query_pcd, Ts, time) -> torch.Tensor:              # :87-96
        key_pcd_multiscale = self.critic.get_key_pcd_multiscale(key_pcd)
        query_pcd = self.critic.get_query_pcd(query_pcd)
        return self.critic.score_head.compute_energy(Ts=Ts, key_pcd_multiscale=key_pcd_multiscale, query_pcd=query_pcd, time=time)

    @torch.no_grad()
    def sample(self, scene_pcd, grasp_pcd, Ts_init,
               N_steps_list: List[List[int]],
               timesteps_list: List[List[float]],
               temperatures_list: List[Union[Union[int, float], Sequence[Union[int, float]]]],
               diffusion_schedules_list: Optional[List[Optional[List[Union[List[float], Tuple[float, float]]]]]] = None,
               log_t_schedule: bool = True,
               time_exponent_temp: float = 1.0,
               time_exponent_alpha: float = 0.5,
               return_info: Optional[bool] = False,
               noise_list: Optional[List[Optional[torch.Tensor]]] = None,
               seed: int = 0):
        """alpha = timestep * L^2 * (t^time_exponent_alpha);  T = temperature * (t^time_exponent_temp)   (agent.py:98-186).

        Returns ``(Ts_out, scene_pcd, grasp_pcd[, info])`` with ``Ts_out`` the concatenated trajectories ``(nTime, nSample, 7)``,
        poses ordered by ascending critic energy when there is a critic.  ``noise_list`` (per model, injected normals) and
        ``seed`` are additions for parity runs / reproducible sharded sampling."""
        if diffusion_schedules_list is None:
            diffusion_schedules_list = [None for _ in range(len(self.models))]
        assert len(self.models) == len(N_steps_list), f"{len(self.models)} != {len(N_steps_list)}"
        assert len(self.models) == len(timesteps_list), f"{len(self.models)} != {len(timesteps_list)}"
        assert len(self.models) == len(temperatures_list), f"{len(self.models)} != {len(temperatures_list)}"
        assert len(self.models) == len(diffusion_schedules_list), f"{len(self.models)} != {len(diffusion_schedules_list)}"
        if noise_list is None:
            noise_list = [None for _ in range(len(self.models))]

        # (composed pipelines are told which input they see: crop_bbox of the sapien task files targets the scene cloud only)
        with_role = getattr(self.proc_fn, "accepts_role", False)
        scene_pcd = self.proc_fn(scene_pcd, role="scene_pcd") if with_role else self.proc_fn(scene_pcd)
        grasp_pcd = self.proc_fn(grasp_pcd, role="grasp_pcd") if with_role else self.proc_fn(grasp_pcd)
        Ts_init = self.proc_fn(Ts_init, role="poses") if with_role else self.proc_fn(Ts_init)
        scene_input, grasp_input = scene_pcd, grasp_pcd
        T0: torch.Tensor = Ts_init.poses if hasattr(Ts_init, 'poses') else Ts_init
        assert T0.ndim == 2 and T0.shape[-1] == 7, f"{T0.shape}"
        sharded = torch.distributed.is_available() and torch.distributed.is_initialized() and torch.distributed.get_world_size() > 1

        info: Dict[str, Any] = {}
        Ts_out = []
        for i, (model, N_steps, timesteps, temperatures, diffusion_schedules) in enumerate(
                zip(self.models, N_steps_list, timesteps_list, temperatures_list, diffusion_schedules_list)):
            scene_out_multiscale = model.get_key_pcd_multiscale(scene_input)
            grasp_out = model.get_query_pcd(grasp_input)
            if diffusion_schedules is None:
                diffusion_schedules = model.diffusion_schedules
            assert len(diffusion_schedules) == len(N_steps), f"{len(diffusion_schedules)} != {len(N_steps)}"
            assert len(diffusion_schedules) == len(timesteps), f"{len(diffusion_schedules)} != {len(timesteps)}"
            kw = dict(diffusion_schedules=diffusion_schedules, N_steps=N_steps, timesteps=timesteps, temperatures=temperatures,
                      log_t_schedule=log_t_schedule, time_exponent_temp=time_exponent_temp, time_exponent_alpha=time_exponent_alpha)
            if sharded:
                assert noise_list[i] is None, "injected noise is a single-process parity tool"
                Ts = ddist.sample_sharded(model, T0.clone().detach(), scene_out_multiscale, grasp_out, gather_trajectory=True,
                                          seed=seed + i, **kw)
            else:
                Ts = model.sample(T_seed=T0.clone().detach(), scene_pcd_multiscale=scene_out_multiscale, grasp_pcd=grasp_out,
                                  noise=noise_list[i], seed=seed + i, **kw)
            Ts = Ts.type(T0.dtype)
            T0 = Ts[-1]
            Ts_out.append(Ts)
        Ts_out = torch.cat(Ts_out, dim=0)          # (nTime, nSample, 7)

        if self.critic is not None:
            key_pcd_multiscale = self.critic.get_key_pcd_multiscale(scene_input)
            query_pcd = self.critic.get_query_pcd(grasp_input)
            # any time value: the critic has no time encoding (agent.py:171)
            energy = self.critic.score_head.compute_energy(Ts=Ts_out[-1, ...], key_pcd_multiscale=key_pcd_multiscale, query_pcd=query_pcd,
                                                           time=torch.ones(Ts_out.shape[-2], device=Ts_out.device, dtype=Ts_out.dtype))
            energy_sorted, idx_sorted = energy.sort(descending=False)
            Ts_out = Ts_out[..., idx_sorted, :]
            info["energy"] = energy_sorted

        if return_info:
            return Ts_out, scene_pcd, grasp_pcd, info
        return Ts_out, scene_pcd, grasp_pcd
