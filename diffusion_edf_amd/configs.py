"""Task front-end: the three files at the top of a reference task directory (``configs/<task>/agent.yaml``, ``server.yaml``,
``preprocess.yaml``) turned into agents and sampling arguments.

What the reference does with them lives in ``agent_server.py:48-86`` (reads the files, builds one pick and one place
``DiffusionEdfAgent``) and ``agent_server.py:187-259`` (feeds ``*_diffusion_configs`` to ``agent.sample``).  The Pyro service
around that is networking and stays out; this module is the part a notebook or a service needs to get from a task directory to
``agent.sample(...)``:

    task = TaskConfigs.load("configs/panda_mug", root="/path/to/reference")
    agent = task.build_agent("pick")                                   # DiffusionEdfAgent with models (extractors included) + critic + proc_fn
    Ts, scene, grasp = agent.sample(scene_pcd, grasp_pcd, T_seed, **task.sample_kwargs("pick"))

Paths inside agent.yaml (``configs_root_dir``, ``checkpoint_dir``) are relative to the reference repository root; ``root``
says where that is.  Checkpoints are optional: every ``*.pt`` of the reference is a git-LFS stub in this build environment, so
``checkpoints="skip"`` builds the models with their seeded init (``"require"`` is the production setting).
"""
from __future__ import annotations

import os
from dataclasses import dataclass, field
from typing import Any, Callable, Dict, List, Optional

import yaml

from .preprocess import compose_proc_fn

TASKS = ("pick", "place")


def _read(path: str) -> Dict[str, Any]:
    with open(path) as f:
        return yaml.safe_load(f) or {}


def _is_lfs_stub(path: str) -> bool:
    try:
        with open(path, "rb") as f:
            return f.read(40).startswith(b"version https://git-lfs")
    except OSError:
        return False


@dataclass
class ModelEntry:
    """one entry of ``*_models_kwargs`` / ``*_critic_kwargs`` of agent.yaml = the keyword arguments of ``get_models``"""
    configs_root_dir: str
    train_configs_file: str = "train_configs.yaml"
    task_configs_file: str = "task_configs.yaml"
    checkpoint_dir: Optional[str] = None
    n_warmups: int = 10

    def resolved(self, root: str, checkpoints: str) -> Dict[str, Any]:
        cfg_dir = self.configs_root_dir if os.path.isabs(self.configs_root_dir) else os.path.join(root, self.configs_root_dir)
        ckpt = self.checkpoint_dir
        if ckpt is not None and not os.path.isabs(ckpt):
            ckpt = os.path.join(root, ckpt)
        if checkpoints == "skip":
            ckpt = None
        elif checkpoints == "if_present":
            if ckpt is None or not os.path.isfile(ckpt) or _is_lfs_stub(ckpt):
                ckpt = None
        elif checkpoints == "require":
            if ckpt is None or not os.path.isfile(ckpt):
                raise FileNotFoundError(f"checkpoint {ckpt} not found")
            if _is_lfs_stub(ckpt):
                raise FileNotFoundError(f"checkpoint {ckpt} is a git-LFS pointer, not the weights (run `git lfs pull` in the reference)")
        else:
            raise ValueError(f"checkpoints must be 'require', 'if_present' or 'skip', not {checkpoints!r}")
        return dict(configs_root_dir=cfg_dir, train_configs_file=self.train_configs_file, task_configs_file=self.task_configs_file,
                    checkpoint_dir=ckpt, n_warmups=self.n_warmups)


@dataclass
class TaskConfigs:
    root: str
    device: str
    models: Dict[str, List[ModelEntry]]
    critic: Dict[str, Optional[ModelEntry]]
    server: Dict[str, Any] = field(default_factory=dict)
    preprocess_config: List[Dict[str, Any]] = field(default_factory=list)
    unprocess_config: List[Dict[str, Any]] = field(default_factory=list)

    @classmethod
    def load(cls, task_dir: str, root: Optional[str] = None) -> "TaskConfigs":
        """``task_dir``: directory holding agent.yaml / server.yaml / preprocess.yaml.  ``root``: what the relative paths inside
        agent.yaml are relative to (default: two levels above ``task_dir``, i.e. the repository that holds ``configs/<task>``)."""
        task_dir = os.path.abspath(task_dir)
        root = os.path.abspath(root) if root is not None else os.path.dirname(os.path.dirname(task_dir))
        agent = _read(os.path.join(task_dir, "agent.yaml"))
        mk = agent.get("model_kwargs") or {}
        models, critic = {}, {}
        for task in TASKS:
            models[task] = [ModelEntry(**e) for e in (mk.get(f"{task}_models_kwargs") or [])]
            c = mk.get(f"{task}_critic_kwargs")
            critic[task] = ModelEntry(**c) if c else None
        server_path, pre_path = os.path.join(task_dir, "server.yaml"), os.path.join(task_dir, "preprocess.yaml")
        server = _read(server_path) if os.path.isfile(server_path) else {}
        pre = _read(pre_path) if os.path.isfile(pre_path) else {}
        return cls(root=root, device=str(agent.get("device", "cuda")), models=models, critic=critic, server=server,
                   preprocess_config=list(pre.get("preprocess_config") or []), unprocess_config=list(pre.get("unprocess_config") or []))

    # ------------------------------------------------------------------------------------------------------------------
    def sample_kwargs(self, task: str) -> Dict[str, Any]:
        """the ``<task>_diffusion_configs`` block of server.yaml = keyword arguments of ``DiffusionEdfAgent.sample``
        (``N_steps_list``, ``timesteps_list``, ``temperatures_list``, ``diffusion_schedules_list``, ``log_t_schedule``,
        ``time_exponent_temp``, ``time_exponent_alpha``); reference agent_server.py:204-213"""
        if task not in TASKS:
            raise ValueError(f"Unknown task name: {task}")
        return dict(self.server[f"{task}_diffusion_configs"])

    def n_denoising_steps(self, task: str) -> int:
        return sum(sum(n) for n in self.sample_kwargs(task)["N_steps_list"])

    def build_agent(self, task: str, *, extractors: Optional[List[Dict[str, Callable]]] = None,
                    critic_extractors: Optional[Dict[str, Callable]] = None, device: Optional[str] = None,
                    checkpoints: str = "require", half_precision: bool = False, n_warmups: Optional[int] = None,
                    proc_registry: Optional[Dict[str, Callable]] = None):
        """``DiffusionEdfAgent`` of one task: its score models (low-res → high-res cascade), its critic if the file names one, and
        the pre- / un-processing pipelines of preprocess.yaml.

        The models build their feature extractors from the YAML blocks (``agent.py``); ``extractors[i]`` = ``{"key_extractor": ...,
        "query_extractor": ...}`` for model i (``critic_extractors`` likewise) overrides them, e.g. with ``PrecomputedFeatures``."""
        from .agent import DiffusionEdfAgent
        if task not in TASKS:
            raise ValueError(f"Unknown task name: {task}")
        entries = self.models[task]
        if not entries:
            raise ValueError(f"agent.yaml has no {task}_models_kwargs")
        extractors = list(extractors) if extractors is not None else [{} for _ in entries]
        if len(extractors) != len(entries):
            raise ValueError(f"{len(entries)} {task} models but {len(extractors)} extractor sets")

        def kw(entry: ModelEntry, ex: Optional[Dict[str, Callable]]):
            d = entry.resolved(self.root, checkpoints)
            if n_warmups is not None:
                d["n_warmups"] = n_warmups
            d.update(ex or {})
            return d

        c = self.critic[task]
        return DiffusionEdfAgent(model_kwargs_list=[kw(e, x) for e, x in zip(entries, extractors)],
                                 preprocess_config=self.preprocess_config, unprocess_config=self.unprocess_config,
                                 device=device or self.device, half_precision=half_precision,
                                 critic_kwargs=kw(c, critic_extractors) if c is not None else None, proc_registry=proc_registry)


def list_task_dirs(configs_dir: str) -> List[str]:
    """every directory below ``configs_dir`` that holds an agent.yaml"""
    out = []
    for name in sorted(os.listdir(configs_dir)):
        d = os.path.join(configs_dir, name)
        if os.path.isfile(os.path.join(d, "agent.yaml")):
            out.append(d)
    return out
