"""Explicit configuration object for the fitting path.

The reference keeps these as module globals in ``config.py`` which opens the model pickle at import
time (reference config.py:48,64-74,81-140).  Here they are fields of a plain object; ``from_tables``
derives the model-dependent ones (N_POSE, N_BETAS, CANONICAL_MODEL_JOINTS, TORSO_JOINTS,
STATIC_JOINT_LOCATIONS) the same way.  ``current`` is the process-wide default used by the drop-in
classes when no explicit config is passed (mirrors ``import config``).
"""
from __future__ import annotations

from dataclasses import dataclass, field
from typing import List, Optional

# reference config.py:64-74 - rows: joint, sil reproj, betas, pose, limits, splay, temporal, iterations, lr
DEFAULT_OPT_WEIGHTS = [
    [25.0, 10.0, 7.5, 5.0],
    [0.0, 500.0, 1000.0, 1000.0],
    [0.0, 1.0, 1.0, 1.0],
    [0.0, 1.0, 1.0, 1.0],
    [0.0, 100.0, 100.0, 100.0],
    [0.0, 0.1, 0.1, 0.1],
    [500.0, 100.0, 100.0, 100.0],
    [600, 400, 600, 600],
    [9e-2, 5e-3, 5e-4, 2e-4],
]

_TORSO_NAMES = ["b_a_1", "l_1_co_r", "l_1_co_l", "b_h", "ma_l", "ma_r"]  # reference config.py:107-109


@dataclass
class FitterConfig:
    SMAL_FILE: Optional[str] = None
    WINDOW_SIZE: int = 10                 # reference config.py:37
    ALLOW_LIMB_SCALING: bool = True       # reference config.py:23
    ignore_sym: bool = True
    ignore_hardcoded_body: bool = True
    DEBUG: bool = False
    STATIC_JOINT_LOCATIONS: bool = False
    N_POSE: int = 0
    N_BETAS: int = 0
    joint_names: List[str] = field(default_factory=list)
    CANONICAL_MODEL_JOINTS: List[int] = field(default_factory=list)
    TORSO_JOINTS: List[int] = field(default_factory=list)
    MESH_COLOR: List[int] = field(default_factory=lambda: [0, 172, 223])
    OPT_WEIGHTS: List[List[float]] = field(default_factory=lambda: [list(r) for r in DEFAULT_OPT_WEIGHTS])
    JOINT_LIMIT: float = 0.01             # joint_limits_prior.py:8-15 (all non-root joints, +-0.01 rad)

    @staticmethod
    def from_tables(tables, smal_file: Optional[str] = None, **overrides) -> "FitterConfig":
        names = list(tables.joint_names)
        cfg = FitterConfig(
            SMAL_FILE=smal_file,
            STATIC_JOINT_LOCATIONS=bool(tables.static_joints),
            N_POSE=tables.J - 1,
            N_BETAS=tables.nB,
            joint_names=names,
            CANONICAL_MODEL_JOINTS=list(range(tables.J)),
            TORSO_JOINTS=[i for i, n in enumerate(names) if n in _TORSO_NAMES],
        )
        for k, v in overrides.items():
            if not hasattr(cfg, k):
                raise AttributeError(f"unknown config field {k}")
            setattr(cfg, k, v)
        return cfg


current: Optional[FitterConfig] = None


def set_current(cfg: FitterConfig) -> FitterConfig:
    global current
    current = cfg
    return cfg
