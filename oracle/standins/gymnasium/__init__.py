"""Minimal stand-in for gymnasium: Env seeding semantics + the few classes the reference subclasses."""
import numpy as np

from . import spaces  # noqa: F401
from . import core  # noqa: F401
from .core import Env, Wrapper, ObservationWrapper  # noqa: F401
from .envs.registration import register, make  # noqa: F401
