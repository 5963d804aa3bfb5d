"""benchpush_amd: MI355X-native batched 2D pushing-physics environments (ship-ice hot path of IvanIZ/BenchPush)."""
from . import envs  # noqa: F401  (registers the gym ids)
from .config import DotDict  # noqa: F401
from .gym_shim import make, register  # noqa: F401

__version__ = "0.1.0"
