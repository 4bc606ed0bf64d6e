"""multigrid_amd: MI355X-native 3-D multigrid Poisson V-cycle (libmg_hip) and its
host-side bindings.  The compute path is the HIP library only; there is no CPU
fallback."""
from . import libmg, problem  # noqa: F401

__all__ = ["libmg", "problem"]
