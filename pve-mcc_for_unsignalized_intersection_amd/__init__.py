"""MI355X-native batched unsignalised-intersection environment (hot path of
Mingtzge/PVE-MCC_for_unsignalized_intersection): Python host code over hand-written HIP kernels
behind a C ABI (include/pve_env.h -> libpveenv.so)."""
from ._capi import PveError, load_library  # noqa: F401
from .batched import BatchedIntersections, PipelinedIntersections  # noqa: F401

__all__ = ["BatchedIntersections", "PipelinedIntersections", "PveError", "load_library"]
