"""Importable alias of the package directory `pve-mcc_for_unsignalized_intersection_amd/`
(its name is not a valid Python identifier): `import pve_mcc_amd` executes that package's
__init__ under this module name, so `from pve_mcc_amd.batched import BatchedIntersections` works."""
import os as _os

__path__ = [_os.path.join(_os.path.dirname(_os.path.abspath(__file__)), "pve-mcc_for_unsignalized_intersection_amd")]
__package__ = "pve_mcc_amd"
_init = _os.path.join(__path__[0], "__init__.py")
with open(_init) as _f:
    exec(compile(_f.read(), _init, "exec"), globals())
