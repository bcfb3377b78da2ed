"""Import alias: the package directory is ``mlperf-deepcam_amd/`` (not a valid Python identifier),
so this shim package extends its own search path to that directory and re-exports its namespace."""
import os as _os

_real = _os.path.join(_os.path.dirname(_os.path.dirname(_os.path.abspath(__file__))), "mlperf-deepcam_amd")
__path__.append(_real)
with open(_os.path.join(_real, "__init__.py")) as _f:
    exec(compile(_f.read(), _os.path.join(_real, "__init__.py"), "exec"))
