"""Import shim: the package directory is named ``totton-rasp-gpu-dsp_amd`` (not a
valid Python identifier), so ``import totton_rasp_gpu_dsp_amd`` loads it from
that directory and registers it under this name."""
import importlib.util as _u
import sys as _sys
from pathlib import Path as _P

_dir = _P(__file__).resolve().parent / "totton-rasp-gpu-dsp_amd"
_spec = _u.spec_from_file_location(__name__, _dir / "__init__.py", submodule_search_locations=[str(_dir)])
_mod = _u.module_from_spec(_spec)
_sys.modules[__name__] = _mod
_spec.loader.exec_module(_mod)
