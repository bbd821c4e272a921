"""Import shim: the package directory is ``f5e-tts_amd/`` (not a valid Python identifier), so ``import f5e_tts_amd``
lands here and registers that directory as the package ``f5e_tts_amd``."""
import importlib.util
import os
import sys

_dir = os.path.join(os.path.dirname(os.path.abspath(__file__)), "f5e-tts_amd")
_spec = importlib.util.spec_from_file_location(
    "f5e_tts_amd", os.path.join(_dir, "__init__.py"), submodule_search_locations=[_dir]
)
_mod = importlib.util.module_from_spec(_spec)
sys.modules["f5e_tts_amd"] = _mod
_spec.loader.exec_module(_mod)
