"""Shim with the reference's module path (``src/utils.py``): re-exports the MI355X drop-in.

Copy (or symlink) this file over the reference's ``src/utils.py`` together with the
``channel-estimation-in-5g-network_amd`` package to switch a reference checkout to the GPU engine;
both import styles the reference uses (``from src.utils import ...`` and
``sys.path.insert(0, 'src'); from utils import ...``) keep working."""
import importlib as _il
import os as _os
import sys as _sys

_root = _os.path.dirname(_os.path.dirname(_os.path.abspath(__file__)))
if _root not in _sys.path:
    _sys.path.insert(0, _root)
_m = _il.import_module("channel-estimation-in-5g-network_amd.utils")
globals().update({k: v for k, v in vars(_m).items() if not k.startswith("__")})
