"""Shim with the reference's module path (``src/dataset_generator.py``): re-exports the MI355X drop-in.

Copy (or symlink) this file over the reference's ``src/dataset_generator.py`` together with the
``channel-estimation-in-5g-network_amd`` package to switch a reference checkout to the GPU engine;
both import styles the reference uses (``from src.dataset_generator import ...`` and
``sys.path.insert(0, 'src'); from dataset_generator import ...``) keep working."""
import importlib as _il
import os as _os
import sys as _sys

_root = _os.path.dirname(_os.path.dirname(_os.path.abspath(__file__)))
if _root not in _sys.path:
    _sys.path.insert(0, _root)
_m = _il.import_module("channel-estimation-in-5g-network_amd.dataset_generator")
globals().update({k: v for k, v in vars(_m).items() if not k.startswith("__")})

if __name__ == "__main__":            # the reference's file is also a script (src/dataset_generator.py:318-319)
    main()                             # noqa: F821 -- re-exported above
