"""MI355X-native OFDM channel-simulation + LS/MMSE estimation engine.

Drop-in modules (same names and call signatures as the reference's ``src/``):
``channel_simulator``, ``baseline_estimators``, ``dataset_generator``.
Batch-first engine: ``engine.Engine`` over the C ABI of ``include/chest.h`` (``libchest.so``).

The directory name is not a valid Python identifier; import it as
``importlib.import_module("channel-estimation-in-5g-network_amd")`` or through the ``chest_amd``
alias module at the repository root (``from chest_amd import engine``).
"""
__version__ = "0.1.0"

from . import _native  # noqa: F401  (ctypes signatures; the library itself is loaded on first use)
