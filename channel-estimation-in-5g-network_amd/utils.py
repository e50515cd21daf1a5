"""Drop-in replacement for the numeric helpers of the reference's ``src/utils.py``.

Names, signatures and error behaviour follow the reference.  The array helpers that sit next to the
hot path (QPSK / 16-QAM mapping, minimum-distance decisions, bit-error counting, MSE / NMSE) run as
HIP kernels through the engine's C ABI and fail loudly without it; the configuration / seeding /
dB helpers are host one-liners.  NumPy in -> NumPy out with the reference's dtypes; device tensors in ->
device tensors out.

Deviation (documented in DESIGN.md): the reference's ``qam_modulation`` indexes a Python list with an
array (src/utils.py:106) and therefore raises ``TypeError`` for every input; this module implements the
mapping that function describes, which is the exact inverse of the reference's (working)
``qam_demodulation`` -- pinned by tests/golden/qam*.npz.
"""
from __future__ import annotations

import random
from pathlib import Path
from typing import Any, Dict, Optional, Tuple

import numpy as np
import torch

from . import engine as _engine


def _eng():
    return _engine.get_engine(_engine.Geometry())


# ---- seeding / configuration (src/utils.py:13-36) ---------------------------------------------------------
def set_seed(seed: int = 42):
    """Seed python, NumPy (the legacy global stream the replay mode consumes) and torch."""
    random.seed(seed)
    np.random.seed(seed)
    torch.manual_seed(seed)
    if torch.cuda.is_available():
        torch.cuda.manual_seed_all(seed)


def load_config(config_path: str = 'configs/experiment_config.yaml') -> Dict[str, Any]:
    import yaml
    with open(config_path, 'r') as f:
        return yaml.safe_load(f)


def create_directories(config: Dict[str, Any]):
    for _, path_value in config.get('paths', {}).items():
        Path(path_value).mkdir(parents=True, exist_ok=True)


# ---- dB helpers (src/utils.py:39-46) ------------------------------------------------------------------------
def db2linear(db_value: float) -> float:
    return 10 ** (db_value / 10)


def linear2db(linear_value: float) -> float:
    return 10 * np.log10(linear_value + 1e-12)


def awgn_noise(signal_shape: Tuple, snr_db: float, signal_power: float = 1.0) -> np.ndarray:
    """Complex AWGN for a given SNR (src/utils.py:49-69).  Like the reference this consumes the global
    ``np.random`` stream: the whole real block is drawn first, then the imaginary block."""
    noise_std = np.sqrt(signal_power / db2linear(snr_db) / 2)
    noise_real = np.random.randn(*signal_shape) * noise_std
    noise_imag = np.random.randn(*signal_shape) * noise_std
    return noise_real + 1j * noise_imag


# ---- QPSK / 16-QAM (src/utils.py:71-153) ----------------------------------------------------------------------
def qam_modulation(bits, M: int = 4):
    """Gray-mapped QPSK (M=4) or 16-QAM (M=16); other orders raise NotImplementedError (:104)."""
    if M not in (4, 16):
        raise NotImplementedError(f"Modulation order {M} not implemented")
    if isinstance(bits, torch.Tensor):
        return _eng().qam_modulate(bits, M)
    return _eng().qam_modulate(np.asarray(bits) != 0, M, wide=True).cpu().numpy()


def qam_demodulation(symbols, M: int = 4):
    """Minimum-distance demodulation -> flattened bits, MSB first (:139-153).  NumPy callers get the reference's
    ``int`` array and float64 decisions on their complex128 input."""
    if M not in (4, 16):
        raise NotImplementedError(f"Demodulation order {M} not implemented")
    if isinstance(symbols, torch.Tensor):
        return _eng().qam_demodulate(symbols, M)
    s = np.asarray(symbols)
    if s.ndim != 1:
        raise IndexError("too many indices for array: symbols must be 1-dimensional")    # :141 `symbols[:, np.newaxis]`
    return _eng().qam_demodulate(s.astype(np.complex128, copy=False), M).cpu().numpy().astype(int)


def calculate_ber(transmitted_bits, received_bits):
    """Bit error rate (:156-158): differing positions / len(transmitted_bits)."""
    n = len(transmitted_bits)
    cnt = _eng().bit_errors(_as_bits(transmitted_bits), _as_bits(received_bits))
    if isinstance(transmitted_bits, torch.Tensor) and isinstance(received_bits, torch.Tensor):
        return cnt.double()[0] / n
    return float(cnt.item()) / n


def _as_bits(b):
    if isinstance(b, torch.Tensor):
        return b if b.dtype == torch.uint8 else (b != 0)
    b = np.asarray(b)
    return b if b.dtype == np.uint8 else (b != 0)


# ---- error measures (src/utils.py:161-170) -----------------------------------------------------------------------
def calculate_mse(true_channel, estimated_channel):
    sums = _eng().error_sums(true_channel, estimated_channel)
    n = int(np.prod(tuple(true_channel.shape)))
    if isinstance(true_channel, torch.Tensor):
        return sums[0] / n
    return float(sums[0].item()) / n


def calculate_nmse(true_channel, estimated_channel):
    sums = _eng().error_sums(true_channel, estimated_channel)
    n = int(np.prod(tuple(true_channel.shape)))
    if isinstance(true_channel, torch.Tensor):
        return (sums[0] / n) / (sums[1] / n + 1e-12)
    s = sums.cpu().numpy()
    return (s[0] / n) / (s[1] / n + 1e-12)


# ---- layout helpers (src/utils.py:173-180) ----------------------------------------------------------------------
def complex_to_real(complex_array):
    if isinstance(complex_array, torch.Tensor):
        return torch.view_as_real(complex_array.contiguous())
    return np.stack([complex_array.real, complex_array.imag], axis=-1)


def real_to_complex(real_array):
    if isinstance(real_array, torch.Tensor):
        return torch.view_as_complex(real_array.contiguous())
    return real_array[..., 0] + 1j * real_array[..., 1]


# ---- torch conveniences the reference's trainers import from here (src/utils.py:183-214) --------------------
def get_device(device_str: str = 'auto') -> torch.device:
    if device_str == 'auto':
        return torch.device('cuda' if torch.cuda.is_available() else 'cpu')
    return torch.device(device_str)


def save_checkpoint(model, optimizer, epoch: int, loss: float, path: str):
    torch.save({'epoch': epoch, 'model_state_dict': model.state_dict(),
                'optimizer_state_dict': optimizer.state_dict(), 'loss': loss}, path)


def load_checkpoint(model, optimizer: Optional[Any], path: str, device) -> Tuple[int, float]:
    checkpoint = torch.load(path, map_location=device, weights_only=True)
    model.load_state_dict(checkpoint['model_state_dict'])
    if optimizer is not None:
        optimizer.load_state_dict(checkpoint['optimizer_state_dict'])
    return checkpoint['epoch'], checkpoint['loss']


def count_parameters(model) -> int:
    return sum(p.numel() for p in model.parameters() if p.requires_grad)
