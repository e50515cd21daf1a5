"""ctypes binding of the C ABI in include/chest.h (libchest.so, built by csrc/Makefile).

The product path has no CPU fallback: if the library is missing or a call fails, this module raises.
"""
from __future__ import annotations

import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("CHEST_LIB") or os.path.join(_HERE, "libchest.so")   # CHEST_LIB: A/B builds only

MAX_PATHS = 16
NUM_OSC = 20
NUM_METRICS = 8
M_SUM_H2, M_SUM_ERR_LS, M_SUM_ERR_MMSE, M_SIGNAL_POWER, M_NOISE_STD, M_ALPHA0, M_COUNT = range(7)
ABI_VERSION = 3
FRAME_OK, FRAME_BAD_DESC, FRAME_NO_PLAN = 0, 1, 2
KERNEL_IDS = {"taps_kernel": 0, "frame_kernel": 1, "pack_kernels": 2, "cgemm_w_kernel": 3, "interp_from_pilots_kernel": 4,
              "plan_kernels": 5}
CHANNEL_CODES = {"EPA": 0, "EVA": 1, "ETU": 2}


class ChestError(RuntimeError):
    """A C-ABI call returned a negative status."""


class Config(C.Structure):
    _fields_ = [("fft_size", C.c_int32), ("cp_length", C.c_int32), ("num_symbols", C.c_int32),
                ("useful_subcarriers", C.c_int32), ("subcarrier_spacing", C.c_double),
                ("num_tx", C.c_int32), ("num_rx", C.c_int32), ("device", C.c_int32)]


class FrameDesc(C.Structure):
    _fields_ = [("channel_type", C.c_int32), ("doppler_hz", C.c_float), ("snr_db", C.c_float),
                ("pattern_slot", C.c_int32)]


class Injected(C.Structure):
    _fields_ = [("u_sym", C.c_void_p), ("u_angle", C.c_void_p), ("u_phase", C.c_void_p),
                ("n_re", C.c_void_p), ("n_im", C.c_void_p), ("inj_paths", C.c_int32)]


class FrameOutputs(C.Structure):
    _fields_ = [("tx", C.c_void_p), ("rx", C.c_void_p), ("H", C.c_void_p), ("H_ls", C.c_void_p),
                ("H_mmse", C.c_void_p), ("pilot_symbols", C.c_void_p), ("pilot_mask", C.c_void_p),
                ("metrics", C.c_void_p), ("pilot_stride", C.c_int32), ("h_ls_stride", C.c_int32),
                ("h_ls_pilots", C.c_void_p), ("pilot_mask_f32", C.c_void_p), ("frame_status", C.c_void_p)]


# name -> (restype, argtypes); this table is also what tests/test_abi_symbols.py checks against chest.h
_vp, _i32, _i64, _u64, _f = C.c_void_p, C.c_int32, C.c_int64, C.c_uint64, C.c_float
SIGNATURES = {
    "chest_create": (_i32, [C.POINTER(Config), C.POINTER(_vp)]),
    "chest_create_grid": (_i32, [_i32, _i32, _i32, _i32, _i32, C.POINTER(_vp)]),
    "chest_set_profile": (_i32, [_vp, _i32, _i32, C.POINTER(C.c_double), C.POINTER(C.c_double)]),
    "chest_destroy": (None, [_vp]),
    "chest_last_error": (C.c_char_p, []),
    "chest_abi_version": (_i32, []),
    "chest_get_used_indices": (_i32, [_vp, C.POINTER(_i32), C.POINTER(_i64)]),
    "chest_get_tap_table": (_i32, [_vp, _i32, C.POINTER(_i32), C.POINTER(_i32), C.POINTER(C.c_double)]),
    "chest_pattern_set": (_i32, [_vp, _i32, _vp, _i32, _vp, _vp]),
    "chest_pattern_set_triangulation": (_i32, [_vp, _i32, _vp, _i32, _vp, _i32]),
    "chest_patterns_build": (_i32, [_vp, _vp, _i32, _i32, _vp, _vp, _i32, _u64, _vp, _i32, _vp, _vp, _vp]),
    "chest_cubic_workspace_bytes": (_i64, [_vp, _i32]),
    "chest_cubic_operator": (_i32, [_vp, _vp, _i32, _vp, _vp, _i32, _vp, _vp, _i64]),
    "chest_pattern_get_pilots": (_i32, [_vp, _i32, _vp]),
    "chest_pattern_get_plan": (_i32, [_vp, _i32, _vp, _vp]),
    "chest_pattern_num_pilots": (_i32, [_vp, _i32]),
    "chest_get_status": (_i32, [_vp, C.POINTER(_i32)]),
    "chest_profile_begin": (_i32, [_vp, _i32]),
    "chest_profile_read": (_i32, [_vp, C.POINTER(C.c_float), C.POINTER(C.c_float), C.POINTER(_i32)]),
    "chest_profile_read_kernel": (_i32, [_vp, _i32, C.POINTER(C.c_float), C.POINTER(_i32)]),
    "chest_tune": (_i32, [_vp, C.c_char_p, _i32]),
    "chest_frames_workspace_bytes": (_i64, [_vp, _i32]),
    "chest_frames_run": (_i32, [_vp, _vp, _i32, _vp, _u64, _u64, C.POINTER(Injected), C.POINTER(FrameOutputs), _vp, _i64]),
    "chest_estimate_batch": (_i32, [_vp, _vp, _i32, _vp, _vp, _i32, _i32, _i32, _f, _vp]),
    "chest_at_pilots": (_i32, [_vp, _vp, _i32, _vp, _vp, _i32, _f, _vp]),
    "chest_error_sums": (_i32, [_vp, _vp, _i64, _vp, _vp, _vp]),
    "chest_apply_channel": (_i32, [_vp, _vp, _i32, _vp, _vp, _vp, _vp, _vp, _u64, _u64, _vp]),
    "chest_time_varying_channel": (_i32, [_vp, _vp, _i32, _f, _i64, _i32, _i32, _vp, _vp, _u64, _u64, _vp]),
    "chest_ofdm_modulate": (_i32, [_vp, _vp, _i32, _vp, _vp]),
    "chest_ofdm_demodulate": (_i32, [_vp, _vp, _i32, _vp, _vp]),
    "chest_equalize": (_i32, [_vp, _vp, _i32, _vp, _vp, _i32, _vp]),
    "chest_tdl_apply": (_i32, [_vp, _vp, _i32, _i64, _i32, _i32, _vp, _vp, _vp]),
    "chest_pack_ml_inputs": (_i32, [_vp, _vp, _i32, _vp, _vp, _vp, _vp, _i32, _vp, _vp]),
    "chest_ml_stats_workspace_bytes": (_i64, []),
    "chest_ml_dataset_stats": (_i32, [_vp, _vp, _i32, _vp, _vp, _vp, _vp, _vp, _i64]),
    "chest_pack_ml_training": (_i32, [_vp, _vp, _i32, _vp, _vp, _vp, _vp, _i32, _i32, _vp, _vp, _vp, _vp]),
    "chest_qam_constellation": (_i32, [_i32, _vp, _vp]),
    "chest_qam_modulate": (_i32, [_vp, _vp, _i32, _i64, _vp, _i32, _vp]),
    "chest_qam_demodulate": (_i32, [_vp, _vp, _i32, _i64, _i32, _vp, _vp]),
    "chest_bit_errors": (_i32, [_vp, _vp, _i64, _vp, _vp, _vp]),
    "chest_mmse_full_prepare": (_i32, [_vp, _vp, _i32, _vp, _f, _vp, _vp, _i64]),
    "chest_mmse_full_workspace_bytes": (_i64, [_i32]),
    "chest_mmse_full_apply_scratch_bytes": (_i64, [_i32, _i32]),
    "chest_mmse_full_apply_pilots": (_i32, [_vp, _vp, _i32, _vp, _i32, _i32, _vp, _vp, _vp]),
    "chest_mmse_full_filter_pilots": (_i32, [_vp, _vp, _i32, _vp, _i32, _i32, _vp, _vp]),
    "chest_mmse_full_packed_filter_bytes": (_i64, [_i32]),
    "chest_mmse_full_pack_filter": (_i32, [_vp, _vp, _i32, _vp, _vp]),
    "chest_mmse_full_filter_pilots_packed": (_i32, [_vp, _vp, _i32, _vp, _i32, _i32, _vp, _vp]),
    "chest_mmse_full_interp_pilots": (_i32, [_vp, _vp, _i32, _i32, _vp, _vp]),
    "chest_estimate_dense_scratch_bytes": (_i64, [_vp, _i32, _i32]),
    "chest_estimate_dense": (_i32, [_vp, _vp, _i32, _vp, _vp, _i32, _i32, _vp, _vp, _vp]),
    "chest_mmse_full_apply": (_i32, [_vp, _vp, _i32, _vp, _vp, _i32, _i32, _vp, _vp, _vp]),
}

_lib = None


def lib():
    """Load libchest.so once; raise (never fall back) if it is not built."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise ChestError(
                f"{LIB_PATH} is missing: build the HIP extension first "
                "(python -c 'import __graft_entry__ as g; g.build()' or make -C <package>/csrc)")
        # Load torch's HIP runtime BEFORE libchest.so: the PyTorch wheel ships its own libamdhip64, and whichever copy is
        # mapped first serves the whole process.  If libchest.so pulled in /opt/rocm's copy first, a later `import torch`
        # would leave two runtimes behind and chest_create would fail with "no ROCm-capable device is detected".
        import torch  # noqa: F401
        L = C.CDLL(LIB_PATH)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(L, name)
            fn.restype = res
            fn.argtypes = args
        _lib = L
    return _lib


def check(rc, what=""):
    if rc is not None and rc < 0:
        msg = lib().chest_last_error().decode("utf-8", "replace")
        raise ChestError(f"{what or 'chest call'} failed ({rc}): {msg}")
    return rc
