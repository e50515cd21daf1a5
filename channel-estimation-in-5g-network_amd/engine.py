"""Batch-first Python front end of the MI355X engine (libchest.so).

``Engine`` owns one C-ABI context per (device, fft, cp, S, U, T, R) plus the uploaded pilot
patterns / interpolation plans.  Tensors live in HBM as torch tensors (torch is only the allocator
and stream provider); every numeric step is a hand-written HIP kernel behind ``include/chest.h``.

Reference interfaces served (paths relative to the reference checkout):
  run_frames            one ``simulate_transmission`` (+LS +MMSE) per frame   src/channel_simulator.py:348-421,
                        src/baseline_estimators.py:83-117, :232-270, loop body src/dataset_generator.py:34-89
  estimate              LSEstimator.estimate / MMSEEstimator.estimate          src/baseline_estimators.py:83-117, :232-270
  at_pilots             *.estimate_at_pilots                                   :23-42, :155-196
  error_sums            evaluate_estimator                                     :315-337
  apply_channel         MIMOChannel.apply_channel                              src/channel_simulator.py:313-345
  time_varying_channel  ChannelModel.generate_time_varying_channel             :84-127
  ofdm_modulate/demod   OFDMSystem.modulate / demodulate                       :150-203
  equalize              equalize_channel                                       src/baseline_estimators.py:273-312
"""
from __future__ import annotations

import ctypes as C
from dataclasses import dataclass

import numpy as np
import torch

from . import _native as N
from . import interp_plan

DESC_DTYPE = np.dtype([("channel_type", "<i4"), ("doppler_hz", "<f4"), ("snr_db", "<f4"), ("pattern_slot", "<i4")])
ALL_OUTPUTS = ("tx", "rx", "H", "H_ls", "H_mmse", "pilot_symbols", "pilot_mask", "metrics")
# further outputs on request: "h_ls_pilots" (input of mmse_full_apply_pilots), "pilot_mask_f32" (the dataset schema's float32
# mask), "frame_status" (per-frame uint8: 0 ok / 1 bad descriptor / 2 no plan)


@dataclass(frozen=True)
class Geometry:
    fft_size: int = 1024
    cp_length: int = 72
    num_symbols: int = 14
    useful_subcarriers: int = 600
    subcarrier_spacing: float = 15000.0
    num_tx: int = 2
    num_rx: int = 2
    num_subcarriers: int = 0      # > 0: an explicit (num_symbols x num_subcarriers) grid without OFDM numerology (``Geometry.grid_only``)

    @classmethod
    def grid_only(cls, num_symbols: int, num_subcarriers: int, num_tx: int, num_rx: int) -> "Geometry":
        """Geometry of an explicit resource grid -- what the reference's estimator-side functions see: they take any (S, nsc), an
        even subcarrier count included (src/baseline_estimators.py:83-117, src/channel_simulator.py:313-345).  The engine of such
        a geometry (chest_create_grid) serves the estimators, apply_channel, equalize and the pattern / plan entry points; it has
        no frame pipeline, OFDM modem or channel model."""
        return cls(fft_size=0, cp_length=0, num_symbols=int(num_symbols), useful_subcarriers=0, subcarrier_spacing=0.0,
                   num_tx=int(num_tx), num_rx=int(num_rx), num_subcarriers=int(num_subcarriers))

    @property
    def is_grid_only(self) -> bool:
        return self.num_subcarriers > 0

    @property
    def num_sc(self) -> int:
        if self.num_subcarriers > 0:
            return self.num_subcarriers
        return 2 * (self.useful_subcarriers // 2) - 1

    @property
    def grid(self) -> int:
        return self.num_symbols * self.num_sc


def channel_code(channel_type) -> int:
    """'EPA'/'EVA'/'ETU' (any case) -> ABI code; unknown names raise KeyError like the reference's
    CHANNEL_PROFILES lookup (src/channel_simulator.py:67-72)."""
    if isinstance(channel_type, (int, np.integer)):
        return int(channel_type)
    return N.CHANNEL_CODES[str(channel_type).upper()]


def make_descs(channel_type, doppler_hz, snr_db, pattern_slot, n=None):
    """Host frame descriptors from scalars or per-frame sequences."""
    if n is None:
        n = max(np.size(channel_type), np.size(doppler_hz), np.size(snr_db), np.size(pattern_slot))
    d = np.empty(n, dtype=DESC_DTYPE)
    if np.ndim(channel_type) == 0:
        d["channel_type"] = channel_code(channel_type)
    else:
        d["channel_type"] = [channel_code(c) for c in channel_type]
    d["doppler_hz"] = doppler_hz
    d["snr_db"] = snr_db
    d["pattern_slot"] = pattern_slot
    return d


_NP_DTYPE = {torch.complex64: np.complex64, torch.complex128: np.complex128, torch.float32: np.float32,
             torch.float64: np.float64, torch.uint8: np.uint8, torch.int32: np.int32, torch.int64: np.int64}


class Engine:
    def __init__(self, geometry: Geometry, device=None):
        if not torch.cuda.is_available():
            raise N.ChestError("no HIP device visible: the MI355X engine has no CPU fallback")
        self.geom = geometry
        self.device = torch.device("cuda", torch.cuda.current_device() if device is None else
                                   (device if isinstance(device, int) else torch.device(device).index or 0))
        h = C.c_void_p()
        if geometry.is_grid_only:
            N.check(N.lib().chest_create_grid(geometry.num_symbols, geometry.num_subcarriers, geometry.num_tx, geometry.num_rx,
                                              self.device.index, C.byref(h)), "chest_create_grid")
        else:
            cfg = N.Config(geometry.fft_size, geometry.cp_length, geometry.num_symbols, geometry.useful_subcarriers,
                           float(geometry.subcarrier_spacing), geometry.num_tx, geometry.num_rx, self.device.index)
            N.check(N.lib().chest_create(C.byref(cfg), C.byref(h)), "chest_create")
        self._h = h
        self._profiles = {}       # channel_type code -> (delays, powers) uploaded with set_profile
        self._slots = {}          # pattern key -> slot
        self._slot_P = {}
        self._next_slot = 0
        self.max_cached_patterns = 4096
        self._stage, self._stage_i, self._stage_pending = [None, None], 0, None     # pinned descriptor staging (run_frames)
        self._workspace = None
        self._dense = {}          # pattern key -> dense interpolation operator on the device
        nsc = C.c_int32()
        N.check(N.lib().chest_get_used_indices(self._h, C.byref(nsc), None))
        assert nsc.value == geometry.num_sc

    # ---- lifetime ----------------------------------------------------------------------------
    def close(self):
        if getattr(self, "_h", None) is not None and self._h.value:
            N.lib().chest_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # ---- host-side tables -------------------------------------------------------------------------
    def used_indices(self) -> np.ndarray:
        out = np.empty(self.geom.num_sc, dtype=np.int64)
        N.check(N.lib().chest_get_used_indices(self._h, None, out.ctypes.data_as(C.POINTER(C.c_int64))))
        return out

    def tap_table(self, channel_type):
        n = C.c_int32()
        d = (C.c_int32 * N.MAX_PATHS)()
        p = (C.c_double * N.MAX_PATHS)()
        N.check(N.lib().chest_get_tap_table(self._h, channel_code(channel_type), C.byref(n), d, p), "chest_get_tap_table")
        return np.array(p[: n.value]), np.array(d[: n.value], dtype=np.int64)

    def set_profile(self, channel_type, delays_s, powers_db):
        """Replace the delay profile behind ``channel_type`` in THIS engine (chest_set_profile): an edited or added entry of the
        reference's class-level ``ChannelModel.CHANNEL_PROFILES`` (src/channel_simulator.py:41-54, :72).  At most 16 paths and 9
        distinct sample delays.  Synchronises."""
        code = channel_code(channel_type)
        d = np.ascontiguousarray(delays_s, dtype=np.float64).reshape(-1)
        p = np.ascontiguousarray(powers_db, dtype=np.float64).reshape(-1)
        if len(d) != len(p):
            raise ValueError("delays and powers differ in length")
        N.check(N.lib().chest_set_profile(self._h, code, len(d), d.ctypes.data_as(C.POINTER(C.c_double)),
                                          p.ctypes.data_as(C.POINTER(C.c_double))), "chest_set_profile")
        self._profiles[code] = (d.copy(), p.copy())

    # ---- patterns -----------------------------------------------------------------------------------
    def set_pattern(self, slot: int, pilot_idx, plan=None):
        """Upload a sorted flat pilot index list (PilotPattern.pilot_indices) and optionally its interpolation plan:
        either a host plan ``(idx uint16[3,G], w float32[3,G])`` or Delaunay simplices ``int32[n,3]`` from which the
        device builds the 'linear' plan itself."""
        pi = np.ascontiguousarray(pilot_idx, dtype=np.int32)
        if isinstance(plan, np.ndarray):
            simp = np.ascontiguousarray(plan, dtype=np.int32)
            if simp.ndim != 2 or simp.shape[1] != 3 or simp.shape[0] < 1:
                raise ValueError("simplices must be an (n, 3) integer array")
            N.check(N.lib().chest_pattern_set_triangulation(self._h, int(slot), pi.ctypes.data, len(pi), simp.ctypes.data,
                                                            simp.shape[0]), "chest_pattern_set_triangulation")
        else:
            if plan is None:
                pidx = pw = None
            else:
                pidx = np.ascontiguousarray(plan[0], dtype=np.uint16)
                pw = np.ascontiguousarray(plan[1], dtype=np.float32)
                if pidx.shape != (3, self.geom.grid) or pw.shape != (3, self.geom.grid):
                    raise ValueError(f"plan must be two (3, {self.geom.grid}) arrays")
            N.check(N.lib().chest_pattern_set(self._h, int(slot), pi.ctypes.data, len(pi),
                                              None if pidx is None else pidx.ctypes.data,
                                              None if pw is None else pw.ctypes.data), "chest_pattern_set")
        self._slot_P[int(slot)] = len(pi)
        self._next_slot = max(self._next_slot, int(slot) + 1)
        return int(slot)

    def get_plan(self, slot: int):
        """The slot's plan as host arrays (idx uint16[3,G], w float32[3,G])."""
        G = self.geom.grid
        idx = np.empty((3, G), dtype=np.uint16)
        w = np.empty((3, G), dtype=np.float32)
        N.check(N.lib().chest_pattern_get_plan(self._h, int(slot), idx.ctypes.data, w.ctypes.data), "chest_pattern_get_plan")
        return idx, w

    def pattern_slot(self, pilot_idx, method="linear", with_plan=True, plan=None):
        """Slot for a pattern, uploading it (and building its plan) on first use.  At most
        ``max_cached_patterns`` auto-managed patterns stay resident; the least recently used slot is
        re-used after that (single-frame drop-in calls draw a new random pattern every frame).  ``plan`` may carry a
        host plan or Delaunay simplices built elsewhere (interp_plan.build_plans_parallel / triangulations_parallel).
        'linear': Qhull triangulation on the host, point location + weights on the device; 'nearest': host cKDTree."""
        key = interp_plan.pattern_key(pilot_idx, method if with_plan else None)
        slot = self._slots.get(key)
        if slot is not None:
            self._slots[key] = self._slots.pop(key)           # mark most recently used
            return slot
        if with_plan and plan is None:
            S, nsc = self.geom.num_symbols, self.geom.num_sc
            ps, pk = np.unravel_index(np.asarray(pilot_idx, dtype=np.int64), (S, nsc))
            if method == "linear":
                plan = interp_plan.linear_triangulation(ps, pk)
            else:
                plan = interp_plan.build_plan(ps, pk, S, nsc, method)
        if len(self._slots) >= self.max_cached_patterns:
            old_key = next(iter(self._slots))
            slot = self._slots.pop(old_key)                   # chest_pattern_set syncs before freeing it
        else:
            slot = self._next_slot
        slot = self.set_pattern(slot, pilot_idx, plan)
        self._slots[key] = slot
        return slot

    def num_pilots(self, slot: int) -> int:
        return self._slot_P[int(slot)]

    METHOD_CODES = {None: -1, "none": -1, "linear": 0, "nearest": 1, "linear_compact": 2}

    @property
    def compact_plans_supported(self) -> bool:
        """Compact 'linear' plans (one u16 triangle id per element + one 16-B record per triangle, 67 KB instead of 151 KB per
        14 x 599 pattern) are read by the frame kernels of grids up to 640 used subcarriers; they are what a producer that
        draws a new pattern per frame should build (``build_patterns(..., method="linear_compact")``): the frame kernel then
        reads less than half the table bytes from HBM.  Only the fused frame pipeline reads them, and a context holds one plan
        format at a time."""
        return (self.geom.num_sc + 1) // 2 <= 320


    def build_patterns(self, first_slot, num_pilots, pilot_idx=None, seed=0, pattern_ids=None, method="linear",
                       return_simplices=False):
        """Build ``len(num_pilots)`` patterns and their plans ON THE DEVICE in slots ``first_slot ...`` (one launch, one
        workgroup per pattern: pilot selection, slot table, Delaunay triangulation / nearest search, plan; ``method`` "linear",
        "nearest", "linear_compact" (see ``compact_plans_supported``) or None for simulation-only patterns).  Either
        ``pattern_ids`` (uint64 per pattern: the pilots are drawn on the device from (seed, id)) or ``pilot_idx`` (device or
        host int32 [count, stride] rows of sorted flat indices).  Rebuilding the same slot range reuses its storage in
        place, ordered on the current stream.  Returns the slot numbers (and, on request, (simplices, counts) tensors)."""
        npil = np.ascontiguousarray(num_pilots, dtype=np.int32).reshape(-1)
        count = len(npil)
        ids = None if pattern_ids is None else np.ascontiguousarray(pattern_ids, dtype=np.uint64).reshape(-1)
        if ids is not None and len(ids) != count:
            raise ValueError("pattern_ids and num_pilots differ in length")
        idx_t, stride = None, 0
        if pilot_idx is not None:
            idx_t = self._dev(pilot_idx, torch.int32)
            if idx_t.dim() != 2 or idx_t.shape[0] != count:
                raise ValueError("pilot_idx must be [count, stride]")
            stride = idx_t.shape[1]
        simp = cnt = nbr = None
        if return_simplices:
            simp = torch.zeros(count, 2 * int(npil.max()), 3, dtype=torch.int32, device=self.device)
            nbr = torch.full_like(simp, -1)
            cnt = torch.zeros(count, dtype=torch.int32, device=self.device)
        N.check(N.lib().chest_patterns_build(self._h, self._stream(), int(first_slot), count, npil.ctypes.data,
                                             None if idx_t is None else idx_t.data_ptr(), int(stride), int(seed) & (2**64 - 1),
                                             None if ids is None else ids.ctypes.data, self.METHOD_CODES[method],
                                             None if simp is None else simp.data_ptr(), None if cnt is None else cnt.data_ptr(),
                                             None if nbr is None else nbr.data_ptr()),
                "chest_patterns_build")
        # the slots now hold other patterns: drop what pattern_slot() cached for them (a later hit would return a slot whose
        # pilots and plan belong to a different pattern)
        lo, hi = int(first_slot), int(first_slot) + count
        for key in [k for k, s in self._slots.items() if lo <= s < hi]:
            del self._slots[key]
        for i in range(count):
            self._slot_P[int(first_slot) + i] = int(npil[i])
        self._next_slot = max(self._next_slot, int(first_slot) + count)
        slots = np.arange(first_slot, first_slot + count, dtype=np.int32)
        if return_simplices:
            self._last_neighbors = nbr
            return slots, simp, cnt
        return slots

    def get_pilots(self, slot: int) -> np.ndarray:
        """Sorted flat pilot indices of a slot (host copy; synchronises)."""
        out = np.empty(self.num_pilots(slot), dtype=np.int32)
        N.check(N.lib().chest_pattern_get_pilots(self._h, int(slot), out.ctypes.data), "chest_pattern_get_pilots")
        return out

    def check_status(self):
        st = C.c_int32()
        N.check(N.lib().chest_get_status(self._h, C.byref(st)), "chest_get_status")
        if st.value == 1:
            raise N.ChestError("a kernel met an invalid frame descriptor (channel_type or pattern_slot)")
        if st.value == 2:
            raise N.ChestError("estimates were requested for a pattern that has no interpolation plan")
        if st.value == 3:
            raise N.ChestError("a pattern's Delaunay flips ran into their round limit: its plan is not guaranteed to be the Delaunay one")

    def profile_begin(self, capacity: int):
        """Record HIP events around the taps and frame kernels of the next `capacity` run_frames calls."""
        N.check(N.lib().chest_profile_begin(self._h, int(capacity)), "chest_profile_begin")

    def profile_read(self):
        """(taps_kernel ms, frame_kernel ms, calls averaged); synchronises and re-arms."""
        a, b, n = C.c_float(), C.c_float(), C.c_int32()
        N.check(N.lib().chest_profile_read(self._h, C.byref(a), C.byref(b), C.byref(n)), "chest_profile_read")
        return a.value, b.value, n.value

    def profile_read_all(self):
        """{kernel name: average ms per launch} for every kernel kind recorded since profile_begin; synchronises, re-arms."""
        res = {}
        for name, kid in N.KERNEL_IDS.items():
            ms, n = C.c_float(), C.c_int32()
            N.check(N.lib().chest_profile_read_kernel(self._h, kid, C.byref(ms), C.byref(n)), "chest_profile_read_kernel")
            if n.value:
                res[name] = ms.value
        self.profile_read()
        return res

    def tune(self, key: str, value: int):
        """Experiment knobs of the library (tools only): 'frame_rx_group', 'frame_min_waves', 'pattern_extra_lds',
        'pattern_flip_cap'."""
        N.check(N.lib().chest_tune(self._h, key.encode(), int(value)), "chest_tune")

    # ---- helpers ----------------------------------------------------------------------------------
    def _stream(self):
        return C.c_void_p(torch.cuda.current_stream(self.device).cuda_stream)

    def _dev(self, a, dtype):
        """Host array / tensor -> contiguous device tensor of `dtype` (no copy if already there)."""
        if isinstance(a, torch.Tensor):
            return a.to(device=self.device, dtype=dtype).contiguous()
        # convert on the host with NumPy (single-threaded), then a plain copy: a dtype-converting Tensor.to() on a CPU
        # tensor starts torch's intra-op thread pool, which on a CPU-quota'd box costs milliseconds per call
        return torch.from_numpy(np.ascontiguousarray(a, dtype=_NP_DTYPE[dtype])).to(device=self.device)

    def _c64(self, a):
        return self._dev(a, torch.complex64)

    def empty_c64(self, *shape):
        return torch.empty(shape, dtype=torch.complex64, device=self.device)

    # ---- fused frame pipeline -------------------------------------------------------------------------
    def run_frames(self, descs, seed=0, frame_offset=0, injected=None, outputs=ALL_OUTPUTS, out=None, check=True):
        """Run ``len(descs)`` frames.  ``descs``: numpy DESC_DTYPE array or a device uint8/int32 tensor
        holding the same bytes.  ``injected``: dict with any of u_sym [B,G], u_angle/u_phase [B,L,T,R,20],
        n_re/n_im [B,S,R,nsc] (float arrays; reference draw order).  Returns a dict of device tensors.
        ``out`` may carry pre-allocated tensors to reuse."""
        g = self.geom
        if isinstance(descs, torch.Tensor):
            d_dev = descs
            B = descs.numel() * descs.element_size() // DESC_DTYPE.itemsize
        else:
            descs = np.ascontiguousarray(descs, dtype=DESC_DTYPE)
            B = len(descs)
            d_dev = self._stage_descs(descs)
        S, T, R, nsc, G = g.num_symbols, g.num_tx, g.num_rx, g.num_sc, g.grid
        pmax = max(self._slot_P.values()) if self._slot_P else 0
        shapes = {"tx": ((B, S, T, nsc), torch.complex64), "rx": ((B, S, R, nsc), torch.complex64),
                  "H": ((B, S, R, T, nsc), torch.complex64), "H_ls": ((B, S, R, T, nsc), torch.complex64),
                  "H_mmse": ((B, S, R, T, nsc), torch.complex64), "pilot_symbols": ((B, pmax), torch.complex64),
                  "pilot_mask": ((B, S, nsc), torch.uint8), "metrics": ((B, N.NUM_METRICS), torch.float32),
                  "h_ls_pilots": ((B, R, (pmax + 15) // 16 * 16), torch.complex64),
                  "pilot_mask_f32": ((B, S, nsc), torch.float32), "frame_status": ((B,), torch.uint8)}
        res = {} if out is None else dict(out)
        fo = N.FrameOutputs()
        fo.pilot_stride = pmax
        fo.h_ls_stride = (pmax + 15) // 16 * 16
        for name in outputs:
            if name not in shapes:
                raise ValueError(f"unknown output {name!r}")
            shp, dt = shapes[name]
            t = res.get(name)
            if t is None:
                t = (torch.zeros if name == "pilot_symbols" else torch.empty)(shp, dtype=dt, device=self.device)
                res[name] = t
            elif tuple(t.shape) != shp or t.dtype != dt or not t.is_contiguous():
                raise ValueError(f"pre-allocated output {name!r} has the wrong shape/dtype")
            setattr(fo, name, t.data_ptr())
        keep = []
        inj_p = None
        if injected:
            inj = N.Injected()
            for name in ("u_sym", "u_angle", "u_phase", "n_re", "n_im"):
                a = injected.get(name)
                if a is None:
                    continue
                t = self._dev(a, torch.float32)
                keep.append(t)
                setattr(inj, name, t.data_ptr())
                if name == "u_angle":
                    if t.dim() != 5 or t.shape[0] != B or tuple(t.shape[2:]) != (T, R, N.NUM_OSC):
                        raise ValueError("u_angle must be [B, paths, T, R, 20]")
                    inj.inj_paths = t.shape[1]
                if name == "u_sym" and tuple(t.shape) != (B, G):
                    raise ValueError(f"u_sym must be [B, {G}]")
                if name in ("n_re", "n_im") and tuple(t.shape) != (B, S, R, nsc):
                    raise ValueError("n_re/n_im must be [B, S, R, nsc]")
            inj_p = C.byref(inj)
        need = N.lib().chest_frames_workspace_bytes(self._h, B)
        if self._workspace is None or self._workspace.numel() < need:
            self._workspace = torch.empty(need, dtype=torch.uint8, device=self.device)
        N.check(N.lib().chest_frames_run(self._h, self._stream(), B, d_dev.data_ptr(), int(seed) & (2**64 - 1),
                                         int(frame_offset), inj_p, C.byref(fo), self._workspace.data_ptr(),
                                         self._workspace.numel()), "chest_frames_run")
        if self._stage_pending is not None:
            self._stage_pending.record()          # this launch has consumed the staged descriptors
            self._stage_pending = None
        if check:
            self.check_status()       # synchronises; pass check=False inside timed loops
        return res

    def _stage_descs(self, descs):
        """Host descriptors -> device through one of two pinned staging buffers, asynchronously on the launch stream.
        (A pageable ``.to(device)`` copy is stream-ordered AND blocks the host, i.e. every call would first wait for the
        previous launch to finish.)"""
        nbytes = descs.nbytes
        self._stage_i ^= 1
        st = self._stage[self._stage_i]
        if st is None or st[0].numel() < nbytes:
            cap = max(nbytes, 1 << 16)
            st = [torch.empty(cap, dtype=torch.uint8).pin_memory(), torch.empty(cap, dtype=torch.uint8, device=self.device),
                  torch.cuda.Event()]
            st[2].record()
            self._stage[self._stage_i] = st
        pin, dev, ev = st
        ev.synchronize()                          # the launch that last read this pair has finished
        pin.numpy()[:nbytes] = descs.view(np.uint8).reshape(-1)
        dev[:nbytes].copy_(pin[:nbytes], non_blocking=True)
        self._stage_pending = ev
        return dev[:nbytes]

    # ---- output placement ---------------------------------------------------------------------------------
    PLACEMENT_GROUPS = (("tx", "rx", "H", "pilot_symbols", "pilot_mask", "pilot_mask_f32"), ("H_ls", "H_mmse"))

    def place_outputs(self, descs, outputs=ALL_OUTPUTS, seed=0, tries=4):
        """Allocate the output tensors for repeated ``run_frames(..., out=...)`` calls, choosing WHERE they live.

        On MI355X the rate at which the frame kernel's concurrent write streams drain depends on the allocation it writes into:
        six independently allocated output sets in one process read 1.81 / 1.86 / 1.87 / 2.02 / 2.03 / 2.22 ms per 10 000 2x2 frames,
        each of them the same to within 2.6 % over five round-robin visits (tools/placement_levels.py,
        profiles/r05_placement_levels.txt), while a plain fill of the same tensors runs at one rate on all of them
        (profiles/r05_fill_vs_kernel.txt): the level is a stable property of (this write pattern x the allocation).  A long-running
        job writes into the same buffers for every batch, so it pays to pick them once: this allocates ``tries`` candidate sets per
        group (synthesis tensors tx/rx/H/..., estimate tensors H_ls/H_mmse; all kept alive meanwhile, so each gets different
        memory), settles the clocks with untimed launches -- the first launches after an idle period run slower whatever they write
        into, which is what made round 4's probe unpredictive --, times the complete kernel on every combination in two round-robin
        passes (five launches back to back per visit), keeps the fastest combination and frees the rest.  Returns (tensors, report)."""
        outputs = list(outputs)
        tries = max(1, int(tries))
        groups = [[n for n in g if n in outputs] for g in self.PLACEMENT_GROUPS]
        rest = [n for n in outputs if not any(n in g for g in self.PLACEMENT_GROUPS)]
        cands = [[], []]
        for k in range(tries):                                   # candidate k of both groups, then k + 1: stop early if HBM runs out
            try:
                new = [self.run_frames(descs, seed=seed, outputs=names + (rest if gi == 0 else []), check=False)
                       if names or (gi == 0 and rest) else {} for gi, names in enumerate(groups)]
            except torch.cuda.OutOfMemoryError:
                if k == 0:
                    raise
                torch.cuda.empty_cache()
                break
            cands[0].append(new[0]); cands[1].append(new[1])
        self.check_status()
        combos = [(i, j) for i in range(len(cands[0])) for j in range(len(cands[1])) if cands[1][j] or j == 0]
        first = {**cands[0][0], **cands[1][0]}
        if len(combos) > 1:
            for _ in range(24):                                  # clocks and power state settled before anything is read
                self.run_frames(descs, seed=seed, outputs=outputs, out=first, check=False)
        samples = {c: [] for c in combos}
        nrep = 5
        for _ in range(2 if len(combos) > 1 else 1):             # round-robin: drift over time and allocation separate
            for (i, j) in combos:
                out = {**cands[0][i], **cands[1][j]}
                # launches back to back, as a job issues them (a synchronisation after every launch lets the part cool down and
                # reads 5 % faster than the sustained rate); the figure is a whole step: taps kernel + frame kernel
                self.run_frames(descs, seed=seed, outputs=outputs, out=out, check=False)
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(nrep):
                    self.run_frames(descs, seed=seed, outputs=outputs, out=out, check=False)
                e1.record()
                e1.synchronize()
                samples[(i, j)].append(e0.elapsed_time(e1) / nrep)
        timing = {c: float(np.mean(v)) for c, v in samples.items()}
        (bi, bj), best = min(timing.items(), key=lambda kv: kv[1])
        chosen = {**cands[0][bi], **cands[1][bj]}
        vals = sorted(timing.values())
        report = {"combinations": len(vals), "step_ms_best": round(vals[0], 4), "step_ms_median": round(vals[len(vals) // 2], 4),
                  "step_ms_worst": round(vals[-1], 4), "first_allocation_ms": round(timing[(0, 0)], 4),
                  "step_ms_all": [round(v, 4) for v in timing.values()],
                  "repeatability": round(max(max(v) / min(v) for v in samples.values()), 4) if len(combos) > 1 else None}
        del cands, out, first
        torch.cuda.empty_cache()                                                      # hand the losers back to the driver
        return chosen, report

    # ---- stand-alone estimator surface ---------------------------------------------------------------
    def estimate(self, rx4d, tx_pilots, slot, mmse=False, snr_db=10.0):
        g = self.geom
        rx = self._c64(rx4d)
        if rx.dim() == 4:
            rx = rx.unsqueeze(0)
        B = rx.shape[0]
        if tuple(rx.shape[1:]) != (g.num_symbols, g.num_rx, g.num_tx, g.num_sc):
            raise ValueError(f"rx_symbols must be (S,R,T,nsc)={(g.num_symbols, g.num_rx, g.num_tx, g.num_sc)}, got {tuple(rx.shape[1:])}")
        tp = self._c64(tx_pilots)
        P = self.num_pilots(slot)
        if tp.dim() == 1:
            stride = 0
            if tp.numel() != P:
                raise ValueError(f"tx_pilots has {tp.numel()} entries, the pattern has {P} pilots")
        else:
            if tuple(tp.shape) != (B, P):
                raise ValueError(f"tx_pilots must be [{B}, {P}]")
            stride = P
        out = torch.empty_like(rx)
        N.check(N.lib().chest_estimate_batch(self._h, self._stream(), B, rx.data_ptr(), tp.data_ptr(), stride,
                                             int(slot), int(bool(mmse)), float(snr_db), out.data_ptr()),
                "chest_estimate_batch")
        return out

    # ---- full-covariance MMSE (MFMA path) -----------------------------------------------------------------
    def mmse_full_prepare(self, covariance, snr_db):
        """W = R (R + sigma^2 I)^-1 with sigma^2 = 10^(-snr/10), float64 on the device; returns W [P,P] complex64."""
        cov = self._dev(covariance, torch.complex128)
        if cov.dim() != 2 or cov.shape[0] != cov.shape[1]:
            raise ValueError("channel_covariance must be a square P x P matrix")
        P = cov.shape[0]
        W = self.empty_c64(P, P)
        nbytes = N.lib().chest_mmse_full_workspace_bytes(P)
        ws = torch.empty(nbytes, dtype=torch.uint8, device=self.device)
        N.check(N.lib().chest_mmse_full_prepare(self._h, self._stream(), P, cov.data_ptr(), float(snr_db), W.data_ptr(),
                                                ws.data_ptr(), nbytes), "chest_mmse_full_prepare")
        torch.cuda.current_stream(self.device).synchronize()       # ws is freed on return
        return W

    def mmse_full_apply(self, rx4d, tx_pilots, slot, W, out=None, scratch=None):
        g = self.geom
        rx = self._c64(rx4d)
        if rx.dim() == 4:
            rx = rx.unsqueeze(0)
        B = rx.shape[0]
        if tuple(rx.shape[1:]) != (g.num_symbols, g.num_rx, g.num_tx, g.num_sc):
            raise ValueError("rx_symbols must be (S,R,T,nsc) for this engine geometry")
        tp = self._c64(tx_pilots)
        P = self.num_pilots(slot)
        if tuple(W.shape) != (P, P):
            raise ValueError(f"W must be [{P}, {P}] for this pattern")
        stride = 0 if tp.dim() == 1 else P
        if tp.numel() != (P if tp.dim() == 1 else B * P):
            raise ValueError("tx_pilots does not match the pattern's pilot count")
        M = B * g.num_rx * g.num_tx
        need = N.lib().chest_mmse_full_apply_scratch_bytes(P, M)
        own_scratch = scratch is None
        if own_scratch:
            scratch = torch.empty(need, dtype=torch.uint8, device=self.device)
        elif scratch.numel() * scratch.element_size() < need:
            raise ValueError(f"scratch holds {scratch.numel() * scratch.element_size()} B, {need} B needed")
        if out is None:
            out = torch.empty_like(rx)
        elif tuple(out.shape) != tuple(rx.shape) or out.dtype != torch.complex64 or not out.is_contiguous():
            raise ValueError("out must be a contiguous complex64 tensor shaped like rx_symbols")
        N.check(N.lib().chest_mmse_full_apply(self._h, self._stream(), B, rx.data_ptr(), tp.data_ptr(), stride, int(slot),
                                              W.contiguous().data_ptr(), scratch.data_ptr(), out.data_ptr()),
                "chest_mmse_full_apply")
        if own_scratch:
            torch.cuda.current_stream(self.device).synchronize()   # our scratch is freed on return
        return out

    def _zero_pads(self, h, P, pads_are_zero):
        """The GEMM may read rows of ``h_ls_pilots`` in place, pad columns P .. stride - 1 included (chest.h: they must be 0; a NaN
        there would poison whole output rows).  ``run_frames`` writes them as zeros; for any other tensor they are zeroed here."""
        if not pads_are_zero and h.shape[2] > P:
            h[:, :, P:].zero_()

    def mmse_full_apply_pilots(self, h_ls_pilots, slot, W, out=None, scratch=None, pads_are_zero=False):
        """Full-covariance MMSE for frames produced by ``run_frames(..., outputs=[..., "h_ls_pilots"])``: one right-hand
        side per (frame, rx) instead of per (frame, rx, tx) -- the reference's callers repeat rx over tx, so the T grids of
        an rx antenna are identical -- and every result written T times.  Returns H_mmse [B,S,R,T,nsc].  The pad entries
        ``h_ls_pilots[:, :, P:]`` are zeroed in place unless ``pads_are_zero`` vouches for them (run_frames' own output)."""
        g = self.geom
        h = self._c64(h_ls_pilots)
        P = self.num_pilots(slot)
        if h.dim() != 3 or h.shape[1] != g.num_rx or h.shape[2] < P:
            raise ValueError(f"h_ls_pilots must be [B, {g.num_rx}, >= {P}]")
        self._zero_pads(h, P, pads_are_zero)
        if W.dtype != torch.uint8 and tuple(W.shape) != (P, P):
            raise ValueError(f"W must be [{P}, {P}] for this pattern (or the result of mmse_full_pack_filter)")
        B = h.shape[0]
        need = N.lib().chest_mmse_full_apply_scratch_bytes(P, B * g.num_rx)
        own_scratch = scratch is None
        if own_scratch:
            scratch = torch.empty(need, dtype=torch.uint8, device=self.device)
        elif scratch.numel() * scratch.element_size() < need:
            raise ValueError(f"scratch holds {scratch.numel() * scratch.element_size()} B, {need} B needed")
        shape = (B, g.num_symbols, g.num_rx, g.num_tx, g.num_sc)
        if out is None:
            out = self.empty_c64(*shape)
        elif tuple(out.shape) != shape or out.dtype != torch.complex64 or not out.is_contiguous():
            raise ValueError("out must be a contiguous complex64 [B,S,R,T,nsc] tensor")
        if W.dtype == torch.uint8:                                       # packed once: the two stages, back to back
            self.mmse_full_filter_pilots(h, slot, W, scratch, pads_are_zero=True)
            self.mmse_full_interp_pilots(B, slot, scratch, out)
        else:
            N.check(N.lib().chest_mmse_full_apply_pilots(self._h, self._stream(), B, h.data_ptr(), h.shape[2], int(slot),
                                                         W.contiguous().data_ptr(), scratch.data_ptr(), out.data_ptr()),
                    "chest_mmse_full_apply_pilots")
        if own_scratch:
            torch.cuda.current_stream(self.device).synchronize()   # our scratch is freed on return
        return out

    def mmse_full_pack_filter(self, W):
        """W [P, P] complex64 -> the filter GEMM's operand format (fp16 planes), packed ONCE for every later
        ``mmse_full_apply_pilots`` / ``mmse_full_filter_pilots`` call that is handed the result as ``W`` (a uint8 tensor)."""
        Wc = self._c64(W)
        if Wc.dim() != 2 or Wc.shape[0] != Wc.shape[1]:
            raise ValueError("W must be a square [P, P] matrix")
        P = int(Wc.shape[0])
        packed = torch.empty(N.lib().chest_mmse_full_packed_filter_bytes(P), dtype=torch.uint8, device=self.device)
        N.check(N.lib().chest_mmse_full_pack_filter(self._h, self._stream(), P, Wc.contiguous().data_ptr(), packed.data_ptr()),
                "chest_mmse_full_pack_filter")
        packed._chest_filter_P = P
        return packed

    def mmse_full_filter_pilots(self, h_ls_pilots, slot, W, scratch, pads_are_zero=False):
        """Stage 1 of ``mmse_full_apply_pilots`` on the current stream: W h_ls for every (frame, rx) on the MFMA pipe; the
        filtered pilot values stay in ``scratch`` (``mmse_full_scratch(slot, B * R)``, one per batch in flight)."""
        g = self.geom
        h = self._c64(h_ls_pilots)
        P = self.num_pilots(slot)
        if h.dim() != 3 or h.shape[1] != g.num_rx or h.shape[2] < P or not h.is_contiguous():
            raise ValueError(f"h_ls_pilots must be a contiguous [B, {g.num_rx}, >= {P}] tensor")
        self._zero_pads(h, P, pads_are_zero)
        packed = W.dtype == torch.uint8                                  # mmse_full_pack_filter's result
        if packed:
            if getattr(W, "_chest_filter_P", None) != P or W.numel() != N.lib().chest_mmse_full_packed_filter_bytes(P):
                raise ValueError(f"the packed filter was not made for {P} pilots")
        elif tuple(W.shape) != (P, P) or not W.is_contiguous():
            raise ValueError(f"W must be a contiguous [{P}, {P}] tensor for this pattern")
        B = h.shape[0]
        need = N.lib().chest_mmse_full_apply_scratch_bytes(P, B * g.num_rx)
        if scratch.numel() * scratch.element_size() < need:
            raise ValueError(f"scratch holds {scratch.numel() * scratch.element_size()} B, {need} B needed")
        if packed:
            N.check(N.lib().chest_mmse_full_filter_pilots_packed(self._h, self._stream(), B, h.data_ptr(), h.shape[2], int(slot),
                                                                 W.data_ptr(), scratch.data_ptr()), "chest_mmse_full_filter_pilots_packed")
        else:
            N.check(N.lib().chest_mmse_full_filter_pilots(self._h, self._stream(), B, h.data_ptr(), h.shape[2], int(slot), W.data_ptr(),
                                                          scratch.data_ptr()), "chest_mmse_full_filter_pilots")

    def mmse_full_interp_pilots(self, num_frames, slot, scratch, out):
        """Stage 2 on the current stream: interpolation of the filtered pilot values in ``scratch`` + T-fold stores into
        ``out`` [B, S, R, T, nsc] (contiguous complex64)."""
        g = self.geom
        shape = (int(num_frames), g.num_symbols, g.num_rx, g.num_tx, g.num_sc)
        if tuple(out.shape) != shape or out.dtype != torch.complex64 or not out.is_contiguous():
            raise ValueError("out must be a contiguous complex64 [B,S,R,T,nsc] tensor")
        N.check(N.lib().chest_mmse_full_interp_pilots(self._h, self._stream(), int(num_frames), int(slot), scratch.data_ptr(),
                                                      out.data_ptr()), "chest_mmse_full_interp_pilots")
        return out

    def mmse_full_scratch(self, slot, num_grids):
        """Reusable scratch tensor for mmse_full_apply(_pilots) on ``num_grids`` right-hand sides of this pattern."""
        return torch.empty(N.lib().chest_mmse_full_apply_scratch_bytes(self.num_pilots(slot), int(num_grids)), dtype=torch.uint8,
                           device=self.device)

    def cubic_operator_from_triangulation(self, slot, simplices, neighbors):
        """Dense Clough-Tocher operator A [G, P] (float32, device) of the pattern in ``slot``, built ON THE DEVICE from a
        triangulation: ``simplices`` / ``neighbors`` int32 [n, 3] (host or device) in scipy.spatial.Delaunay's conventions
        (Qhull's to reproduce the reference, or build_patterns(..., return_simplices=True)'s own)."""
        simp = self._dev(simplices, torch.int32)
        nbr = self._dev(neighbors, torch.int32)
        if simp.dim() != 2 or simp.shape[1] != 3 or tuple(nbr.shape) != tuple(simp.shape):
            raise ValueError("simplices and neighbors must be equally shaped (n, 3) integer arrays")
        P = self.num_pilots(slot)
        A = torch.empty(self.geom.grid, P, dtype=torch.float32, device=self.device)
        nbytes = N.lib().chest_cubic_workspace_bytes(self._h, P)
        ws = torch.empty(nbytes, dtype=torch.uint8, device=self.device)
        N.check(N.lib().chest_cubic_operator(self._h, self._stream(), int(slot), simp.data_ptr(), nbr.data_ptr(), simp.shape[0],
                                             A.data_ptr(), ws.data_ptr(), nbytes), "chest_cubic_operator")
        torch.cuda.current_stream(self.device).synchronize()       # ws is freed on return
        return A

    def cubic_operator(self, pilot_idx):
        """Device copy of the dense Clough-Tocher operator of a pattern (cached, least recently used of 4 dropped): Qhull's
        triangulation from the host (scipy.spatial.Delaunay, ~3 ms; its choice among co-circular lattice points is what
        the reference's values depend on), everything after it -- gradient system, its float64 solve, the patch
        coefficients of every resource element -- on the device (chest_cubic_operator)."""
        key = interp_plan.pattern_key(pilot_idx, "cubic")
        hit = self._dense.get(key)
        if hit is None:
            S, nsc = self.geom.num_symbols, self.geom.num_sc
            ps, pk = np.unravel_index(np.asarray(pilot_idx, dtype=np.int64), (S, nsc))
            simp, nbr = interp_plan.delaunay_with_neighbors(ps, pk)
            slot = self.pattern_slot(pilot_idx, "linear", plan=simp)
            hit = self.cubic_operator_from_triangulation(slot, simp, nbr)
            if len(self._dense) >= 4:
                self._dense.pop(next(iter(self._dense)))
        else:
            self._dense.pop(key)
        self._dense[key] = hit
        return hit

    def estimate_dense(self, rx4d, tx_pilots, slot, A):
        """LS pilot division + dense interpolation operator A [G, P] (LSEstimator('cubic'))."""
        g = self.geom
        rx = self._c64(rx4d)
        if rx.dim() == 4:
            rx = rx.unsqueeze(0)
        B = rx.shape[0]
        if tuple(rx.shape[1:]) != (g.num_symbols, g.num_rx, g.num_tx, g.num_sc):
            raise ValueError("rx_symbols must be (S,R,T,nsc) for this engine geometry")
        P = self.num_pilots(slot)
        if tuple(A.shape) != (g.grid, P) or A.dtype != torch.float32:
            raise ValueError(f"A must be float32 [{g.grid}, {P}]")
        tp = self._c64(tx_pilots)
        stride = 0 if tp.dim() == 1 else P
        if tp.numel() != (P if tp.dim() == 1 else B * P):
            raise ValueError("tx_pilots does not match the pattern's pilot count")
        M = B * g.num_rx * g.num_tx
        scratch = torch.empty(N.lib().chest_estimate_dense_scratch_bytes(self._h, P, M), dtype=torch.uint8, device=self.device)
        out = torch.empty_like(rx)
        N.check(N.lib().chest_estimate_dense(self._h, self._stream(), B, rx.data_ptr(), tp.data_ptr(), stride, int(slot),
                                             A.contiguous().data_ptr(), scratch.data_ptr(), out.data_ptr()),
                "chest_estimate_dense")
        torch.cuda.current_stream(self.device).synchronize()       # scratch is freed on return
        return out

    def at_pilots(self, rx_pilots, tx_pilots, mmse=False, snr_db=10.0):
        rx = self._c64(rx_pilots).reshape(-1)
        tx = self._c64(tx_pilots).reshape(-1)
        if rx.numel() != tx.numel():
            raise ValueError("rx_pilots and tx_pilots differ in length")
        out = torch.empty_like(rx)
        N.check(N.lib().chest_at_pilots(self._h, self._stream(), rx.numel(), rx.data_ptr(), tx.data_ptr(),
                                        int(bool(mmse)), float(snr_db), out.data_ptr()), "chest_at_pilots")
        return out

    def error_sums(self, H_true, H_est):
        a = self._c64(H_true).reshape(-1)
        b = self._c64(H_est).reshape(-1)
        if a.numel() != b.numel():
            raise ValueError("H_true and H_est differ in size")
        out = torch.empty(2, dtype=torch.float64, device=self.device)
        N.check(N.lib().chest_error_sums(self._h, self._stream(), a.numel(), a.data_ptr(), b.data_ptr(), out.data_ptr()),
                "chest_error_sums")
        return out

    # ---- rest of the simulator surface ---------------------------------------------------------------
    def apply_channel(self, tx, H, snr_db, n_re=None, n_im=None, seed=0, frame_offset=0):
        g = self.geom
        txd, Hd = self._c64(tx), self._c64(H)
        if txd.dim() == 3:
            txd, Hd = txd.unsqueeze(0), Hd.unsqueeze(0)
        B = txd.shape[0]
        if tuple(txd.shape[1:]) != (g.num_symbols, g.num_tx, g.num_sc) or \
                tuple(Hd.shape[1:]) != (g.num_symbols, g.num_rx, g.num_tx, g.num_sc):
            raise ValueError("tx must be (S,T,nsc) and H (S,R,T,nsc) for this engine geometry")
        snr = self._dev(np.broadcast_to(np.asarray(snr_db, dtype=np.float32), (B,)).copy(), torch.float32)
        nr = None if n_re is None else self._dev(n_re, torch.float32)
        ni = None if n_im is None else self._dev(n_im, torch.float32)
        rx = self.empty_c64(B, g.num_symbols, g.num_rx, g.num_sc)
        N.check(N.lib().chest_apply_channel(self._h, self._stream(), B, txd.data_ptr(), Hd.data_ptr(), snr.data_ptr(),
                                            None if nr is None else nr.data_ptr(), None if ni is None else ni.data_ptr(),
                                            int(seed), int(frame_offset), rx.data_ptr()), "chest_apply_channel")
        return rx

    def time_varying_channel(self, channel_type, doppler_hz, num_samples, num_tx, num_rx, u_angle=None, u_phase=None,
                             seed=0, frame_index=0):
        code = channel_code(channel_type)
        _, d = self.tap_table(code)
        D1 = int(d.max()) + 1
        ua = None if u_angle is None else self._dev(u_angle, torch.float32)
        up = None if u_phase is None else self._dev(u_phase, torch.float32)
        if ua is not None and tuple(ua.shape) != (len(d), num_tx, num_rx, N.NUM_OSC):
            raise ValueError(f"u_angle must be [{len(d)}, {num_tx}, {num_rx}, 20]")
        out = self.empty_c64(int(num_samples), num_rx, num_tx, D1)
        N.check(N.lib().chest_time_varying_channel(self._h, self._stream(), code, float(doppler_hz), int(num_samples),
                                                   int(num_tx), int(num_rx), None if ua is None else ua.data_ptr(),
                                                   None if up is None else up.data_ptr(), int(seed), int(frame_index),
                                                   out.data_ptr()), "chest_time_varying_channel")
        return out

    def tdl_apply(self, time_tx, h, channel_type):
        """y[r][n] = sum_t sum_l h[n][r][t][d_l] x[t][n-d_l]: time_tx [T][ns], h [ns][R][T][D+1] -> [R][ns]."""
        x = self._c64(time_tx)
        hh = self._c64(h)
        if x.dim() != 2 or hh.dim() != 4 or hh.shape[0] != x.shape[1] or hh.shape[2] != x.shape[0]:
            raise ValueError("time_tx must be [T, ns] and h [ns, R, T, D+1]")
        code = channel_code(channel_type)
        _, d = self.tap_table(code)
        if hh.shape[3] != int(d.max()) + 1:
            raise ValueError(f"h has {hh.shape[3]} delay taps, profile {channel_type} needs {int(d.max()) + 1}")
        ns, R, T = int(hh.shape[0]), int(hh.shape[1]), int(hh.shape[2])
        y = self.empty_c64(R, ns)
        N.check(N.lib().chest_tdl_apply(self._h, self._stream(), code, ns, T, R, x.data_ptr(), hh.data_ptr(), y.data_ptr()),
                "chest_tdl_apply")
        return y

    def pack_ml_inputs(self, rx, H_ls, H_true, pilot_mask, normalize=True):
        """Batched prepare_ml_inputs on the GPU: (inputs [B,S,nsc,5], targets [B,S,nsc,2]) float32 device tensors."""
        g = self.geom
        rxd, ls, ht = self._c64(rx), self._c64(H_ls), self._c64(H_true)
        m = self._dev(pilot_mask, torch.uint8)
        if rxd.dim() == 3:
            rxd, ls, ht, m = rxd.unsqueeze(0), ls.unsqueeze(0), ht.unsqueeze(0), m.unsqueeze(0)
        B = rxd.shape[0]
        if tuple(rxd.shape[1:]) != (g.num_symbols, g.num_rx, g.num_sc) or \
                tuple(ls.shape) != (B, g.num_symbols, g.num_rx, g.num_tx, g.num_sc) or ls.shape != ht.shape or \
                tuple(m.shape) != (B, g.num_symbols, g.num_sc):
            raise ValueError("rx [B,S,R,nsc], H_ls/H_true [B,S,R,T,nsc], pilot_mask [B,S,nsc] expected")
        inputs = torch.empty(B, g.num_symbols, g.num_sc, 5, dtype=torch.float32, device=self.device)
        targets = torch.empty(B, g.num_symbols, g.num_sc, 2, dtype=torch.float32, device=self.device)
        N.check(N.lib().chest_pack_ml_inputs(self._h, self._stream(), B, rxd.data_ptr(), ls.data_ptr(), ht.data_ptr(),
                                             m.data_ptr(), int(bool(normalize)), inputs.data_ptr(), targets.data_ptr()),
                "chest_pack_ml_inputs")
        return inputs, targets

    # ---- training-side packing (src/train.py:22-94, run_phase4_training.py:60-112) ------------------------------------------
    ML_STATS_FIELDS = ("rx_mean", "rx_std", "H_ls_mean", "H_ls_std", "H_true_mean", "H_true_std",
                       "phase4_rx_std", "phase4_H_ls_std", "phase4_H_true_std")

    def _ml_check(self, rx, H_ls, H_true):
        g = self.geom
        rxd, ls, ht = self._c64(rx), self._c64(H_ls), self._c64(H_true)
        B = rxd.shape[0]
        if rxd.dim() != 4 or tuple(rxd.shape[1:]) != (g.num_symbols, g.num_rx, g.num_sc) or \
                tuple(ls.shape) != (B, g.num_symbols, g.num_rx, g.num_tx, g.num_sc) or ls.shape != ht.shape:
            raise ValueError("rx_symbols [N,S,R,nsc] and H_ls / H_true [N,S,R,T,nsc] expected for this engine geometry")
        return rxd, ls, ht, B

    def ml_dataset_stats(self, rx, H_ls, H_true):
        """Dataset-level normalisation statistics of both reference loaders over the first antenna pair, float64 on the
        device: a [9] tensor in ``ML_STATS_FIELDS`` order (src/train.py:41-58; run_phase4_training.py:65-76)."""
        rxd, ls, ht, B = self._ml_check(rx, H_ls, H_true)
        if B < 1:
            raise ValueError("statistics of an empty dataset")
        stats = torch.empty(9, dtype=torch.float64, device=self.device)
        nbytes = N.lib().chest_ml_stats_workspace_bytes()
        ws = torch.empty(nbytes, dtype=torch.uint8, device=self.device)
        N.check(N.lib().chest_ml_dataset_stats(self._h, self._stream(), B, rxd.data_ptr(), ls.data_ptr(), ht.data_ptr(),
                                               stats.data_ptr(), ws.data_ptr(), nbytes), "chest_ml_dataset_stats")
        torch.cuda.current_stream(self.device).synchronize()       # ws is freed on return
        return stats

    def pack_ml_training(self, rx, H_ls, H_true, pilot_mask, mode="train", stats=None):
        """``__getitem__`` of the reference's training loaders for a whole dataset in one launch: (inputs [N,5,S,nsc], targets
        [N,2,S,nsc], pilot_mask [N,S,nsc]) float32 device tensors.  ``mode``: "raw" (normalize=False), "train" ((x - mean) /
        (std + 1e-8) with dataset-level statistics, src/train.py:63-94) or "phase4" (x / std(|x|), run_phase4_training.py:81-112);
        ``stats`` = ``ml_dataset_stats`` of the dataset the statistics are to come from (computed from these tensors if None)."""
        code = {"raw": 0, "train": 1, "phase4": 2}.get(mode)
        if code is None:
            raise ValueError(f"unknown packing mode {mode!r}")
        g = self.geom
        rxd, ls, ht, B = self._ml_check(rx, H_ls, H_true)
        if isinstance(pilot_mask, torch.Tensor):
            m = pilot_mask.to(self.device).contiguous()
            if m.dtype not in (torch.float32, torch.uint8):
                m = m.to(torch.float32)
        else:
            a = np.asarray(pilot_mask)
            m = self._dev(a, torch.uint8 if a.dtype in (np.uint8, np.bool_) else torch.float32)
        if tuple(m.shape) != (B, g.num_symbols, g.num_sc):
            raise ValueError("pilot_mask must be [N,S,nsc]")
        if code and stats is None:
            stats = self.ml_dataset_stats(rxd, ls, ht)
        if stats is not None and (stats.dtype != torch.float64 or stats.numel() != 9 or stats.device != self.device):
            raise ValueError("stats must be the [9] float64 device tensor of ml_dataset_stats")
        inputs = torch.empty(B, 5, g.num_symbols, g.num_sc, dtype=torch.float32, device=self.device)
        targets = torch.empty(B, 2, g.num_symbols, g.num_sc, dtype=torch.float32, device=self.device)
        mask_out = torch.empty(B, g.num_symbols, g.num_sc, dtype=torch.float32, device=self.device)
        N.check(N.lib().chest_pack_ml_training(self._h, self._stream(), B, rxd.data_ptr(), ls.data_ptr(), ht.data_ptr(),
                                               m.data_ptr(), int(m.dtype == torch.float32), code,
                                               None if stats is None else stats.data_ptr(), inputs.data_ptr(), targets.data_ptr(),
                                               mask_out.data_ptr()), "chest_pack_ml_training")
        return inputs, targets, mask_out

    def ofdm_modulate(self, symbols):
        g = self.geom
        s = self._c64(symbols)
        if s.dim() != 2 or s.shape[1] != g.num_sc:
            raise ValueError(f"symbols must be (n, {g.num_sc})")
        out = self.empty_c64(s.shape[0], g.fft_size + g.cp_length)
        N.check(N.lib().chest_ofdm_modulate(self._h, self._stream(), s.shape[0], s.data_ptr(), out.data_ptr()),
                "chest_ofdm_modulate")
        return out

    def ofdm_demodulate(self, received):
        g = self.geom
        r = self._c64(received)
        if r.dim() != 2 or r.shape[1] != g.fft_size + g.cp_length:
            raise ValueError(f"received_signal must be (n, {g.fft_size + g.cp_length})")
        out = self.empty_c64(r.shape[0], g.num_sc)
        N.check(N.lib().chest_ofdm_demodulate(self._h, self._stream(), r.shape[0], r.data_ptr(), out.data_ptr()),
                "chest_ofdm_demodulate")
        return out

    def equalize(self, rx, H_est, method="zf"):
        g = self.geom
        if method not in ("zf", "mmse"):
            raise ValueError(f"Unknown equalization method: {method}")
        r, H = self._c64(rx), self._c64(H_est)
        if r.dim() == 3:
            r, H = r.unsqueeze(0), H.unsqueeze(0)
        B = r.shape[0]
        out = self.empty_c64(B, g.num_symbols, g.num_tx, g.num_sc)
        N.check(N.lib().chest_equalize(self._h, self._stream(), B, r.data_ptr(), H.data_ptr(),
                                       0 if method == "zf" else 1, out.data_ptr()), "chest_equalize")
        return out

    # ---- QPSK / 16-QAM / BER helpers (src/utils.py:71-158) -----------------------------------------------
    def qam_modulate(self, bits, M=4, wide=False):
        """bits (0/1, one element per bit, MSB of a symbol first) -> constellation points; trailing bits that do
        not fill a symbol are dropped (:84-87).  Device tensor complex64 (or complex128 with ``wide``)."""
        bps = {4: 2, 16: 4}.get(int(M))
        if bps is None:
            raise NotImplementedError(f"Modulation order {M} not implemented")
        b = self._dev(bits, torch.uint8).reshape(-1)
        n = b.numel() // bps
        out = torch.empty(n, dtype=torch.complex128 if wide else torch.complex64, device=self.device)
        N.check(N.lib().chest_qam_modulate(self._h, self._stream(), int(M), n, b.data_ptr(), int(bool(wide)), out.data_ptr()),
                "chest_qam_modulate")
        return out

    def qam_demodulate(self, symbols, M=4):
        """Minimum-distance decisions -> uint8 device tensor of bits, MSB first, flattened (:139-153).  complex128
        input is decided in float64 as given; anything else is taken as complex64."""
        bps = {4: 2, 16: 4}.get(int(M))
        if bps is None:
            raise NotImplementedError(f"Demodulation order {M} not implemented")
        wide = (symbols.dtype == torch.complex128) if isinstance(symbols, torch.Tensor) else \
            (np.asarray(symbols).dtype == np.complex128)
        s = self._dev(symbols, torch.complex128 if wide else torch.complex64)
        if s.dim() != 1:
            raise ValueError("symbols must be one-dimensional")
        bits = torch.empty(s.numel() * bps, dtype=torch.uint8, device=self.device)
        N.check(N.lib().chest_qam_demodulate(self._h, self._stream(), int(M), s.numel(), int(wide), s.data_ptr(),
                                             bits.data_ptr()), "chest_qam_demodulate")
        return bits

    def bit_errors(self, a, b):
        """Number of differing positions of two equally long bit strings (device uint64 scalar as int64 tensor)."""
        x, y = self._dev(a, torch.uint8).reshape(-1), self._dev(b, torch.uint8).reshape(-1)
        if x.numel() != y.numel():
            raise ValueError("bit strings differ in length")
        cnt = torch.empty(1, dtype=torch.int64, device=self.device)
        N.check(N.lib().chest_bit_errors(self._h, self._stream(), x.numel(), x.data_ptr(), y.data_ptr(), cnt.data_ptr()),
                "chest_bit_errors")
        return cnt



_ENGINES = {}


def get_engine(geometry: Geometry, device=None, tag: str = "") -> Engine:
    """Process-wide engine cache keyed by geometry, device and ``tag`` (a context holds interpolation plans of one format, so
    the producers of compact per-frame plans keep an engine of their own: tag "compact")."""
    dev = torch.cuda.current_device() if device is None else device
    key = (geometry, dev, tag)
    e = _ENGINES.get(key)
    if e is None:
        e = Engine(geometry, dev)
        _ENGINES[key] = e
    return e
