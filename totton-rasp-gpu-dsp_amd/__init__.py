"""MI355X-native overlap-save FIR upsampler -- Python face of the C ABI.

Thin ctypes bindings over ``lib/libmi_upsampler.so`` (``include/mi_upsampler.h``).
The classes mirror the reference's operator interface for this path
(``include/vulkan/vulkan_streaming_upsampler.h:12-49``): ``StreamingUpsampler``
has ``load_filter / process_block / reset / config`` with the same argument
meaning and the same failure behaviour (``load_filter`` -> ``(False, message)``,
``process_block`` -> empty array). ``Filter`` + ``Engine`` are the batched form
(all channels x many blocks per call, interleaved PCM resident in HBM).

There is no CPU compute path in this package: if the shared library is missing
the import fails, and every compute entry point fails loudly when no HIP device
is usable. PyTorch is not required here; callers may pass raw device pointers
(e.g. ``tensor.data_ptr()``) and a stream handle.
"""
from __future__ import annotations

import ctypes as C
import os
from pathlib import Path

import numpy as np

_PKG = Path(__file__).resolve().parent
# MIUPS_LIB: point at another build of the same library (kernel ablation
# experiments under profiles/); never at anything else.
LIB_PATH = Path(os.environ["MIUPS_LIB"]) if os.environ.get("MIUPS_LIB") else _PKG / "lib" / "libmi_upsampler.so"

PCM_F32, PCM_S16, PCM_S24_3LE, PCM_S32 = 0, 1, 2, 3
PCM_NAMES = {"f32": PCM_F32, "s16": PCM_S16, "s24": PCM_S24_3LE, "s32": PCM_S32}
PCM_BYTES = {PCM_F32: 4, PCM_S16: 2, PCM_S24_3LE: 3, PCM_S32: 4}
LOAD_DEFAULT, LOAD_REF_COMPAT_SPECTRUM = 0, 1
MULTI_SPLIT_CHANNELS, MULTI_SPLIT_TIME = 0x10000, 0x20000
MI_OK, MI_ERR_ARG, MI_ERR_FILTER, MI_ERR_DEVICE, MI_ERR_SIZE = range(5)


class UpsamplerError(RuntimeError):
    pass


class _Negotiated(C.Structure):
    _fields_ = [("input_rate", C.c_int), ("family", C.c_int), ("output_rate", C.c_int), ("ratio", C.c_int),
                ("valid", C.c_int), ("requires_reconfiguration", C.c_int), ("error", C.c_char * 256)]


class _RuntimeConfig(C.Structure):
    _fields_ = [("eq_enabled", C.c_int), ("eq_profile", C.c_char * 256), ("eq_profile_path", C.c_char * 1024),
                ("ratio", C.c_uint), ("phase_type", C.c_char * 32), ("filter_directory", C.c_char * 1024),
                ("sample_rate", C.c_uint), ("channels", C.c_uint), ("period_frames", C.c_uint),
                ("buffer_frames", C.c_uint), ("format", C.c_char * 32), ("input_device", C.c_char * 128),
                ("output_device", C.c_char * 128)]


class _LoopParams(C.Structure):
    _fields_ = [("channels", C.c_uint), ("format", C.c_int), ("period_frames", C.c_size_t),
                ("block_in_frames", C.c_size_t), ("block_out_frames", C.c_size_t), ("max_blocks_per_call", C.c_size_t),
                ("drain_at_end", C.c_int), ("pinned_rings", C.c_int)]


class _LoopStats(C.Structure):
    _fields_ = [(n, C.c_size_t) for n in ("periods_read", "blocks_processed", "frames_written", "silence_frames_written",
                                          "input_overflows", "output_overflows", "process_calls", "in_place_calls")]


READ_FN = C.CFUNCTYPE(C.c_long, C.c_void_p, C.c_void_p, C.c_size_t)
WRITE_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.c_size_t)
PROCESS_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t)
BETWEEN_FN = C.CFUNCTYPE(None, C.c_void_p)
LOG_FN = C.CFUNCTYPE(None, C.c_void_p, C.c_char_p)


class _EqResidual(C.Structure):
    _fields_ = [("active", C.c_int), ("over_limit", C.c_int), ("tail_complete", C.c_int), ("reserved", C.c_int),
                ("tail_l1", C.c_double), ("tail_l2", C.c_double), ("tail_l1_db", C.c_double), ("tail_l2_db", C.c_double),
                ("response_dev", C.c_double), ("response_dev_db", C.c_double), ("limit", C.c_double),
                ("fir_taps", C.c_size_t), ("taper", C.c_size_t)]


def _residual(r: "_EqResidual") -> dict:
    return {n: (getattr(r, n) if t in (C.c_double,) else int(getattr(r, n))) for n, t in r._fields_ if n != "reserved"}


class _Config(C.Structure):
    _fields_ = [("taps", C.c_size_t), ("fft_size", C.c_size_t), ("block_size", C.c_size_t),
                ("upsample_factor", C.c_size_t), ("coefficients_path", C.c_char * 1024)]


def _load():
    if not LIB_PATH.exists():
        raise ImportError(
            f"{LIB_PATH} not built: run `python -c 'import __graft_entry__ as g; g.build()'` "
            "(or `make -C totton-rasp-gpu-dsp_amd`). This package has no fallback implementation.")
    lib = C.CDLL(str(LIB_PATH))
    vp, cp, sz, i32, dbl = C.c_void_p, C.c_char_p, C.c_size_t, C.c_int, C.c_double
    f32p, f64p = C.POINTER(C.c_float), C.POINTER(C.c_double)
    sig = {
        "mi_ups_abi_version": (i32, []),
        "mi_ups_device_count": (i32, []),
        "mi_ups_last_error": (cp, []),
        "mi_ups_create": (vp, [i32]),
        "mi_ups_clone": (vp, [vp]),
        "mi_ups_destroy": (None, [vp]),
        "mi_ups_load_filter": (i32, [vp, cp, i32, cp, sz]),
        "mi_ups_get_config": (i32, [vp, C.POINTER(_Config)]),
        "mi_ups_process_block": (C.c_long, [vp, f32p, sz, f32p, sz]),
        "mi_ups_reset": (i32, [vp]),
        "mi_ups_set_eq": (i32, [vp, cp, dbl]),
        "mi_engine_last_coop_frames": (i32, [vp]),
        "mi_debug_unsafe_host_copies": (C.c_ulonglong, []),
        "mi_debug_fail_host_call_at": (None, [vp, i32]),
        "mi_debug_multi_fail_host_call_at": (None, [vp, i32, i32]),
        "mi_ups_eq_residual": (i32, [vp, C.POINTER(_EqResidual)]),
        "mi_ups_set_eq_limit": (i32, [vp, dbl, i32]),
        "mi_filter_eq_residual": (i32, [vp, C.POINTER(_EqResidual)]),
        "mi_filter_set_eq_limit": (i32, [vp, dbl, i32]),
        "mi_multi_eq_residual": (i32, [vp, C.POINTER(_EqResidual)]),
        "mi_multi_set_eq_limit": (i32, [vp, dbl, i32]),
        "mi_eq_fold_host": (i32, [C.POINTER(C.c_float), sz, sz, cp, dbl, f64p, C.POINTER(_EqResidual)]),
        "mi_filter_load": (i32, [i32, cp, i32, C.POINTER(vp), cp, sz]),
        "mi_filter_from_taps": (i32, [i32, f32p, sz, sz, sz, sz, i32, C.POINTER(vp), cp, sz]),
        "mi_filter_get_config": (i32, [vp, C.POINTER(_Config)]),
        "mi_filter_set_eq": (i32, [vp, cp, dbl]),
        "mi_eq_response_device": (i32, [i32, cp, sz, sz, dbl, f64p]),
        "mi_filter_release": (None, [vp]),
        "mi_engine_create": (i32, [vp, i32, i32, i32, i32, C.POINTER(vp)]),
        "mi_engine_destroy": (None, [vp]),
        "mi_engine_reset": (i32, [vp]),
        "mi_engine_in_frames_per_block": (sz, [vp]),
        "mi_engine_out_frames_per_block": (sz, [vp]),
        "mi_engine_path": (cp, [vp]),
        "mi_engine_process_device": (i32, [vp, vp, sz, vp, sz, sz, vp]),
        "mi_engine_process_host": (i32, [vp, vp, sz, vp, sz, sz]),
        "mi_host_alloc": (vp, [sz]),
        "mi_host_free": (None, [vp]),
        "mi_host_register": (i32, [vp, sz]),
        "mi_host_unregister": (None, [vp]),
        "mi_device_copy_rate": (i32, [i32, sz, i32, f64p]),
        "mi_engine_rebind": (i32, [vp, vp, i32]),
        "mi_filter_generation": (C.c_ulonglong, [vp]),
        "mi_engine_last_generation": (C.c_ulonglong, [vp]),
        "mi_debug_fail_next_table_upload": (None, [vp]),
        "mi_engine_enable_kernel_timing": (i32, [vp, i32]),
        "mi_engine_set_kernel_timing_stride": (i32, [vp, i32]),
        "mi_engine_last_phase_parts": (i32, [vp]),
        "mi_engine_last_two_level": (i32, [vp]),
        "mi_debug_install_abort_backtrace": (None, []),
        "mi_engine_last_kernel_ms": (dbl, [vp]),
        "mi_engine_kernel_ms_stats": (i32, [vp, f64p, f64p, f64p, C.POINTER(i32)]),
        "mi_opra_to_apo": (i32, [C.c_char_p, i32, C.c_char_p, C.c_size_t, C.POINTER(C.c_size_t), C.c_char_p, C.c_size_t]),
        "mi_engine_enable_class_timing": (i32, [vp, i32]),
        "mi_engine_last_class_ms": (i32, [vp, f64p]),
        "mi_multi_create": (i32, [cp, i32, C.POINTER(i32), sz, i32, i32, i32, i32, C.POINTER(vp), cp, sz]),
        "mi_multi_destroy": (None, [vp]),
        "mi_multi_set_eq": (i32, [vp, cp, dbl]),
        "mi_multi_reset": (i32, [vp]),
        "mi_multi_process_host": (i32, [vp, vp, sz, vp, sz, sz]),
        "mi_multi_in_frames_per_block": (sz, [vp]),
        "mi_multi_out_frames_per_block": (sz, [vp]),
        "mi_multi_device_of_stream": (i32, [vp, i32]),
        "mi_multi_device_of_channel": (i32, [vp, i32]),
        "mi_multi_partition_channels": (i32, [i32, i32, C.POINTER(i32)]),
        "mi_multi_worker_cpus": (i32, [vp, i32, cp, sz]),
        "mi_debug_multi_fail_next_eq_on_slot": (None, [vp, i32]),
        "mi_multi_partition": (i32, [i32, i32, C.POINTER(i32)]),
        "mi_bank_load": (i32, [i32, cp, C.POINTER(vp), cp, sz, cp, sz]),
        "mi_bank_release": (None, [vp]),
        "mi_bank_size": (sz, [vp]),
        "mi_bank_entry": (i32, [vp, sz, C.POINTER(C.c_uint), C.POINTER(C.c_uint), cp, sz, cp, sz, C.POINTER(_Config)]),
        "mi_bank_select": (vp, [vp, C.c_uint, C.c_uint, cp, cp, sz]),
        "mi_rate_family": (i32, [i32]),
        "mi_same_family": (i32, [i32, i32]),
        "mi_upsample_ratio": (i32, [i32, i32]),
        "mi_negotiate": (i32, [i32, i32, i32, i32, C.POINTER(i32), sz, i32, C.POINTER(_Negotiated)]),
        "mi_parse_runtime_config": (i32, [cp, C.POINTER(_RuntimeConfig), cp, sz]),
        "mi_ring_create": (vp, [sz]),
        "mi_ring_destroy": (None, [vp]),
        "mi_ring_write": (i32, [vp, vp, sz]),
        "mi_ring_read": (i32, [vp, vp, sz]),
        "mi_ring_available_to_read": (sz, [vp]),
        "mi_ring_available_to_write": (sz, [vp]),
        "mi_ring_clear": (None, [vp]),
        "mi_stream_loop_run": (i32, [C.POINTER(_LoopParams), READ_FN, WRITE_FN, PROCESS_FN, BETWEEN_FN, LOG_FN, vp,
                                     C.POINTER(i32), C.POINTER(_LoopStats)]),
        "mi_read_filter": (i32, [cp, C.POINTER(_Config), cp, sz]),
        "mi_resolve_filter_path": (i32, [cp, cp, cp, C.c_uint, C.c_uint, cp, sz, cp, sz]),
        "mi_parse_format": (i32, [cp]),
        "mi_bytes_per_sample": (sz, [i32]),
        "mi_pcm_to_float": (i32, [vp, i32, sz, f32p]),
        "mi_float_to_pcm": (i32, [f32p, sz, i32, vp]),
        "mi_eq_parse": (C.c_long, [cp, f64p, f64p, sz]),
        "mi_eq_parse_filter_type": (i32, [cp]),
        "mi_eq_filter_type_name": (cp, [i32]),
        "mi_eq_biquad": (i32, [i32, i32, dbl, dbl, dbl, dbl, f64p]),
        "mi_eq_response_host": (i32, [cp, sz, sz, dbl, f64p]),
        "mi_eq_magnitude_host": (i32, [cp, sz, sz, dbl, f64p]),
        "mi_tables_build": (i32, [cp, i32, cp, dbl, C.POINTER(vp), cp, sz]),
        "mi_tables_geometry": (i32, [vp, C.POINTER(i32)]),
        "mi_tables_size": (sz, [vp, i32]),
        "mi_tables_copy": (i32, [vp, i32, f32p, sz]),
        "mi_tables_block_b": (i32, [vp, C.POINTER(i32), sz]),
        "mi_lds_swizzle": (i32, [i32]),
        "mi_fused_set_of_block": (i32, [i32, i32]),
        "mi_fused_block_a": (i32, [i32, i32]),
        "mi_fused_plan_radices": (i32, [i32, C.POINTER(i32), sz]),
        "mi_tables_free": (None, [vp]),
    }
    for name, (res, args) in sig.items():
        fn = getattr(lib, name)
        fn.restype = res
        fn.argtypes = args
    return lib


lib = _load()

EXPORTED_SYMBOLS = [
    "mi_ups_abi_version", "mi_ups_device_count", "mi_ups_last_error", "mi_ups_create", "mi_ups_clone",
    "mi_ups_destroy", "mi_ups_load_filter", "mi_ups_get_config", "mi_ups_process_block", "mi_ups_reset",
    "mi_ups_set_eq", "mi_filter_load", "mi_filter_from_taps", "mi_filter_get_config", "mi_filter_set_eq",
    "mi_eq_response_device", "mi_filter_release", "mi_engine_create", "mi_engine_destroy", "mi_engine_reset",
    "mi_engine_in_frames_per_block", "mi_engine_out_frames_per_block", "mi_engine_path", "mi_engine_process_device",
    "mi_engine_process_host", "mi_host_alloc", "mi_host_free", "mi_host_register", "mi_host_unregister",
    "mi_device_copy_rate", "mi_engine_rebind", "mi_filter_generation",
    "mi_engine_last_generation", "mi_debug_fail_next_table_upload", "mi_engine_enable_kernel_timing", "mi_engine_set_kernel_timing_stride", "mi_engine_last_phase_parts", "mi_engine_last_two_level", "mi_debug_install_abort_backtrace", "mi_engine_last_kernel_ms",
    "mi_engine_kernel_ms_stats", "mi_engine_enable_class_timing", "mi_engine_last_class_ms", "mi_opra_to_apo", "mi_multi_create", "mi_multi_destroy", "mi_multi_set_eq", "mi_multi_reset",
    "mi_multi_process_host", "mi_multi_in_frames_per_block", "mi_multi_out_frames_per_block",
    "mi_multi_device_of_stream", "mi_multi_partition", "mi_multi_device_of_channel", "mi_multi_partition_channels",
    "mi_multi_worker_cpus", "mi_debug_multi_fail_next_eq_on_slot", "mi_bank_load", "mi_bank_release", "mi_bank_size",
    "mi_bank_entry", "mi_bank_select", "mi_rate_family", "mi_same_family", "mi_upsample_ratio", "mi_negotiate",
    "mi_parse_runtime_config", "mi_ring_create", "mi_ring_destroy", "mi_ring_write", "mi_ring_read",
    "mi_ring_available_to_read", "mi_ring_available_to_write", "mi_ring_clear", "mi_stream_loop_run",
    "mi_read_filter",
    "mi_resolve_filter_path", "mi_parse_format", "mi_bytes_per_sample", "mi_pcm_to_float", "mi_float_to_pcm",
    "mi_eq_parse", "mi_eq_parse_filter_type", "mi_eq_filter_type_name", "mi_eq_biquad", "mi_eq_response_host",
    "mi_eq_magnitude_host", "mi_tables_build", "mi_tables_geometry", "mi_tables_size", "mi_tables_copy",
    "mi_tables_free", "mi_tables_block_b", "mi_lds_swizzle", "mi_fused_set_of_block", "mi_fused_block_a",
    "mi_fused_plan_radices", "mi_ups_eq_residual", "mi_ups_set_eq_limit", "mi_filter_eq_residual",
    "mi_filter_set_eq_limit", "mi_multi_eq_residual", "mi_multi_set_eq_limit", "mi_eq_fold_host",
    "mi_debug_unsafe_host_copies", "mi_debug_fail_host_call_at", "mi_debug_multi_fail_host_call_at",
    "mi_engine_last_coop_frames",
]


def unsafe_host_copies() -> int:
    """Asynchronous host copies that broke the one-in-flight-per-unpinned-page rule since the process started (must be 0)."""
    return int(lib.mi_debug_unsafe_host_copies())


def last_error() -> str:
    return lib.mi_ups_last_error().decode(errors="replace")


def device_count() -> int:
    return int(lib.mi_ups_device_count())


def _cfg(c: _Config) -> dict:
    return dict(taps=int(c.taps), fft_size=int(c.fft_size), block_size=int(c.block_size),
                upsample_factor=int(c.upsample_factor), coefficients_path=c.coefficients_path.decode())


def _f32(a):
    return a.ctypes.data_as(C.POINTER(C.c_float))


def _f64(a):
    return a.ctypes.data_as(C.POINTER(C.c_double))


# ---------------------------------------------------------------------------
# reference-shaped single-channel operator
# ---------------------------------------------------------------------------
class StreamingUpsampler:
    """One channel, host float buffers: LoadFilter / ProcessBlock / Reset /
    GetConfig of the reference class, backed by HIP."""

    def __init__(self, device: int = 0, _handle=None):
        self.device = device
        self._h = _handle if _handle is not None else lib.mi_ups_create(device)
        if not self._h:
            raise UpsamplerError("mi_ups_create failed")

    def load_filter(self, json_path, flags: int = LOAD_DEFAULT) -> tuple[bool, str]:
        err = C.create_string_buffer(1280)
        rc = lib.mi_ups_load_filter(self._h, os.fsencode(str(json_path)), flags, err, len(err))
        return rc == MI_OK, err.value.decode(errors="replace")

    @property
    def config(self) -> dict:
        c = _Config()
        lib.mi_ups_get_config(self._h, C.byref(c))
        return _cfg(c)

    def process_block(self, x, count: int | None = None) -> np.ndarray:
        """Returns block_size float32 samples, or an EMPTY array in every case
        where the reference returns an empty vector."""
        x = np.ascontiguousarray(x, dtype=np.float32)
        cap = max(self.config["block_size"], 1)
        out = np.empty(cap, dtype=np.float32)
        n = lib.mi_ups_process_block(self._h, _f32(x) if x.size else None, x.size if count is None else count,
                                     _f32(out), cap)
        return out[: max(int(n), 0)].copy()

    def reset(self) -> None:
        if lib.mi_ups_reset(self._h) != MI_OK:
            raise UpsamplerError(last_error())

    def set_eq(self, apo_text: str, fs_out: float) -> str:
        """Returns the warning text ("" when the EQ fold stayed within the limit)."""
        if lib.mi_ups_set_eq(self._h, (apo_text or "").encode(), float(fs_out)) != MI_OK:
            raise UpsamplerError(last_error())
        return last_error()

    def eq_residual(self) -> dict:
        r = _EqResidual()
        if lib.mi_ups_eq_residual(self._h, C.byref(r)) != MI_OK:
            raise UpsamplerError(last_error())
        return _residual(r)

    def set_eq_limit(self, max_tail_l1: float = -1.0, strict: bool = False) -> None:
        if lib.mi_ups_set_eq_limit(self._h, float(max_tail_l1), int(strict)) != MI_OK:
            raise UpsamplerError(last_error())

    def clone(self) -> "StreamingUpsampler":
        h = lib.mi_ups_clone(self._h)
        if not h:
            raise UpsamplerError(last_error())
        return StreamingUpsampler(self.device, _handle=h)

    def close(self):
        if getattr(self, "_h", None):
            lib.mi_ups_destroy(self._h)
            self._h = None

    __del__ = close


# ---------------------------------------------------------------------------
# batched engine
# ---------------------------------------------------------------------------
class Filter:
    def __init__(self, json_path=None, device: int = 0, flags: int = LOAD_DEFAULT, *, taps=None, fft_size=None,
                 block_size=None, upsample_factor=1):
        self.device = device
        h = C.c_void_p()
        err = C.create_string_buffer(1280)
        if json_path is not None:
            rc = lib.mi_filter_load(device, os.fsencode(str(json_path)), flags, C.byref(h), err, len(err))
        else:
            t = np.ascontiguousarray(taps, dtype=np.float32)
            rc = lib.mi_filter_from_taps(device, _f32(t), t.size, fft_size, block_size, upsample_factor, flags,
                                         C.byref(h), err, len(err))
        if rc != MI_OK:
            raise UpsamplerError(err.value.decode(errors="replace") or last_error())
        self._h = h

    @property
    def config(self) -> dict:
        c = _Config()
        lib.mi_filter_get_config(self._h, C.byref(c))
        return _cfg(c)

    def set_eq(self, apo_text: str, fs_out: float) -> str:
        """Glitch-free: engines already running keep the old spectrum for the calls they have enqueued and use the
        new one from their next call on; on failure the old spectrum stays (include/mi_upsampler.h)."""
        if lib.mi_filter_set_eq(self._h, (apo_text or "").encode(), float(fs_out)) != MI_OK:
            raise UpsamplerError(last_error())
        return last_error()  # "" or the over-the-limit warning

    def eq_residual(self) -> dict:
        """What folding the EQ into the FIR dropped (include/mi_upsampler.h mi_eq_residual)."""
        r = _EqResidual()
        if lib.mi_filter_eq_residual(self._h, C.byref(r)) != MI_OK:
            raise UpsamplerError(last_error())
        return _residual(r)

    def set_eq_limit(self, max_tail_l1: float = -1.0, strict: bool = False) -> None:
        if lib.mi_filter_set_eq_limit(self._h, float(max_tail_l1), int(strict)) != MI_OK:
            raise UpsamplerError(last_error())

    @property
    def generation(self) -> int:
        return int(lib.mi_filter_generation(self._h))

    def close(self):
        if getattr(self, "_h", None):
            lib.mi_filter_release(self._h)
            self._h = None

    __del__ = close


class Engine:
    """`streams` independent streams x `channels` interleaved channels."""

    def __init__(self, filt: Filter, streams: int = 1, channels: int = 2, in_fmt: int = PCM_S32,
                 out_fmt: int = PCM_S32):
        self.filter = filt
        self.streams, self.channels, self.in_fmt, self.out_fmt = streams, channels, in_fmt, out_fmt
        h = C.c_void_p()
        if lib.mi_engine_create(filt._h, streams, channels, in_fmt, out_fmt, C.byref(h)) != MI_OK:
            raise UpsamplerError(last_error())
        self._h = h
        self.in_frames = int(lib.mi_engine_in_frames_per_block(h))
        self.out_frames = int(lib.mi_engine_out_frames_per_block(h))
        self.path = lib.mi_engine_path(h).decode()

    def in_bytes(self, blocks: int) -> int:
        return blocks * self.in_frames * self.channels * PCM_BYTES[self.in_fmt]

    def out_bytes(self, blocks: int) -> int:
        return blocks * self.out_frames * self.channels * PCM_BYTES[self.out_fmt]

    def process_device(self, d_in: int, d_out: int, blocks: int, stream: int = 0, in_stride: int | None = None,
                       out_stride: int | None = None) -> None:
        """Enqueue only. d_in/d_out are device addresses (e.g. tensor.data_ptr())."""
        rc = lib.mi_engine_process_device(self._h, C.c_void_p(d_in), self.in_bytes(blocks) if in_stride is None else in_stride,
                                          C.c_void_p(d_out), self.out_bytes(blocks) if out_stride is None else out_stride,
                                          blocks, C.c_void_p(stream))
        if rc != MI_OK:
            raise UpsamplerError(last_error())

    def process_host(self, x: np.ndarray, blocks: int, out: np.ndarray | None = None) -> np.ndarray:
        """x: raw bytes / array laid out [stream][frame][channel]; returns uint8 bytes same layout (written into `out`
        when given, e.g. a PinnedBuffer's array)."""
        raw = np.ascontiguousarray(x).view(np.uint8).reshape(-1)
        if raw.size != self.in_bytes(blocks) * self.streams:
            raise UpsamplerError(f"input holds {raw.size} bytes, expected {self.in_bytes(blocks) * self.streams}")
        if out is None:
            out = np.empty(self.out_bytes(blocks) * self.streams, dtype=np.uint8)
        elif out.dtype != np.uint8 or out.size != self.out_bytes(blocks) * self.streams or not out.flags.c_contiguous:
            raise UpsamplerError("out must be a contiguous uint8 array of out_bytes(blocks) * streams bytes")
        rc = lib.mi_engine_process_host(self._h, raw.ctypes.data_as(C.c_void_p), self.in_bytes(blocks),
                                        out.ctypes.data_as(C.c_void_p), self.out_bytes(blocks), blocks)
        if rc != MI_OK:
            raise UpsamplerError(last_error())
        return out

    def reset(self) -> None:
        if lib.mi_engine_reset(self._h) != MI_OK:
            raise UpsamplerError(last_error())

    def rebind(self, filt: "Filter", reset_history: bool = False) -> None:
        """Switch to another resident filter at a block boundary (mi_engine_rebind)."""
        if lib.mi_engine_rebind(self._h, filt._h, int(reset_history)) != MI_OK:
            raise UpsamplerError(last_error())
        self.filter = filt
        self.in_frames = int(lib.mi_engine_in_frames_per_block(self._h))
        self.out_frames = int(lib.mi_engine_out_frames_per_block(self._h))
        self.path = lib.mi_engine_path(self._h).decode()

    @property
    def last_generation(self) -> int:
        return int(lib.mi_engine_last_generation(self._h))

    @property
    def last_phase_parts(self) -> int:
        """Workgroups per channel-block of the latest fused call (0 = not phase-split; small calls only)."""
        return int(lib.mi_engine_last_phase_parts(self._h))

    @property
    def last_two_level(self) -> bool:
        """True when the latest call of a staged engine ran the two-level transforms (K = 2^15 .. 2^18)."""
        return bool(lib.mi_engine_last_two_level(self._h))

    @property
    def last_coop_frames(self) -> bool:
        """True when the latest call's transform kernel assembled frames itself (cooperative frames)."""
        return bool(lib.mi_engine_last_coop_frames(self._h))

    def enable_kernel_timing(self, slots: int = 1, every: int = 1) -> None:
        """Ring of `slots` hipEvent pairs around the main kernel(s) of a process call; only every `every`-th call carries one."""
        if (lib.mi_engine_enable_kernel_timing(self._h, int(slots)) != MI_OK
                or lib.mi_engine_set_kernel_timing_stride(self._h, int(every)) != MI_OK):
            raise UpsamplerError(last_error())

    def kernel_ms_stats(self) -> dict:
        a, lo, hi, n = C.c_double(), C.c_double(), C.c_double(), C.c_int()
        if lib.mi_engine_kernel_ms_stats(self._h, C.byref(a), C.byref(lo), C.byref(hi), C.byref(n)) != MI_OK:
            raise UpsamplerError(last_error())
        return dict(avg=a.value, min=lo.value, max=hi.value, count=n.value)

    def last_kernel_ms(self) -> float:
        return float(lib.mi_engine_last_kernel_ms(self._h))

    def enable_class_timing(self, on: bool = True):
        """Diagnostic: every launch of the next calls gets its own event pair (perturbs the call)."""
        if lib.mi_engine_enable_class_timing(self._h, 1 if on else 0) != MI_OK:
            raise UpsamplerError(last_error())

    def last_class_ms(self) -> dict:
        """ms per kernel class of the latest call: planarize / transform / frames / history (None = no such launch)."""
        out = (C.c_double * 4)()
        if lib.mi_engine_last_class_ms(self._h, out) != MI_OK:
            raise UpsamplerError(last_error())
        names = ("planarize", "transform", "frames", "history")
        return {n: (round(float(v), 5) if v >= 0 else None) for n, v in zip(names, out)}

    def close(self):
        if getattr(self, "_h", None):
            lib.mi_engine_destroy(self._h)
            self._h = None

    __del__ = close


class MultiEngine:
    """Independent streams sharded over GPUs (stream s -> devices[s mod n]); one worker thread, filter copy and engine
    per slot, no exchange between devices (mi_multi_*)."""

    def __init__(self, json_path, devices, streams: int, channels: int, in_fmt: int = PCM_S32, out_fmt: int = PCM_S32,
                 flags: int = LOAD_DEFAULT, split_channels: bool = False, split_time: bool = False):
        """split_channels: cut every stream's channels into len(devices) contiguous groups instead of dealing whole
        streams (MI_MULTI_SPLIT_CHANNELS); split_time: cut every stream's blocks into contiguous ranges (MI_MULTI_SPLIT_TIME)."""
        if split_channels:
            flags |= MULTI_SPLIT_CHANNELS
        if split_time:
            flags |= MULTI_SPLIT_TIME
        self.devices = list(devices)
        self.streams, self.channels, self.in_fmt, self.out_fmt = streams, channels, in_fmt, out_fmt
        h = C.c_void_p()
        err = C.create_string_buffer(1280)
        dev = (C.c_int * len(self.devices))(*self.devices)
        rc = lib.mi_multi_create(os.fsencode(str(json_path)), flags, dev, len(self.devices), streams, channels, in_fmt,
                                 out_fmt, C.byref(h), err, len(err))
        if rc != MI_OK:
            raise UpsamplerError(err.value.decode(errors="replace") or last_error())
        self._h = h
        self.in_frames = int(lib.mi_multi_in_frames_per_block(h))
        self.out_frames = int(lib.mi_multi_out_frames_per_block(h))

    def in_bytes(self, blocks: int) -> int:
        return blocks * self.in_frames * self.channels * PCM_BYTES[self.in_fmt]

    def out_bytes(self, blocks: int) -> int:
        return blocks * self.out_frames * self.channels * PCM_BYTES[self.out_fmt]

    def device_of_stream(self, s: int) -> int:
        return int(lib.mi_multi_device_of_stream(self._h, s))

    def device_of_channel(self, c: int) -> int:
        return int(lib.mi_multi_device_of_channel(self._h, c))

    def worker_cpus(self, slot: int) -> str:
        out = C.create_string_buffer(1024)
        if lib.mi_multi_worker_cpus(self._h, slot, out, len(out)) != MI_OK:
            raise UpsamplerError("mi_multi_worker_cpus")
        return out.value.decode()

    def set_eq(self, apo_text: str, fs_out: float) -> str:
        if lib.mi_multi_set_eq(self._h, (apo_text or "").encode(), float(fs_out)) != MI_OK:
            raise UpsamplerError(last_error())
        return last_error()

    def eq_residual(self) -> dict:
        r = _EqResidual()
        if lib.mi_multi_eq_residual(self._h, C.byref(r)) != MI_OK:
            raise UpsamplerError(last_error())
        return _residual(r)

    def set_eq_limit(self, max_tail_l1: float = -1.0, strict: bool = False) -> None:
        if lib.mi_multi_set_eq_limit(self._h, float(max_tail_l1), int(strict)) != MI_OK:
            raise UpsamplerError(last_error())

    def reset(self) -> None:
        if lib.mi_multi_reset(self._h) != MI_OK:
            raise UpsamplerError(last_error())

    def process_host(self, x: np.ndarray, blocks: int, out: np.ndarray | None = None) -> np.ndarray:
        raw = np.ascontiguousarray(x).view(np.uint8).reshape(-1)
        if raw.size != self.in_bytes(blocks) * self.streams:
            raise UpsamplerError(f"input holds {raw.size} bytes, expected {self.in_bytes(blocks) * self.streams}")
        if out is None:
            out = np.empty(self.out_bytes(blocks) * self.streams, dtype=np.uint8)
        rc = lib.mi_multi_process_host(self._h, raw.ctypes.data_as(C.c_void_p), self.in_bytes(blocks),
                                       out.ctypes.data_as(C.c_void_p), self.out_bytes(blocks), blocks)
        if rc != MI_OK:
            raise UpsamplerError(last_error())
        return out

    def close(self):
        if getattr(self, "_h", None):
            lib.mi_multi_destroy(self._h)
            self._h = None

    __del__ = close


class FilterBank:
    """Every (rate family, ratio, phase) a filter directory serves, resident on one device (mi_bank_*)."""

    def __init__(self, filter_dir, device: int = 0):
        h = C.c_void_p()
        warn, err = C.create_string_buffer(4096), C.create_string_buffer(1280)
        if lib.mi_bank_load(device, os.fsencode(str(filter_dir)), C.byref(h), warn, len(warn), err, len(err)) != MI_OK:
            raise UpsamplerError(err.value.decode(errors="replace") or last_error())
        self._h = h
        self.device = device
        self.warnings = warn.value.decode(errors="replace")

    def entries(self) -> list[dict]:
        out = []
        for i in range(int(lib.mi_bank_size(self._h))):
            base, ratio = C.c_uint(), C.c_uint()
            phase, path, cfg = C.create_string_buffer(32), C.create_string_buffer(2048), _Config()
            lib.mi_bank_entry(self._h, i, C.byref(base), C.byref(ratio), phase, len(phase), path, len(path), C.byref(cfg))
            out.append(dict(family=base.value, ratio=ratio.value, phase=phase.value.decode(), path=os.fsdecode(path.value),
                            config=_cfg(cfg)))
        return out

    def select(self, input_rate: int, ratio: int, phase: str = "min") -> "Filter":
        err = C.create_string_buffer(1280)
        h = lib.mi_bank_select(self._h, input_rate, ratio, phase.encode(), err, len(err))
        if not h:
            raise UpsamplerError(err.value.decode(errors="replace"))
        f = Filter.__new__(Filter)
        f.device = self.device
        f._h = C.c_void_p(h)
        return f

    def close(self):
        if getattr(self, "_h", None):
            lib.mi_bank_release(self._h)
            self._h = None

    __del__ = close


def device_copy_rate(device: int = 0, nbytes: int = 1 << 30, iters: int = 5) -> float:
    """GB/s (read + write) of a plain device-to-device copy kernel on this box (mi_device_copy_rate)."""
    out = C.c_double()
    if lib.mi_device_copy_rate(device, nbytes, iters, C.byref(out)) != MI_OK:
        raise UpsamplerError(last_error())
    return float(out.value)


class RegisteredBuffer:
    """A numpy array the caller owns, page-locked in place for DMA (mi_host_register) until close()."""

    def __init__(self, array: np.ndarray):
        self.array = array
        self._p = array.ctypes.data
        if lib.mi_host_register(C.c_void_p(self._p), array.nbytes) != MI_OK:
            self._p = None
            raise UpsamplerError(last_error())

    def close(self):
        if getattr(self, "_p", None):
            lib.mi_host_unregister(C.c_void_p(self._p))
            self._p = None

    __del__ = close


class PinnedBuffer:
    """Page-locked host memory (mi_host_alloc) exposed as a numpy uint8 array: what mi_engine_process_host moves by
    DMA without a staging copy."""

    def __init__(self, nbytes: int):
        self._p = lib.mi_host_alloc(nbytes)
        if not self._p:
            raise UpsamplerError(last_error())
        self.array = np.ctypeslib.as_array((C.c_ubyte * nbytes).from_address(self._p))

    def close(self):
        if getattr(self, "_p", None):
            self.array = None
            lib.mi_host_free(self._p)
            self._p = None

    __del__ = close


# ---------------------------------------------------------------------------
# host-side helpers (no GPU)
# ---------------------------------------------------------------------------
def read_filter(json_path) -> tuple[bool, str, dict | None]:
    c = _Config()
    err = C.create_string_buffer(1280)
    rc = lib.mi_read_filter(os.fsencode(str(json_path)), C.byref(c), err, len(err))
    return rc == MI_OK, err.value.decode(errors="replace"), (_cfg(c) if rc == MI_OK else None)


def resolve_filter_path(filter_path: str, filter_dir: str, phase: str, ratio: int, input_rate: int):
    out = C.create_string_buffer(4096)
    err = C.create_string_buffer(1280)
    ok = lib.mi_resolve_filter_path(os.fsencode(filter_path), os.fsencode(filter_dir), phase.encode(), ratio,
                                    input_rate, out, len(out), err, len(err))
    return (os.fsdecode(out.value) if ok else None), err.value.decode(errors="replace")


def rate_family(rate: int) -> int:
    """0 unknown, 1 = 44.1k family, 2 = 48k family (auto_negotiation.cpp:13-31)."""
    return int(lib.mi_rate_family(rate))


def same_family(a: int, b: int) -> bool:
    return bool(lib.mi_same_family(a, b))


def upsample_ratio(input_rate: int, output_rate: int) -> int:
    return int(lib.mi_upsample_ratio(input_rate, output_rate))


def negotiate(input_rate: int, dac: dict | None, current_output_rate: int = 0) -> dict:
    """dac: dict(min=..., max=..., rates=[...]) as the reference's capability probe reports it, or None = invalid."""
    n = _Negotiated()
    rates = list((dac or {}).get("rates", []))
    arr = (C.c_int * max(len(rates), 1))(*rates)
    lib.mi_negotiate(input_rate, 1 if dac else 0, (dac or {}).get("min", 0), (dac or {}).get("max", 0), arr, len(rates),
                     current_output_rate, C.byref(n))
    return dict(input_rate=n.input_rate, family=n.family, output_rate=n.output_rate, ratio=n.ratio, valid=bool(n.valid),
                requires_reconfiguration=bool(n.requires_reconfiguration), error=n.error.decode(errors="replace"))


def parse_runtime_config(text: str) -> tuple[bool, str, dict | None]:
    c = _RuntimeConfig()
    err = C.create_string_buffer(512)
    rc = lib.mi_parse_runtime_config(text.encode(), C.byref(c), err, len(err))
    if rc != MI_OK:
        return False, err.value.decode(errors="replace"), None
    d = {n: getattr(c, n) for n, _ in _RuntimeConfig._fields_}
    d = {k: (v.decode(errors="replace") if isinstance(v, bytes) else v) for k, v in d.items()}
    d["eq_enabled"] = bool(d["eq_enabled"])
    return True, "", d


def opra_to_apo(record, modern_target: bool = False) -> str:
    """One OPRA EQ record (dict or JSON text) -> Equalizer APO text (reference: scripts/integration/opra.py
    convert_opra_to_apo(...).to_apo_format(), optionally after apply_modern_target_correction)."""
    import json as _json

    text = record if isinstance(record, str) else _json.dumps(record)
    need = C.c_size_t(0)
    err = C.create_string_buffer(512)
    out = C.create_string_buffer(4096)
    rc = lib.mi_opra_to_apo(text.encode(), 1 if modern_target else 0, out, len(out), C.byref(need), err, len(err))
    if rc == MI_ERR_ARG and need.value > len(out):
        out = C.create_string_buffer(need.value)
        rc = lib.mi_opra_to_apo(text.encode(), 1 if modern_target else 0, out, len(out), C.byref(need), err, len(err))
    if rc != MI_OK:
        raise UpsamplerError(err.value.decode(errors="replace"))
    return out.value.decode()


def multi_partition_channels(channels: int, slots: int) -> list[int]:
    out = (C.c_int * (slots + 1))()
    if lib.mi_multi_partition_channels(channels, slots, out) != MI_OK:
        raise UpsamplerError("mi_multi_partition_channels")
    return list(out)


def multi_partition(streams: int, slots: int) -> list[int]:
    out = (C.c_int * max(streams, 1))()
    if lib.mi_multi_partition(streams, slots, out) != MI_OK:
        raise UpsamplerError("mi_multi_partition")
    return list(out[:streams])


def stream_loop_run(params: dict, read, write, process=None, between=None, log=None, running=None) -> tuple[bool, dict]:
    """The streamer's loop over Python callbacks (tests): read(nframes) -> bytes, write(bytes) -> bool,
    process(in_bytes, blocks) -> out_bytes."""
    fb = PCM_BYTES[params["format"]] * params["channels"]
    lp = _LoopParams(params["channels"], params["format"], params["period_frames"], params.get("block_in_frames", 0),
                     params.get("block_out_frames", 0), params.get("max_blocks_per_call", 1),
                     int(params.get("drain_at_end", False)), int(params.get("pinned_rings", False)))

    def _read(_u, dst, frames):
        data = read(frames)
        C.memmove(dst, data, len(data))
        return len(data) // fb

    def _write(_u, src, frames):
        return 1 if write(C.string_at(src, frames * fb)) else 0

    def _process(_u, src, dst, blocks):
        out = process(C.string_at(src, blocks * params["block_in_frames"] * fb), blocks)
        if out is None or len(out) != blocks * params["block_out_frames"] * fb:
            return 0
        C.memmove(dst, out, len(out))
        return 1

    flag = running if running is not None else C.c_int(1)
    st = _LoopStats()
    rc = lib.mi_stream_loop_run(C.byref(lp), READ_FN(_read), WRITE_FN(_write),
                                PROCESS_FN(_process) if process else C.cast(None, PROCESS_FN),
                                BETWEEN_FN(lambda _u: between()) if between else C.cast(None, BETWEEN_FN),
                                LOG_FN(lambda _u, m: log(m.decode())) if log else C.cast(None, LOG_FN), None,
                                C.byref(flag), C.byref(st))
    return rc == MI_OK, {n: int(getattr(st, n)) for n, _ in _LoopStats._fields_}


def parse_format(name: str) -> int:
    return int(lib.mi_parse_format(name.encode()))


def bytes_per_sample(fmt: int) -> int:
    return int(lib.mi_bytes_per_sample(fmt))


def pcm_to_float(raw, fmt: int) -> np.ndarray:
    b = np.ascontiguousarray(np.frombuffer(raw, dtype=np.uint8) if not isinstance(raw, np.ndarray) else raw.view(np.uint8))
    n = b.size // PCM_BYTES[fmt]
    out = np.empty(n, dtype=np.float32)
    if lib.mi_pcm_to_float(b.ctypes.data_as(C.c_void_p), fmt, n, _f32(out)) != MI_OK:
        raise UpsamplerError("mi_pcm_to_float")
    return out


def float_to_pcm(x, fmt: int) -> np.ndarray:
    x = np.ascontiguousarray(x, dtype=np.float32)
    out = np.empty(x.size * PCM_BYTES[fmt], dtype=np.uint8)
    if lib.mi_float_to_pcm(_f32(x), x.size, fmt, out.ctypes.data_as(C.c_void_p)) != MI_OK:
        raise UpsamplerError("mi_float_to_pcm")
    return out


def eq_parse(text: str):
    bands = np.zeros(9 * 256)
    pre = C.c_double()
    n = lib.mi_eq_parse(text.encode(), C.byref(pre), _f64(bands), 256)
    if n < 0:
        return None
    return pre.value, bands[: 9 * n].reshape(n, 9).copy()


def eq_parse_filter_type(s: str) -> int:
    return int(lib.mi_eq_parse_filter_type(s.encode()))


def eq_filter_type_name(t: int) -> str:
    return lib.mi_eq_filter_type_name(t).decode()


def eq_biquad(enabled, type_id, freq, gain, q, fs) -> np.ndarray:
    out = np.empty(5)
    lib.mi_eq_biquad(int(enabled), int(type_id), freq, gain, q, fs, _f64(out))
    return out


def eq_response_host(text: str, num_bins: int, full_fft: int, fs_out: float) -> np.ndarray:
    out = np.empty(num_bins, dtype=np.complex128)
    lib.mi_eq_response_host(text.encode(), num_bins, full_fft, fs_out, _f64(out.view(np.float64)))
    return out


def eq_magnitude_host(text: str, num_bins: int, full_fft: int, fs_out: float) -> np.ndarray:
    out = np.empty(num_bins)
    lib.mi_eq_magnitude_host(text.encode(), num_bins, full_fft, fs_out, _f64(out))
    return out


def eq_fold_host(taps, fft_size: int, text: str, fs_out: float):
    """(fir float64[taps], residual dict): the EQ-folded FIR the table build receives (host only, no device)."""
    t = np.ascontiguousarray(taps, dtype=np.float32)
    fir = np.empty(t.size, dtype=np.float64)
    r = _EqResidual()
    if lib.mi_eq_fold_host(_f32(t), t.size, int(fft_size), text.encode(), float(fs_out), _f64(fir), C.byref(r)) != MI_OK:
        raise UpsamplerError("mi_eq_fold_host failed")
    return fir, _residual(r)


def eq_response_device(text: str, num_bins: int, full_fft: int, fs_out: float, device: int = 0) -> np.ndarray:
    out = np.empty(num_bins, dtype=np.complex128)
    if lib.mi_eq_response_device(device, text.encode(), num_bins, full_fft, fs_out, _f64(out.view(np.float64))) != MI_OK:
        raise UpsamplerError(last_error())
    return out


def fused_plan_radices(log2k: int) -> list[int]:
    out = (C.c_int * 8)()
    n = lib.mi_fused_plan_radices(log2k, out, 8)
    if n < 0:
        raise UpsamplerError("mi_fused_plan_radices")
    return list(out[:n])


def build_tables(json_path, flags: int = LOAD_DEFAULT, apo_text: str | None = None, fs_out: float = 0.0) -> dict:
    """Host-built kernel tables (for inspection/tests)."""
    h = C.c_void_p()
    err = C.create_string_buffer(1280)
    rc = lib.mi_tables_build(os.fsencode(str(json_path)), flags, apo_text.encode() if apo_text else None, fs_out,
                             C.byref(h), err, len(err))
    if rc != MI_OK:
        raise UpsamplerError(err.value.decode(errors="replace"))
    try:
        g = (C.c_int * 10)()
        lib.mi_tables_geometry(h, g)
        names = ["log2k", "K", "M", "P", "S", "Oc", "Bc", "n_in", "B", "hist_frames"]
        res = {"geometry": dict(zip(names, [int(v) for v in g]))}
        for which, name in enumerate(["Gs", "Gc", "Wm", "tw", "WmT", "GT", "G0"]):
            n = int(lib.mi_tables_size(h, which))
            a = np.empty(n, dtype=np.complex64)
            if n:
                lib.mi_tables_copy(h, which, _f32(a.view(np.float32)), n)
            res[name] = a
        nb = res["WmT"].size
        bb = (C.c_int * max(nb, 1))()
        if nb:
            lib.mi_tables_block_b(h, bb, nb)
        res["blockB"] = np.array(bb[:nb], dtype=np.int64)
        return res
    finally:
        lib.mi_tables_free(h)
