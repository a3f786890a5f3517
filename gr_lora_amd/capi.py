"""ctypes binding of the C ABI declared in include/lora_hip.h (liblora_hip.so).

The library is loaded on first use; if it is missing or no MI355X is visible the
calls raise -- there is no CPU fallback anywhere in this package.
"""
from __future__ import annotations

import ctypes as C
import os
from typing import List, Optional, Sequence, Tuple

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("LORA_HIP_LIB") or os.path.join(_HERE, "liblora_hip.so")  # (LORA_HIP_LIB: a library variant, tools/ab.sh)

DEMOD_GRAD, DEMOD_FFT, DEMOD_FFT_COMPAT = 0, 1, 2
FLAG_TRACE = 1
FLAG_PIN_HOST = 2
FLAG_FAST_SYNC = 4  # keep SYNC's closed-form maximum (no exact re-evaluation of near-tied shifts)
FLAG_NO_DECOUPLED = 8  # never run a pass decoupled (header-only jobs + the symbol-parallel payload pass)

EXPORTS = [
    "lora_hip_abi_version", "lora_hip_strerror", "lora_hip_last_error", "lora_hip_create", "lora_hip_destroy",
    "lora_hip_get_geometry", "lora_hip_set_sf", "lora_hip_set_samp_rate", "lora_hip_work", "lora_hip_flush",
    "lora_hip_decode_device", "lora_hip_frames_available", "lora_hip_poll_frame", "lora_hip_drain_frames", "lora_hip_drain_slots", "lora_hip_demod_symbols_device", "lora_hip_demod_symbols_ex_device",
    "lora_hip_last_timing", "lora_hip_last_plan", "lora_hip_get_table", "lora_hip_last_payload_pass", "lora_hip_gap_starts_device", "lora_hip_decode_device_begin", "lora_hip_decode_device_end", "lora_hip_decode_device_prepass", "lora_hip_trace", "lora_hip_trace_clear", "lora_hip_check_frame", "lora_hip_estimate_cfo_device", "lora_hip_ref_ifreq_device",
    "lora_hip_set_stream_latency", "lora_hip_stream_info", "lora_hip_walker_kernel_name", "lora_hip_window_stats_device", "lora_hip_detect_preambles_device", "lora_hip_decode_at_headers_device",
    "lora_hip_mux_create", "lora_hip_mux_destroy", "lora_hip_mux_work", "lora_hip_mux_flush", "lora_hip_mux_set_latency", "lora_hip_mux_set_max_ahead", "lora_hip_mux_frames_available",
    "lora_hip_mux_poll_frame", "lora_hip_mux_passes", "lora_hip_mux_last_error",
]


EXPORTS_CHANNELIZER = [
    "lora_hip_channelizer_create", "lora_hip_channelizer_destroy", "lora_hip_channelizer_last_error", "lora_hip_channelizer_taps",
    "lora_hip_channelizer_output_items", "lora_hip_channelizer_run_device", "lora_hip_channelizer_work", "lora_hip_channelizer_apply_cfo",
    "lora_hip_channelizer_last_kernel_ms",
]


class ChannelizerConfig(C.Structure):
    _fields_ = [("struct_size", C.c_uint32), ("samp_rate", C.c_float), ("center_freq", C.c_float), ("channel_list", C.POINTER(C.c_float)),
                ("n_channels", C.c_uint32), ("bandwidth", C.c_uint32), ("decimation", C.c_uint32), ("device", C.c_int32),
                ("cutoff_hz", C.c_float), ("transition_hz", C.c_float), ("flags", C.c_uint32)]


CHANNELIZER_FLAG_UINT32_OFFSET = 1  # d_freq_offset in upstream's uint32_t arithmetic (a negative offset wraps), include/lora_hip_channelizer.h


class Config(C.Structure):
    _fields_ = [("struct_size", C.c_uint32), ("samp_rate", C.c_float), ("bandwidth", C.c_uint32),
                ("sf", C.c_uint8), ("implicit", C.c_uint8), ("cr", C.c_uint8), ("crc", C.c_uint8),
                ("reduced_rate", C.c_uint8), ("disable_drift_correction", C.c_uint8), ("reserved0", C.c_uint8 * 2),
                ("device", C.c_int32), ("demod", C.c_int32), ("flags", C.c_uint32),
                ("segment_symbols", C.c_uint32), ("batch_items", C.c_uint32)]


class FrameInfo(C.Structure):
    _fields_ = [("stream", C.c_uint32), ("length", C.c_uint32), ("header_pos", C.c_int64), ("end_pos", C.c_int64)]


class Step(C.Structure):
    _fields_ = [("state", C.c_int32), ("consumed", C.c_int32), ("pos", C.c_int64), ("bin", C.c_int32),
                ("fine", C.c_int32), ("value", C.c_float), ("stream", C.c_uint32), ("cycles", C.c_uint32), ("reserved", C.c_uint32)]


class Timing(C.Structure):
    _fields_ = [("walker_ms", C.c_float), ("total_device_ms", C.c_float), ("walker_launches", C.c_uint32),
                ("jobs", C.c_uint32), ("probes", C.c_uint32), ("slow_path_relaunches", C.c_uint32),
                ("items", C.c_uint64)]


class FrameCheck(C.Structure):
    _fields_ = [("has_header", C.c_uint8), ("header_checksum_ok", C.c_uint8), ("has_crc", C.c_uint8), ("crc_ok", C.c_uint8),
                ("header_checksum_rx", C.c_uint8), ("header_checksum_calc", C.c_uint8), ("crc_rx", C.c_uint16), ("crc_calc", C.c_uint16),
                ("reserved", C.c_uint16)]


class StreamInfo(C.Structure):
    _fields_ = [("batch_items", C.c_uint64), ("buffered_items", C.c_uint64), ("passes", C.c_uint64), ("passes_by_latency", C.c_uint64),
                ("consumed_base", C.c_int64), ("max_latency_ms", C.c_float), ("pass_in_flight", C.c_uint32)]


class WindowStats(C.Structure):
    _fields_ = [("bin_down", C.c_int32), ("peak_down", C.c_float), ("total_down", C.c_float), ("bin_up", C.c_int32), ("peak_up", C.c_float), ("total_up", C.c_float)]


class Preamble(C.Structure):
    _fields_ = [("header_pos", C.c_int64), ("run_pos", C.c_int64), ("stream", C.c_uint32), ("run_len", C.c_uint32), ("bin", C.c_int32), ("sfd_index", C.c_int32),
                ("pmr", C.c_float), ("cfo_bins", C.c_float), ("cfo_hz", C.c_float), ("delta", C.c_int32)]


class LoraHipError(RuntimeError):
    def __init__(self, status: int, msg: str):
        super().__init__("lora_hip status %d: %s" % (status, msg))
        self.status = status


_lib = None


def load():
    """Loads liblora_hip.so; raises if it was not built (run __graft_entry__.build())."""
    global _lib
    if _lib is not None:
        return _lib
    # One HIP runtime per process: torch bundles its own libamdhip64.so (SONAME
    # libamdhip64.so.7).  Importing torch first makes the dynamic loader bind this
    # library to that same runtime; loading /opt/rocm's copy beside it leaves one
    # of the two without devices.
    try:
        import torch  # noqa: F401
    except ImportError:
        pass
    if not os.path.exists(LIB_PATH):
        raise ImportError("liblora_hip.so not built: run `python -c 'import __graft_entry__ as g; g.build()'` "
                          "(hipcc --offload-arch=gfx950).  There is no CPU fallback.")
    L = C.CDLL(LIB_PATH)
    vp = C.c_void_p
    L.lora_hip_abi_version.restype = C.c_uint32
    L.lora_hip_strerror.restype = C.c_char_p
    L.lora_hip_strerror.argtypes = [C.c_int]
    L.lora_hip_last_error.restype = C.c_char_p
    L.lora_hip_last_error.argtypes = [vp]
    L.lora_hip_create.argtypes = [C.POINTER(Config), C.POINTER(vp)]
    L.lora_hip_destroy.argtypes = [vp]
    L.lora_hip_destroy.restype = None
    L.lora_hip_get_geometry.argtypes = [vp, C.POINTER(C.c_uint32), C.POINTER(C.c_uint32), C.POINTER(C.c_uint32)]
    L.lora_hip_set_sf.argtypes = [vp, C.c_uint8]
    L.lora_hip_set_samp_rate.argtypes = [vp, C.c_float]
    L.lora_hip_work.argtypes = [vp, vp, C.c_size_t, C.POINTER(C.c_size_t)]
    L.lora_hip_flush.argtypes = [vp]
    L.lora_hip_decode_device.argtypes = [vp, vp, C.c_size_t, vp, vp, C.c_uint32, vp]
    L.lora_hip_frames_available.restype = C.c_size_t
    L.lora_hip_frames_available.argtypes = [vp]
    L.lora_hip_poll_frame.argtypes = [vp, vp, C.c_size_t, C.POINTER(C.c_size_t), C.POINTER(FrameInfo)]
    L.lora_hip_drain_frames.argtypes = [vp, vp, C.c_size_t, C.POINTER(FrameInfo), C.c_size_t, C.POINTER(C.c_size_t)]
    L.lora_hip_drain_slots.argtypes = [vp, vp, C.c_size_t, C.c_size_t, C.POINTER(C.c_size_t)]
    L.lora_hip_demod_symbols_device.argtypes = [vp, vp, C.c_size_t, vp, C.c_size_t, C.c_int, vp, vp]
    L.lora_hip_demod_symbols_ex_device.argtypes = [vp, vp, C.c_size_t, vp, C.c_size_t, C.c_int, vp, vp, vp]
    L.lora_hip_last_timing.argtypes = [vp, C.POINTER(Timing)]
    L.lora_hip_last_plan.argtypes = [vp, C.POINTER(C.c_uint32), C.POINTER(C.c_uint32)]
    L.lora_hip_last_payload_pass.argtypes = [vp, C.POINTER(C.c_uint32), C.POINTER(C.c_uint32), C.POINTER(C.c_uint32), C.POINTER(C.c_uint32), C.POINTER(C.c_uint32), C.POINTER(C.c_float)]
    L.lora_hip_decode_device_begin.argtypes = [vp, vp, C.c_size_t, vp, vp, C.c_uint32, vp, C.c_uint32]
    L.lora_hip_decode_device_end.argtypes = [vp]
    L.lora_hip_decode_device_prepass.argtypes = [vp, vp, C.c_size_t, vp, vp, C.c_uint32, vp, C.c_uint32]
    L.lora_hip_gap_starts_device.argtypes = [vp, vp, C.c_size_t, vp, vp, C.c_uint32, vp, C.c_size_t, vp, vp]
    L.lora_hip_trace.restype = C.c_size_t
    L.lora_hip_trace.argtypes = [vp, C.POINTER(C.POINTER(Step))]
    L.lora_hip_trace_clear.argtypes = [vp]
    L.lora_hip_trace_clear.restype = None
    L.lora_hip_estimate_cfo_device.argtypes = [vp, vp, C.c_size_t, C.POINTER(C.c_int64), C.c_size_t, C.c_int, C.POINTER(C.c_float), vp]
    L.lora_hip_estimate_cfo_device.restype = C.c_int
    L.lora_hip_ref_ifreq_device.argtypes = [vp, vp, C.c_size_t, C.POINTER(C.c_float), C.POINTER(C.c_float), vp]
    L.lora_hip_ref_ifreq_device.restype = C.c_int
    if hasattr(L, "lora_hip_get_table") or not os.environ.get("LORA_HIP_LIB"):   # (a library variant of an older ABI under tools/ab.sh does without)
        L.lora_hip_get_table.argtypes = [vp, C.c_int, C.POINTER(C.c_float), C.c_size_t, C.POINTER(C.c_size_t)]
        L.lora_hip_get_table.restype = C.c_int
    L.lora_hip_window_stats_device.argtypes = [vp, vp, C.c_size_t, vp, C.c_size_t, C.POINTER(WindowStats), vp]
    L.lora_hip_detect_preambles_device.argtypes = [vp, vp, C.c_size_t, vp, vp, C.c_uint32, C.c_float, C.POINTER(Preamble), C.c_size_t, C.POINTER(C.c_size_t), vp]
    L.lora_hip_decode_at_headers_device.argtypes = [vp, vp, C.c_size_t, vp, vp, C.c_uint32, C.POINTER(Preamble), C.c_size_t, vp]
    L.lora_hip_walker_kernel_name.argtypes = [vp]
    L.lora_hip_walker_kernel_name.restype = C.c_char_p
    L.lora_hip_mux_create.argtypes = [C.POINTER(Config), C.c_uint32, C.POINTER(vp)]
    L.lora_hip_mux_destroy.argtypes = [vp]
    L.lora_hip_mux_destroy.restype = None
    L.lora_hip_mux_work.argtypes = [vp, C.c_uint32, vp, C.c_size_t]
    L.lora_hip_mux_flush.argtypes = [vp]
    L.lora_hip_mux_set_latency.argtypes = [vp, C.c_float]
    L.lora_hip_mux_set_max_ahead.argtypes = [vp, C.c_size_t]
    L.lora_hip_mux_frames_available.argtypes = [vp]
    L.lora_hip_mux_frames_available.restype = C.c_size_t
    L.lora_hip_mux_poll_frame.argtypes = [vp, vp, C.c_size_t, C.POINTER(C.c_size_t), C.POINTER(FrameInfo)]
    L.lora_hip_mux_passes.argtypes = [vp, C.POINTER(C.c_uint64), C.POINTER(C.c_uint64)]
    L.lora_hip_mux_last_error.argtypes = [vp]
    L.lora_hip_mux_last_error.restype = C.c_char_p
    L.lora_hip_set_stream_latency.argtypes = [vp, C.c_float]
    L.lora_hip_stream_info.argtypes = [vp, C.POINTER(StreamInfo)]
    L.lora_hip_check_frame.argtypes = [C.c_char_p, C.c_size_t, C.POINTER(FrameCheck)]
    L.lora_hip_check_frame.restype = C.c_int
    L.lora_hip_channelizer_create.argtypes = [C.POINTER(ChannelizerConfig), C.POINTER(vp)]
    L.lora_hip_channelizer_destroy.argtypes = [vp]
    L.lora_hip_channelizer_destroy.restype = None
    L.lora_hip_channelizer_last_error.argtypes = [vp]
    L.lora_hip_channelizer_last_error.restype = C.c_char_p
    L.lora_hip_channelizer_taps.argtypes = [vp, vp, C.c_size_t, C.POINTER(C.c_size_t)]
    L.lora_hip_channelizer_output_items.argtypes = [vp, C.c_size_t]
    L.lora_hip_channelizer_output_items.restype = C.c_size_t
    L.lora_hip_channelizer_run_device.argtypes = [vp, vp, C.c_size_t, vp, C.c_size_t, C.POINTER(C.c_size_t), vp]
    L.lora_hip_channelizer_work.argtypes = [vp, vp, C.c_size_t, vp, C.c_size_t, C.POINTER(C.c_size_t)]
    L.lora_hip_channelizer_apply_cfo.argtypes = [vp, C.c_float]
    L.lora_hip_channelizer_last_kernel_ms.argtypes = [vp]
    L.lora_hip_channelizer_last_kernel_ms.restype = C.c_float
    _lib = L
    return L


def check_frame(blob: bytes) -> FrameCheck:
    """Header checksum and payload CRC of a published frame blob (lora_hip_check_frame: host only, no GPU needed)."""
    out = FrameCheck()
    st = load().lora_hip_check_frame(bytes(blob), len(blob), C.byref(out))
    if st != 0:
        raise LoraHipError(st, "lora_hip_check_frame: blob too short")
    return out


class Handle:
    """Thin RAII wrapper of lora_hip_decoder_t."""

    def __init__(self, samp_rate=1e6, bandwidth=125000, sf=7, implicit=False, cr=4, crc=True, reduced_rate=False,
                 disable_drift_correction=False, device=0, demod=DEMOD_FFT_COMPAT, flags=0, segment_symbols=0,
                 batch_items=0):
        self.L = load()
        cfg = Config(struct_size=C.sizeof(Config), samp_rate=float(samp_rate), bandwidth=int(bandwidth), sf=int(sf),
                     implicit=int(bool(implicit)), cr=int(cr), crc=int(bool(crc)), reduced_rate=int(bool(reduced_rate)),
                     disable_drift_correction=int(bool(disable_drift_correction)), device=int(device), demod=int(demod),
                     flags=int(flags), segment_symbols=int(segment_symbols), batch_items=int(batch_items))
        h = C.c_void_p()
        st = self.L.lora_hip_create(C.byref(cfg), C.byref(h))
        if st != 0:
            raise LoraHipError(st, "%s (%s)" % (self.L.lora_hip_strerror(st).decode(), self.L.lora_hip_last_error(None).decode()))
        self.h = h
        a, b, d = C.c_uint32(), C.c_uint32(), C.c_uint32()
        self.L.lora_hip_get_geometry(self.h, C.byref(a), C.byref(b), C.byref(d))
        self.sps, self.nbins, self.decim = a.value, b.value, d.value
        self.device, self.batch_items = int(device), int(batch_items)

    def close(self):
        if getattr(self, "h", None):
            self.L.lora_hip_destroy(self.h)
            self.h = None

    __del__ = close

    def _check(self, st: int):
        if st != 0:
            raise LoraHipError(st, "%s (%s)" % (self.L.lora_hip_strerror(st).decode(), self.L.lora_hip_last_error(self.h).decode()))

    # streaming (host buffers)
    def work(self, iq: np.ndarray) -> int:
        a = np.ascontiguousarray(iq, dtype=np.complex64)
        n = C.c_size_t(0)
        self._check(self.L.lora_hip_work(self.h, a.ctypes.data, a.size, C.byref(n)))
        return n.value

    def flush(self):
        self._check(self.L.lora_hip_flush(self.h))

    def kernel_name(self) -> str:
        """lora_hip_walker_kernel_name: the state-machine kernel this handle's passes launch."""
        return self.L.lora_hip_walker_kernel_name(self.h).decode()

    def set_stream_latency(self, max_latency_ms: float):
        """lora_hip_set_stream_latency: wall-clock bound on how long a delivered sample waits for a device pass (0 = off)."""
        self._check(self.L.lora_hip_set_stream_latency(self.h, float(max_latency_ms)))

    def stream_info(self) -> StreamInfo:
        out = StreamInfo()
        self._check(self.L.lora_hip_stream_info(self.h, C.byref(out)))
        return out

    # batched, device-resident
    def decode_device(self, dev_ptr: int, total_items: int, offs: Sequence[int], lens: Sequence[int], stream: int = 0):
        o = np.ascontiguousarray(offs, dtype=np.uint64)
        l = np.ascontiguousarray(lens, dtype=np.uint64)
        self._check(self.L.lora_hip_decode_device(self.h, dev_ptr, total_items, o.ctypes.data, l.ctypes.data, o.size, stream))

    def decode_device_begin(self, dev_ptr: int, total_items: int, offs: Sequence[int], lens: Sequence[int], stream: int = 0, iq_ready: bool = False):
        """First half of a pass (plan + main launch); finish it with decode_device_end().  iq_ready: LORA_HIP_BEGIN_IQ_READY."""
        o = np.ascontiguousarray(offs, dtype=np.uint64)
        l = np.ascontiguousarray(lens, dtype=np.uint64)
        self._check(self.L.lora_hip_decode_device_begin(self.h, dev_ptr, total_items, o.ctypes.data, l.ctypes.data, o.size, stream, 1 if iq_ready else 0))

    def decode_device_prepass(self, dev_ptr: int, total_items: int, offs: Sequence[int], lens: Sequence[int], stream: int = 0, iq_ready: bool = False):
        """Issues the envelope pre-pass of the pass that will be begun next on this handle (lora_hip_decode_device_prepass)."""
        o = np.ascontiguousarray(offs, dtype=np.uint64)
        l = np.ascontiguousarray(lens, dtype=np.uint64)
        self._check(self.L.lora_hip_decode_device_prepass(self.h, dev_ptr, total_items, o.ctypes.data, l.ctypes.data, o.size, stream, 1 if iq_ready else 0))

    def decode_device_end(self):
        self._check(self.L.lora_hip_decode_device_end(self.h))

    def gap_starts_device(self, dev_ptr: int, total_items: int, offs: Sequence[int], lens: Sequence[int], stream: int = 0):
        """Per stream, the item positions where the envelope pre-pass sees a gap begin (lora_hip_gap_starts_device)."""
        o = np.ascontiguousarray(offs, dtype=np.uint64)
        l = np.ascontiguousarray(lens, dtype=np.uint64)
        cap = int(sum(int(x) for x in l) // 64 + 16)
        pos = np.zeros(cap, dtype=np.int64)
        cnt = np.zeros(o.size, dtype=np.uint32)
        self._check(self.L.lora_hip_gap_starts_device(self.h, dev_ptr, total_items, o.ctypes.data, l.ctypes.data, o.size, pos.ctypes.data, cap,
                                                      cnt.ctypes.data, stream))
        out, k = [], 0
        for c in cnt:
            out.append(pos[k:k + int(c)].copy()); k += int(c)
        return out

    def demod_symbols_device(self, dev_ptr: int, total_items: int, offsets: Sequence[int], demod: int, stream: int = 0) -> np.ndarray:
        off = np.ascontiguousarray(offsets, dtype=np.int64)
        out = np.zeros(off.size, dtype=np.uint32)
        self._check(self.L.lora_hip_demod_symbols_device(self.h, dev_ptr, total_items, off.ctypes.data, off.size, demod, out.ctypes.data, stream))
        return out

    def demod_symbols_ex_device(self, dev_ptr: int, total_items: int, offsets: Sequence[int], demod: int, stream: int = 0):
        """(shifts, fine): get_shift_fft's value and d_fine_sync after the per-symbol fine_sync, per window."""
        off = np.ascontiguousarray(offsets, dtype=np.int64)
        out = np.zeros(off.size, dtype=np.uint32)
        fine = np.zeros(off.size, dtype=np.int32)
        self._check(self.L.lora_hip_demod_symbols_ex_device(self.h, dev_ptr, total_items, off.ctypes.data, off.size, demod, out.ctypes.data,
                                                            fine.ctypes.data, stream))
        return out, fine

    def estimate_cfo_device(self, dev_ptr: int, total_items: int, offsets: Sequence[int], mode: int = 1, stream: int = 0) -> np.ndarray:
        """CFO in Hz of the windows at `offsets` (experimental_determine_cfo, decoder_impl.cc:730-738; mode 1: mean over the window)."""
        off = np.ascontiguousarray(offsets, dtype=np.int64)
        out = np.zeros(off.size, dtype=np.float32)
        self._check(self.L.lora_hip_estimate_cfo_device(self.h, C.c_void_p(dev_ptr), total_items, off.ctypes.data_as(C.POINTER(C.c_int64)), off.size, mode,
                                                        out.ctypes.data_as(C.POINTER(C.c_float)), C.c_void_p(stream)))
        return out

    def ref_ifreq_device(self, dev_ptr: int, n_items: int, stream: int = 0):
        """(atan2f of every item, instantaneous_frequency of every pair) as the strict SYNC path computes them (decoder_impl.cc:231-240)."""
        arg = np.zeros(n_items, dtype=np.float32)
        f = np.zeros(n_items - 1, dtype=np.float32)
        self._check(self.L.lora_hip_ref_ifreq_device(self.h, C.c_void_p(dev_ptr), n_items, arg.ctypes.data_as(C.POINTER(C.c_float)),
                                                     f.ctypes.data_as(C.POINTER(C.c_float)), C.c_void_p(stream)))
        return arg, f

    def window_stats_device(self, dev_ptr: int, total_items: int, offsets: Sequence[int], stream: int = 0):
        """lora_hip_window_stats_device: per window (bin_down, peak_down, total_down, bin_up, peak_up, total_up)."""
        off = np.ascontiguousarray(offsets, dtype=np.int64)
        out = (WindowStats * max(off.size, 1))()
        self._check(self.L.lora_hip_window_stats_device(self.h, dev_ptr, total_items, off.ctypes.data, off.size, out, stream))
        return [(o.bin_down, o.peak_down, o.total_down, o.bin_up, o.peak_up, o.total_up) for o in out[: off.size]]

    def detect_preambles_device(self, dev_ptr: int, total_items: int, offs: Sequence[int], lens: Sequence[int], threshold: float = 0.0, stream: int = 0, cap: int = 4096):
        """lora_hip_detect_preambles_device: FFT-domain preamble detection (acquires below 0 dB); list of dicts."""
        o = np.ascontiguousarray(offs, dtype=np.uint64)
        l = np.ascontiguousarray(lens, dtype=np.uint64)
        out = (Preamble * cap)()
        n = C.c_size_t(0)
        self._check(self.L.lora_hip_detect_preambles_device(self.h, dev_ptr, total_items, o.ctypes.data, l.ctypes.data, o.size, float(threshold), out, cap, C.byref(n), stream))
        return [dict(header_pos=p.header_pos, run_pos=p.run_pos, stream=p.stream, run_len=p.run_len, bin=p.bin, sfd_index=p.sfd_index, pmr=p.pmr,
                     cfo_bins=p.cfo_bins, cfo_hz=p.cfo_hz, delta=p.delta) for p in out[: n.value]]

    def decode_at_headers_device(self, dev_ptr: int, total_items: int, offs: Sequence[int], lens: Sequence[int], preambles, stream: int = 0):
        """lora_hip_decode_at_headers_device: decodes the packets at the detector's header positions (dicts of detect_preambles_device, or
        (stream, header_pos) pairs); frames go to the handle's queue."""
        o = np.ascontiguousarray(offs, dtype=np.uint64)
        l = np.ascontiguousarray(lens, dtype=np.uint64)
        arr = (Preamble * max(len(preambles), 1))()
        for i, p in enumerate(preambles):
            st, hp = (p["stream"], p["header_pos"]) if isinstance(p, dict) else p
            arr[i].stream, arr[i].header_pos = int(st), int(hp)
        self._check(self.L.lora_hip_decode_at_headers_device(self.h, dev_ptr, total_items, o.ctypes.data, l.ctypes.data, o.size, arr, len(preambles), stream))

    def frames_available(self) -> int:
        return self.L.lora_hip_frames_available(self.h)

    def poll_frame(self) -> Optional[Tuple[bytes, FrameInfo]]:
        buf = (C.c_uint8 * 320)()
        n = C.c_size_t(0)
        info = FrameInfo()
        self._check(self.L.lora_hip_poll_frame(self.h, buf, 320, C.byref(n), C.byref(info)))
        if n.value == 0:
            return None
        return bytes(buf[: n.value]), info

    def drain(self) -> List[Tuple[bytes, FrameInfo]]:
        """All queued frames, through the bulk call (one ABI crossing per 1024 frames)."""
        out = []
        while True:
            n_avail = self.frames_available()
            if n_avail == 0:
                return out
            k = min(n_avail, 4096)
            buf = (C.c_uint8 * (k * 280))()
            infos = (FrameInfo * k)()
            n = C.c_size_t(0)
            self._check(self.L.lora_hip_drain_frames(self.h, buf, k * 280, infos, k, C.byref(n)))
            raw = bytes(buf)
            off = 0
            for i in range(n.value):
                ln = infos[i].length
                inf = FrameInfo(infos[i].stream, ln, infos[i].header_pos, infos[i].end_pos)
                out.append((raw[off:off + ln], inf))
                off += ln
            if n.value == 0:
                return out

    INFO_DTYPE = np.dtype([("stream", "<u4"), ("length", "<u4"), ("header_pos", "<i8"), ("end_pos", "<i8")])

    def drain_raw(self):
        """All queued frames without per-frame Python objects: (blob bytes back to back as uint8[], infos as a
        structured array with fields stream / length / header_pos / end_pos)."""
        bufs, infos = [], []
        while True:
            n_avail = self.frames_available()
            if n_avail == 0:
                break
            k = min(n_avail, 8192)
            buf = np.empty(k * 280, dtype=np.uint8)
            inf = np.empty(k, dtype=self.INFO_DTYPE)
            n = C.c_size_t(0)
            self._check(self.L.lora_hip_drain_frames(self.h, buf.ctypes.data, buf.size, C.cast(inf.ctypes.data, C.POINTER(FrameInfo)), k, C.byref(n)))
            if n.value == 0:
                break
            inf = inf[: n.value]
            bufs.append(buf[: int(inf["length"].sum())])
            infos.append(inf)
        if not bufs:
            return np.empty(0, dtype=np.uint8), np.empty(0, dtype=self.INFO_DTYPE)
        return np.concatenate(bufs), np.concatenate(infos)

    def drain_slots(self, slot_bytes: int) -> np.ndarray:
        """All queued frames as fixed-size slots uint8[n, slot_bytes]: u32 stream | u32 length | i64 header_pos | blob | zeros."""
        n_avail = self.frames_available()
        out = np.empty((max(n_avail, 1), slot_bytes), dtype=np.uint8)
        n = C.c_size_t(0)
        self._check(self.L.lora_hip_drain_slots(self.h, out.ctypes.data, slot_bytes, n_avail, C.byref(n)))
        return out[: n.value]

    def timing(self) -> Timing:
        t = Timing()
        self._check(self.L.lora_hip_last_timing(self.h, C.byref(t)))
        return t

    def plan(self):
        """(burst_aware, segments) of the last pass: lora_hip_last_plan."""
        b, n = C.c_uint32(0), C.c_uint32(0)
        self._check(self.L.lora_hip_last_plan(self.h, C.byref(b), C.byref(n)))
        return bool(b.value), int(n.value)

    def table(self, which: int) -> np.ndarray:
        """The handle's ideal-chirp table `which` (0 downchirp, 1 upchirp, 2 downchirp ifreq, 3 upchirp ifreq, 4 d_upchirp_ifreq_v + guard) as float32: lora_hip_get_table."""
        n = C.c_size_t()
        self._check(self.L.lora_hip_get_table(self.h, which, None, 0, C.byref(n)))
        out = np.empty(n.value, np.float32)
        self._check(self.L.lora_hip_get_table(self.h, which, out.ctypes.data_as(C.POINTER(C.c_float)), n.value, C.byref(n)))
        return out

    def payload_pass(self):
        """dict(packets, moved, rerun, rounds, symbols, ms) of the last pass's payload pass (all zero unless it ran decoupled): lora_hip_last_payload_pass."""
        a, mv, b, r, c, m = C.c_uint32(0), C.c_uint32(0), C.c_uint32(0), C.c_uint32(0), C.c_uint32(0), C.c_float(0)
        self._check(self.L.lora_hip_last_payload_pass(self.h, C.byref(a), C.byref(mv), C.byref(b), C.byref(r), C.byref(c), C.byref(m)))
        return dict(packets=int(a.value), moved=int(mv.value), rerun=int(b.value), rounds=int(r.value), symbols=int(c.value), ms=float(m.value))

    def trace(self):
        p = C.POINTER(Step)()
        n = self.L.lora_hip_trace(self.h, C.byref(p))
        return [(p[i].state, p[i].pos, p[i].consumed, p[i].bin, p[i].fine, p[i].value, p[i].stream, p[i].cycles) for i in range(n)]

    def trace_clear(self):
        self.L.lora_hip_trace_clear(self.h)


class Mux:
    """lora_hip_mux_*: n_channels independent decoders fed in any order, decoded by ONE device pass per chunk (the gateway flowgraph)."""

    def __init__(self, n_channels, samp_rate=1e6, bandwidth=125000, sf=7, implicit=False, cr=4, crc=True, reduced_rate=False,
                 disable_drift_correction=False, device=0, demod=DEMOD_FFT_COMPAT, flags=0, segment_symbols=0, batch_items=0):
        self.L = load()
        cfg = Config(struct_size=C.sizeof(Config), samp_rate=float(samp_rate), bandwidth=int(bandwidth), sf=int(sf), implicit=int(bool(implicit)), cr=int(cr),
                     crc=int(bool(crc)), reduced_rate=int(bool(reduced_rate)), disable_drift_correction=int(bool(disable_drift_correction)), device=int(device),
                     demod=int(demod), flags=int(flags), segment_symbols=int(segment_symbols), batch_items=int(batch_items))
        self.h = C.c_void_p()
        st = self.L.lora_hip_mux_create(C.byref(cfg), int(n_channels), C.byref(self.h))
        if st != 0:
            raise LoraHipError(st, "%s (%s)" % (self.L.lora_hip_strerror(st).decode(), self.L.lora_hip_last_error(None).decode()))
        self.n_channels = int(n_channels)

    def _check(self, st):
        if st != 0:
            raise LoraHipError(st, "%s (%s)" % (self.L.lora_hip_strerror(st).decode(), (self.L.lora_hip_mux_last_error(self.h) or b"").decode()))

    def work(self, channel: int, iq: np.ndarray):
        a = np.ascontiguousarray(iq, dtype=np.complex64)
        self._check(self.L.lora_hip_mux_work(self.h, int(channel), a.ctypes.data, a.size))

    def flush(self):
        self._check(self.L.lora_hip_mux_flush(self.h))

    def set_latency(self, ms: float):
        self._check(self.L.lora_hip_mux_set_latency(self.h, float(ms)))

    def set_max_ahead(self, items: int):
        self._check(self.L.lora_hip_mux_set_max_ahead(self.h, int(items)))

    def passes(self):
        a, b = C.c_uint64(0), C.c_uint64(0)
        self._check(self.L.lora_hip_mux_passes(self.h, C.byref(a), C.byref(b)))
        return int(a.value), int(b.value)

    def drain(self) -> List[Tuple[bytes, FrameInfo]]:
        out = []
        buf = (C.c_uint8 * 320)()
        while self.L.lora_hip_mux_frames_available(self.h):
            n = C.c_size_t(0)
            info = FrameInfo()
            self._check(self.L.lora_hip_mux_poll_frame(self.h, buf, 320, C.byref(n), C.byref(info)))
            out.append((bytes(buf[: n.value]), info))
        return out

    def close(self):
        if getattr(self, "h", None):
            self.L.lora_hip_mux_destroy(self.h)
            self.h = None

    __del__ = close


class Channelizer:
    """lora_hip_channelizer_* (include/lora_hip_channelizer.h): the frequency-translating FIR in front of the decoder."""

    def __init__(self, samp_rate, center_freq, channel_list, bandwidth, decimation=1, device=0, cutoff_hz=0.0, transition_hz=0.0, flags=0):
        self.L = load()
        self.n_channels = len(channel_list)
        self._chan = (C.c_float * self.n_channels)(*[float(c) for c in channel_list])
        cfg = ChannelizerConfig(struct_size=C.sizeof(ChannelizerConfig), samp_rate=float(samp_rate), center_freq=float(center_freq),
                                channel_list=self._chan, n_channels=self.n_channels, bandwidth=int(bandwidth), decimation=int(decimation), device=int(device),
                                cutoff_hz=float(cutoff_hz), transition_hz=float(transition_hz), flags=int(flags))
        self.h = C.c_void_p()
        st = self.L.lora_hip_channelizer_create(C.byref(cfg), C.byref(self.h))
        if st != 0:
            raise LoraHipError(st, self.L.lora_hip_strerror(st).decode())

    def _check(self, st):
        if st != 0:
            raise LoraHipError(st, (self.L.lora_hip_channelizer_last_error(self.h) or b"").decode() or self.L.lora_hip_strerror(st).decode())

    def taps(self) -> np.ndarray:
        n = C.c_size_t(0)
        self._check(self.L.lora_hip_channelizer_taps(self.h, None, 0, C.byref(n)))
        t = np.zeros(n.value, dtype=np.float32)
        self._check(self.L.lora_hip_channelizer_taps(self.h, t.ctypes.data, t.size, C.byref(n)))
        return t

    def output_items(self, n_in: int) -> int:
        return int(self.L.lora_hip_channelizer_output_items(self.h, n_in))

    def work(self, x) -> np.ndarray:
        """Host buffers in and out: complex64[n_in] -> complex64[n_channels, n_out]."""
        a = np.ascontiguousarray(x, dtype=np.complex64)
        no = self.output_items(a.size)
        out = np.zeros((self.n_channels, max(no, 1)), dtype=np.complex64)
        n = C.c_size_t(0)
        self._check(self.L.lora_hip_channelizer_work(self.h, a.ctypes.data, a.size, out.ctypes.data, out.shape[1], C.byref(n)))
        return out[:, : n.value]

    def run_device(self, d_in: int, n_in: int, d_out: int, out_stride: int, stream: int = 0) -> int:
        n = C.c_size_t(0)
        self._check(self.L.lora_hip_channelizer_run_device(self.h, d_in, n_in, d_out, out_stride, C.byref(n), stream))
        return int(n.value)

    def apply_cfo(self, cfo: float):
        self._check(self.L.lora_hip_channelizer_apply_cfo(self.h, float(cfo)))

    def kernel_ms(self) -> float:
        return float(self.L.lora_hip_channelizer_last_kernel_ms(self.h))

    def close(self):
        if self.h:
            self.L.lora_hip_channelizer_destroy(self.h)
            self.h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
