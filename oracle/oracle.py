"""ctypes binding of the CPU oracle (TEST INFRASTRUCTURE ONLY).

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg import
this module.  The product package gr_lora_amd never does.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess
from typing import List, Optional, Sequence

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "liblora_oracle.so")

DEMOD_GRAD, DEMOD_FFT, DEMOD_FFT_COMPAT = 0, 1, 2
ST_NAMES = ["DETECT", "SYNC", "FIND_SFD", "PAUSE", "DECODE_HEADER", "DECODE_PAYLOAD", "STOP"]


class Step(C.Structure):
    _fields_ = [("state", C.c_int32), ("pos", C.c_int64), ("consumed", C.c_int32),
                ("bin", C.c_int32), ("fine", C.c_int32), ("value", C.c_float)]


def build(force: bool = False) -> str:
    if force or not os.path.exists(_LIB_PATH) or os.path.getmtime(_LIB_PATH) < os.path.getmtime(os.path.join(_HERE, "lora_oracle.c")):
        subprocess.check_call(["make", "-C", _HERE, "-s"])
    return _LIB_PATH


_lib = None
_lib_fast = None
_FAST_PATH = os.path.join(_HERE, "liblora_oracle_fast.so")


def lib_fast():
    """The CPU-BASELINE build (-O3 -march=native): rebuilt on the machine that runs it (the flags are
    host-specific), used by bench.py's cpu_baseline leg only - never for parity."""
    global _lib_fast
    if _lib_fast is None:
        subprocess.check_call(["make", "-C", _HERE, "-s", "-B", "liblora_oracle_fast.so"])
        _lib_fast = _bind(C.CDLL(_FAST_PATH))
    return _lib_fast


def lib():
    global _lib
    if _lib is None:
        build()
        _lib = _bind(C.CDLL(_LIB_PATH))
    return _lib


def _bind(L):
    if True:
        fp = C.POINTER(C.c_float)
        L.lora_oracle_create.restype = C.c_void_p
        L.lora_oracle_create.argtypes = [C.c_float, C.c_uint32, C.c_uint8, C.c_int, C.c_uint8, C.c_int, C.c_int, C.c_int, C.c_int]
        L.lora_oracle_destroy.argtypes = [C.c_void_p]
        L.lora_oracle_run.restype = C.c_size_t
        L.lora_oracle_run.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t]
        L.lora_oracle_num_frames.argtypes = [C.c_void_p]
        L.lora_oracle_get_frame.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_int]
        L.lora_oracle_frame_pos.restype = C.c_int64
        L.lora_oracle_frame_pos.argtypes = [C.c_void_p, C.c_int]
        L.lora_oracle_clear_frames.argtypes = [C.c_void_p]
        L.lora_oracle_enable_trace.argtypes = [C.c_void_p, C.c_int]
        L.lora_oracle_trace.restype = C.c_size_t
        L.lora_oracle_trace.argtypes = [C.c_void_p, C.POINTER(C.POINTER(Step))]
        L.lora_oracle_sps.restype = C.c_uint32
        L.lora_oracle_sps.argtypes = [C.c_void_p]
        L.lora_oracle_bins.restype = C.c_uint32
        L.lora_oracle_bins.argtypes = [C.c_void_p]
        L.lora_oracle_table.restype = fp
        L.lora_oracle_table.argtypes = [C.c_void_p, C.c_int, C.POINTER(C.c_size_t)]
        L.lora_oracle_get_shift_fft.restype = C.c_uint32
        L.lora_oracle_get_shift_fft.argtypes = [C.c_void_p, C.c_void_p]
        L.lora_oracle_determine_cfo.restype = C.c_float
        L.lora_oracle_determine_cfo.argtypes = [C.c_void_p, C.c_void_p, C.c_int]
        L.lora_oracle_max_frequency_gradient_idx.restype = C.c_uint32
        L.lora_oracle_max_frequency_gradient_idx.argtypes = [C.c_void_p, C.c_void_p]
        L.lora_oracle_fine_sync.restype = C.c_int32
        L.lora_oracle_fine_sync.argtypes = [C.c_void_p, C.c_void_p, C.c_int32, C.c_int32]
        L.lora_oracle_detect_preamble_autocorr.restype = C.c_float
        L.lora_oracle_detect_preamble_autocorr.argtypes = [C.c_void_p, C.c_void_p]
        L.lora_oracle_detect_downchirp.restype = C.c_float
        L.lora_oracle_detect_downchirp.argtypes = [C.c_void_p, C.c_void_p]
        L.lora_oracle_detect_upchirp.restype = C.c_float
        L.lora_oracle_detect_upchirp.argtypes = [C.c_void_p, C.c_void_p, C.POINTER(C.c_int32)]
        L.lora_oracle_instantaneous_frequency.argtypes = [C.c_void_p, C.c_void_p, C.c_uint32]
        L.lora_oracle_demod_at.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t, C.c_int, C.c_void_p]
        L.lora_oracle_rotl.restype = C.c_uint32
        L.lora_oracle_rotl.argtypes = [C.c_uint32, C.c_uint32, C.c_uint32]
        L.lora_oracle_hamming_encode.restype = C.c_uint8
        L.lora_oracle_hamming_encode.argtypes = [C.c_uint8]
        L.lora_oracle_hamming84_decode.restype = C.c_uint8
        L.lora_oracle_hamming84_decode.argtypes = [C.c_uint8]
        L.lora_oracle_deinterleave.argtypes = [C.c_void_p, C.c_uint32, C.c_uint32, C.c_void_p]
        L.lora_oracle_deshuffle_byte.restype = C.c_uint8
        L.lora_oracle_deshuffle_byte.argtypes = [C.c_uint8]
        L.lora_oracle_snr_byte.restype = C.c_uint8
        L.lora_oracle_snr_byte.argtypes = [C.c_float]
    return L


def _iq(a) -> np.ndarray:
    a = np.ascontiguousarray(a, dtype=np.complex64)
    return a


class Oracle:
    """One reference-decoder instance (same constructor arguments as
    gr::lora::decoder::make, include/lora/decoder.h:705)."""

    def __init__(self, samp_rate=1e6, bandwidth=125000, sf=7, implicit=False, cr=4, crc=True,
                 reduced_rate=False, disable_drift_correction=False, demod=DEMOD_GRAD, fast=False):
        self.L = lib_fast() if fast else lib()
        self.h = self.L.lora_oracle_create(samp_rate, int(bandwidth), int(sf), int(implicit), int(cr), int(crc),
                                           int(reduced_rate), int(disable_drift_correction), int(demod))
        if not self.h:
            raise ValueError("oracle: unsupported configuration (reference would exit(1))")
        self.sps = self.L.lora_oracle_sps(self.h)
        self.nbins = self.L.lora_oracle_bins(self.h)

    def __del__(self):
        if getattr(self, "h", None):
            self.L.lora_oracle_destroy(self.h)
            self.h = None

    def run(self, iq) -> int:
        a = _iq(iq)
        return self.L.lora_oracle_run(self.h, a.ctypes.data, a.size)

    def frames(self) -> List[bytes]:
        out = []
        for i in range(self.L.lora_oracle_num_frames(self.h)):
            n = self.L.lora_oracle_get_frame(self.h, i, None, 0)
            buf = (C.c_uint8 * n)()
            self.L.lora_oracle_get_frame(self.h, i, buf, n)
            out.append(bytes(buf))
        return out

    def frame_positions(self) -> List[int]:
        return [self.L.lora_oracle_frame_pos(self.h, i) for i in range(self.L.lora_oracle_num_frames(self.h))]

    def clear(self):
        self.L.lora_oracle_clear_frames(self.h)

    def enable_trace(self, on=True):
        self.L.lora_oracle_enable_trace(self.h, int(on))

    def trace(self):
        p = C.POINTER(Step)()
        n = self.L.lora_oracle_trace(self.h, C.byref(p))
        return [(p[i].state, p[i].pos, p[i].consumed, p[i].bin, p[i].fine, p[i].value) for i in range(n)]

    def table(self, which: int) -> np.ndarray:
        n = C.c_size_t()
        p = self.L.lora_oracle_table(self.h, which, C.byref(n))
        return np.ctypeslib.as_array(p, shape=(n.value,)).copy()

    # primitives
    def get_shift_fft(self, iq) -> int:
        a = _iq(iq); assert a.size >= self.sps
        return self.L.lora_oracle_get_shift_fft(self.h, a.ctypes.data)

    def determine_cfo(self, iq, mode: int = 0) -> float:
        a = _iq(iq); assert a.size >= self.sps
        return self.L.lora_oracle_determine_cfo(self.h, a.ctypes.data, mode)

    def max_frequency_gradient_idx(self, iq) -> int:
        a = _iq(iq); assert a.size >= self.sps
        return self.L.lora_oracle_max_frequency_gradient_idx(self.h, a.ctypes.data)

    def fine_sync(self, iq, bin_idx: int, search: int) -> int:
        a = _iq(iq); assert a.size >= self.sps
        return self.L.lora_oracle_fine_sync(self.h, a.ctypes.data, bin_idx, search)

    def detect_preamble_autocorr(self, iq) -> float:
        a = _iq(iq); assert a.size >= 2 * self.sps
        return self.L.lora_oracle_detect_preamble_autocorr(self.h, a.ctypes.data)

    def detect_downchirp(self, iq) -> float:
        a = _iq(iq); assert a.size >= self.sps
        return self.L.lora_oracle_detect_downchirp(self.h, a.ctypes.data)

    def detect_upchirp(self, iq):
        a = _iq(iq); assert a.size >= 2 * self.sps
        idx = C.c_int32(0)
        c = self.L.lora_oracle_detect_upchirp(self.h, a.ctypes.data, C.byref(idx))
        return c, idx.value

    def demod_at(self, iq, offsets: Sequence[int], mode=DEMOD_FFT) -> np.ndarray:
        a = _iq(iq)
        off = np.ascontiguousarray(offsets, dtype=np.int64)
        assert off.size == 0 or (off.min() >= 0 and off.max() + self.sps <= a.size)
        out = np.zeros(off.size, dtype=np.uint32)
        self.L.lora_oracle_demod_at(self.h, a.ctypes.data, off.ctypes.data, off.size, mode, out.ctypes.data)
        return out


def instantaneous_frequency(iq) -> np.ndarray:
    a = _iq(iq)
    out = np.zeros(a.size, dtype=np.float32)
    lib().lora_oracle_instantaneous_frequency(a.ctypes.data, out.ctypes.data, a.size)
    return out


def decode_stream(iq, demod=DEMOD_GRAD, **cfg) -> List[bytes]:
    o = Oracle(demod=demod, **cfg)
    o.run(iq)
    return o.frames()
