"""ctypes binding of oracle/_ref/libref_decoder.so - the REFERENCE decoder itself (lib/decoder_impl.cc compiled
unmodified against stand-in headers, oracle/ref_build/).  TEST INFRASTRUCTURE ONLY: tests/ use it to pin the
restated oracle (oracle/lora_oracle.c) and to generate tests/golden/.  The product never loads it.

The library can only be BUILT where /root/reference exists (this container); on the GPU box the prebuilt file
travels with the snapshot.  `available()` says whether it can be used at all.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess
from typing import List

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_BUILD = os.path.join(_HERE, "ref_build")
_LIB_PATH = os.path.join(_HERE, "_ref", "libref_decoder.so")
_LIB_SIMD_PATH = os.path.join(_HERE, "_ref", "libref_decoder_simd.so")   # VOLK stand-in summing in 8 lanes
_LIB_FAST_PATH = os.path.join(_HERE, "_ref", "libref_decoder_fast.so")   # -O3 -march=x86-64-v3 timing build (bench.py cpu_baseline)
REFERENCE_ROOT = "/root/reference"


class Step(C.Structure):
    _fields_ = [("state", C.c_int32), ("pos", C.c_int64), ("consumed", C.c_int32),
                ("bin", C.c_int32), ("fine", C.c_int32), ("value", C.c_float)]


def build() -> str:
    """(Re)build from the reference sources when they are present; otherwise keep the prebuilt file."""
    if os.path.exists(os.path.join(REFERENCE_ROOT, "lib", "decoder_impl.cc")):
        subprocess.check_call(["make", "-C", _BUILD, "-s"])
    return _LIB_PATH


def available() -> bool:
    try:
        build()
    except Exception:
        pass
    return os.path.exists(_LIB_PATH)


_lib = None
_lib_simd = None
_lib_fast = None


def lib_fast():
    """The timing build (never used for parity)."""
    global _lib_fast
    if _lib_fast is None:
        build()
        _lib_fast = _bind(C.CDLL(_LIB_FAST_PATH))
    return _lib_fast


def lib(simd: bool = False):
    global _lib, _lib_simd
    if simd:
        if _lib_simd is None:
            build()
            _lib_simd = _bind(C.CDLL(_LIB_SIMD_PATH))
        return _lib_simd
    if _lib is None:
        build()
        _lib = _bind(C.CDLL(_LIB_PATH))
    return _lib


def _bind(L):
    if True:
        vp = C.c_void_p
        L.ref_create.restype = vp
        L.ref_create.argtypes = [C.c_float, C.c_uint32, C.c_uint8, C.c_int, C.c_uint8, C.c_int, C.c_int, C.c_int]
        L.ref_destroy.argtypes = [vp]
        L.ref_make_smoke.restype = C.c_int
        for name in ("ref_sps", "ref_bins", "ref_bins_hdr", "ref_decim", "ref_delay_after_sync"):
            getattr(L, name).restype = C.c_uint32
            getattr(L, name).argtypes = [vp]
        for name in ("ref_output_multiple", "ref_state", "ref_phdr_cr", "ref_num_ports", "ref_num_frames"):
            getattr(L, name).restype = C.c_int
            getattr(L, name).argtypes = [vp]
        L.ref_dt.restype = C.c_double
        L.ref_dt.argtypes = [vp]
        L.ref_port_name.restype = C.c_char_p
        L.ref_port_name.argtypes = [vp, C.c_int]
        L.ref_in_sig.restype = C.c_int
        L.ref_in_sig.argtypes = [vp, C.c_int]
        L.ref_stdout.restype = C.c_size_t
        L.ref_stdout.argtypes = [vp, C.c_char_p, C.c_size_t]
        L.ref_enable_trace.argtypes = [vp, C.c_int]
        L.ref_trace.restype = C.c_size_t
        L.ref_trace.argtypes = [vp, C.POINTER(C.POINTER(Step))]
        L.ref_run.restype = C.c_size_t
        L.ref_run.argtypes = [vp, vp, C.c_size_t]
        L.ref_get_frame.restype = C.c_int
        L.ref_get_frame.argtypes = [vp, C.c_int, vp, C.c_int]
        L.ref_frame_pos.restype = C.c_int64
        L.ref_frame_pos.argtypes = [vp, C.c_int]
        L.ref_clear_frames.argtypes = [vp]
        L.ref_table.restype = C.POINTER(C.c_float)
        L.ref_table.argtypes = [vp, C.c_int, C.POINTER(C.c_size_t)]
        L.ref_get_shift_fft.restype = C.c_uint32
        L.ref_get_shift_fft.argtypes = [vp, vp]
        L.ref_experimental_determine_cfo.restype = C.c_float
        L.ref_experimental_determine_cfo.argtypes = [vp, vp, C.c_uint32]
        L.ref_max_frequency_gradient_idx.restype = C.c_uint32
        L.ref_max_frequency_gradient_idx.argtypes = [vp, vp]
        L.ref_fine_sync.restype = C.c_int32
        L.ref_fine_sync.argtypes = [vp, vp, C.c_int32, C.c_int32]
        L.ref_detect_preamble_autocorr.restype = C.c_float
        L.ref_detect_preamble_autocorr.argtypes = [vp, vp]
        L.ref_energy_threshold.restype = C.c_float
        L.ref_energy_threshold.argtypes = [vp]
        L.ref_pwr_queue.restype = C.c_int
        L.ref_pwr_queue.argtypes = [vp, vp]
        L.ref_determine_energy.restype = C.c_float
        L.ref_determine_energy.argtypes = [vp, vp]
        L.ref_detect_downchirp.restype = C.c_float
        L.ref_detect_downchirp.argtypes = [vp, vp]
        L.ref_detect_upchirp.restype = C.c_float
        L.ref_detect_upchirp.argtypes = [vp, vp, C.POINTER(C.c_int32)]
        L.ref_instantaneous_frequency.argtypes = [vp, vp, vp, C.c_uint32]
        L.ref_deinterleave.argtypes = [vp, vp, C.c_uint32, C.c_uint32, vp]
        L.ref_decode.restype = C.c_int
        L.ref_decode.argtypes = [vp, vp, C.c_uint32, C.c_int, C.c_uint8, vp, C.c_int, C.POINTER(C.c_int)]
        L.ref_rotl.restype = C.c_uint32
        L.ref_rotl.argtypes = [C.c_uint32, C.c_uint32, C.c_uint32]
        L.ref_hamming_encode_soft.restype = C.c_uint8
        L.ref_hamming_encode_soft.argtypes = [C.c_uint8]
        L.ref_select_bits.restype = C.c_uint32
        L.ref_select_bits.argtypes = [C.c_uint32, vp, C.c_uint8]
        L.ref_swap_nibbles.argtypes = [vp, C.c_uint32]
        L.ref_build_packet.restype = C.c_uint32
        L.ref_build_packet.argtypes = [vp, C.c_uint32, vp, C.c_uint32]
        L.ref_hamming84_decode_stub.restype = C.c_uint8
        L.ref_hamming84_decode_stub.argtypes = [C.c_uint8]
        L.ref_prng.restype = C.POINTER(C.c_uint8)
        L.ref_prng.argtypes = [C.c_int, C.POINTER(C.c_size_t)]
        L.ref_sizeof.restype = C.c_int
        L.ref_sizeof.argtypes = [C.c_int]
    return L


def _iq(a) -> np.ndarray:
    return np.ascontiguousarray(a, dtype=np.complex64)


class Reference:
    """One gr::lora::decoder_impl instance (constructor arguments of decoder::make, include/lora/decoder.h:705),
    driven by the scheduler loop of oracle/ref_build/ref_driver.cc."""

    def __init__(self, samp_rate=1e6, bandwidth=125000, sf=7, implicit=False, cr=4, crc=True,
                 reduced_rate=False, disable_drift_correction=False, simd=False, fast=False):
        self.L = lib_fast() if fast else lib(simd)
        self.h = self.L.ref_create(samp_rate, int(bandwidth), int(sf), int(implicit), int(cr), int(crc),
                                   int(reduced_rate), int(disable_drift_correction))
        if not self.h:
            raise ValueError("reference: constructor would exit(1)")
        self.sps = self.L.ref_sps(self.h)
        self.nbins = self.L.ref_bins(self.h)

    def __del__(self):
        if getattr(self, "h", None):
            self.L.ref_destroy(self.h)
            self.h = None

    def run(self, iq) -> int:
        a = _iq(iq)
        return self.L.ref_run(self.h, a.ctypes.data, a.size)

    def frames(self) -> List[bytes]:
        out = []
        for i in range(self.L.ref_num_frames(self.h)):
            n = self.L.ref_get_frame(self.h, i, None, 0)
            buf = (C.c_uint8 * n)()
            self.L.ref_get_frame(self.h, i, buf, n)
            out.append(bytes(buf))
        return out

    def frame_positions(self) -> List[int]:
        return [self.L.ref_frame_pos(self.h, i) for i in range(self.L.ref_num_frames(self.h))]

    def clear(self):
        self.L.ref_clear_frames(self.h)

    def enable_trace(self, on=True):
        self.L.ref_enable_trace(self.h, int(on))

    def trace(self):
        p = C.POINTER(Step)()
        n = self.L.ref_trace(self.h, C.byref(p))
        return [(p[i].state, p[i].pos, p[i].consumed, p[i].bin, p[i].fine, p[i].value) for i in range(n)]

    def stdout(self) -> str:
        n = self.L.ref_stdout(self.h, None, 0)
        buf = C.create_string_buffer(n + 1)
        self.L.ref_stdout(self.h, buf, n + 1)
        return buf.value.decode("latin1")

    def table(self, which: int) -> np.ndarray:
        n = C.c_size_t()
        p = self.L.ref_table(self.h, which, C.byref(n))
        return np.ctypeslib.as_array(p, shape=(n.value,)).copy()

    def get_shift_fft(self, iq) -> int:
        a = _iq(iq); assert a.size >= self.sps
        return self.L.ref_get_shift_fft(self.h, a.ctypes.data)

    def experimental_determine_cfo(self, iq) -> float:
        a = _iq(iq); assert a.size >= self.sps
        return self.L.ref_experimental_determine_cfo(self.h, a.ctypes.data, self.sps)

    def max_frequency_gradient_idx(self, iq) -> int:
        a = _iq(iq); assert a.size >= self.sps
        return self.L.ref_max_frequency_gradient_idx(self.h, a.ctypes.data)

    def fine_sync(self, iq, bin_idx: int, search: int) -> int:
        a = _iq(iq); assert a.size >= self.sps
        return self.L.ref_fine_sync(self.h, a.ctypes.data, bin_idx, search)

    def detect_preamble_autocorr(self, iq) -> float:
        a = _iq(iq); assert a.size >= 2 * self.sps
        return self.L.ref_detect_preamble_autocorr(self.h, a.ctypes.data)

    def pwr_queue(self) -> List[float]:
        buf = (C.c_float * 4)()
        n = self.L.ref_pwr_queue(self.h, buf)
        return [buf[i] for i in range(n)]

    def energy_threshold(self) -> float:
        return self.L.ref_energy_threshold(self.h)

    def determine_energy(self, iq) -> float:
        a = _iq(iq); assert a.size >= self.sps
        return self.L.ref_determine_energy(self.h, a.ctypes.data)

    def detect_downchirp(self, iq) -> float:
        a = _iq(iq); assert a.size >= self.sps
        return self.L.ref_detect_downchirp(self.h, a.ctypes.data)

    def detect_upchirp(self, iq):
        a = _iq(iq); assert a.size >= 2 * self.sps
        idx = C.c_int32(0)
        c = self.L.ref_detect_upchirp(self.h, a.ctypes.data, C.byref(idx))
        return c, idx.value

    def instantaneous_frequency(self, iq) -> np.ndarray:
        a = _iq(iq)
        out = np.zeros(a.size, dtype=np.float32)
        self.L.ref_instantaneous_frequency(self.h, a.ctypes.data, out.ctypes.data, a.size)
        return out

    def deinterleave(self, words, ppm: int) -> bytes:
        w = np.ascontiguousarray(words, dtype=np.uint32)
        out = (C.c_uint8 * ppm)()
        self.L.ref_deinterleave(self.h, w.ctypes.data, w.size, ppm, out)
        return bytes(out)

    def decode(self, codewords: bytes, is_header: bool, cr: int):
        src = (C.c_uint8 * max(1, len(codewords))).from_buffer_copy(bytes(codewords) or b"\0")
        out = (C.c_uint8 * 1024)()
        left = C.c_int(0)
        n = self.L.ref_decode(self.h, src, len(codewords), int(is_header), cr, out, 1024, C.byref(left))
        return bytes(out[:n]), left.value


def prng(which: int) -> bytes:
    n = C.c_size_t()
    p = lib().ref_prng(which, C.byref(n))
    return bytes(p[i] for i in range(n.value))


def decode_stream(iq, **cfg) -> List[bytes]:
    r = Reference(**cfg)
    r.run(iq)
    return r.frames()
