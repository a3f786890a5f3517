"""The strict SYNC path (gr_lora_amd/csrc/lora_strict_sync.inc.hip) re-evaluates near-tied shifts with the reference's own
arithmetic, and the reference's std::arg (decoder_impl.cc:232-233) is libm's atan2f.  The device cannot call libm: it runs a
restatement of the algorithm glibc 2.35 uses (fdlibm's float atanf / atan2f).  This test holds the SAME restatement, in C
inside the oracle library, to the host's libm bit for bit - on whatever host runs the suite - so that "device == restatement"
(tests/test_gpu_strict_sync.py) means "device == the reference's atan2f"."""
import ctypes as C

import numpy as np
import pytest


def _libm_is_fdlibm(L):
    """glibc up to 2.3x computes atan2f with fdlibm's float algorithm; newer releases ship a correctly rounded one.  The strict SYNC
    parity claim is made against the former (include/lora_hip.h, LORA_HIP_FLAG_FAST_SYNC): on a host with another libm this test has
    nothing to say, and says so instead of failing."""
    L.lora_oracle_fd_atan2f_mismatches.restype = C.c_uint64
    L.lora_oracle_fd_atan2f_mismatches.argtypes = [C.c_uint64, C.c_uint64]
    return L.lora_oracle_fd_atan2f_mismatches(100_000, 99) == 0


def test_restated_atan2f_is_this_hosts_libm(oracle_mod):
    L = oracle_mod.lib()
    if not _libm_is_fdlibm(L):
        pytest.skip("this host's libm does not compute atan2f with the fdlibm float algorithm (glibc 2.35's): the restatement is pinned to that one")
    L.lora_oracle_fd_atan2f_mismatches.restype = C.c_uint64
    L.lora_oracle_fd_atan2f_mismatches.argtypes = [C.c_uint64, C.c_uint64]
    for seed in (0, 1, 2):
        assert L.lora_oracle_fd_atan2f_mismatches(10_000_000, seed) == 0


def test_restated_atan2f_special_values(oracle_mod):
    L = oracle_mod.lib()
    if not _libm_is_fdlibm(L):
        pytest.skip("this host's libm does not compute atan2f with the fdlibm float algorithm")
    L.lora_oracle_fd_atan2f.restype = C.c_float
    L.lora_oracle_fd_atan2f.argtypes = [C.c_float, C.c_float]
    libm = C.CDLL("libm.so.6")
    libm.atan2f.restype = C.c_float
    libm.atan2f.argtypes = [C.c_float, C.c_float]
    inf, nan = float("inf"), float("nan")
    vals = [0.0, -0.0, 1.0, -1.0, inf, -inf, 1e-45, -1e-45, 1e-30, 3e38, 0.4375, 0.6875, 1.1875, 2.4375, 2.0 ** -29, 2.0 ** 25, 2.0 ** 61]
    for y in vals:
        for x in vals:
            a, b = np.float32(libm.atan2f(y, x)), np.float32(L.lora_oracle_fd_atan2f(y, x))
            assert a.view(np.uint32) == b.view(np.uint32), (y, x, a, b)
    assert np.isnan(L.lora_oracle_fd_atan2f(nan, 1.0)) and np.isnan(L.lora_oracle_fd_atan2f(1.0, nan))
