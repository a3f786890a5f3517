"""An exact parallel form of detect_upchirp's sequential float sum as a C model (tests/host_sim/par_sum_model.c; the device version was exact and slower
than the single adding lane the kernels keep: profiles/r04_ab_wave_parallel_exact_sum.txt): bit-equal to the plain loop on random sequences (every sign / magnitude mix, values built to sit exactly half way
between two representable sums, cancellation down through binades) and on real product sequences - instantaneous frequency of synthetic preambles
times the ideal upchirp's, as the SYNC re-evaluation forms them."""
import ctypes as C
import os
import subprocess

import numpy as np
import pytest

from gr_lora_amd import synth

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "tests", "host_sim", "par_sum_model.c")
LIB = os.path.join(ROOT, "tests", "host_sim", "libpar_sum_model.so")


@pytest.fixture(scope="module")
def model():
    if not os.path.exists(LIB) or os.path.getmtime(SRC) > os.path.getmtime(LIB):
        subprocess.check_call(["gcc", "-O2", "-ffp-contract=off", "-fPIC", "-shared", "-o", LIB, SRC, "-lm"])
    L = C.CDLL(LIB)
    L.par_sum.restype = C.c_float
    L.par_sum.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_void_p]
    L.par_sum_seq.restype = C.c_float
    L.par_sum_seq.argtypes = [C.c_void_p, C.c_int, C.c_float]
    return L


def _both(L, p, W):
    p = np.ascontiguousarray(p, dtype=np.float32)
    steps = C.c_int(0)
    a = L.par_sum(p.ctypes.data, p.size, W, C.byref(steps))
    b = L.par_sum_seq(p.ctypes.data, p.size, 0.0)
    return np.float32(a), np.float32(b), steps.value


def test_random_sequences(model):
    rng = np.random.default_rng(1)
    n_steps = []
    for trial in range(3000):
        n = int(rng.integers(1, 5000))
        kind = trial % 6
        if kind == 0:
            p = rng.uniform(0.0, 1.0, n)                      # a growing sum: the preamble case
        elif kind == 1:
            p = rng.normal(0.0, 1.0, n)                       # cancellation: the sum wanders through binades and signs
        elif kind == 2:
            p = rng.uniform(0.5, 1.0, n) * rng.choice([1.0, -0.25], n)
        elif kind == 3:
            p = np.round(rng.uniform(0, 4, n) * 4) / 4        # many exact half-way cases
        elif kind == 4:
            p = rng.uniform(0, 1, n) * 10.0 ** rng.uniform(-12, 6, n)
        else:
            p = np.concatenate([rng.uniform(0.9, 1.0, n), -rng.uniform(0.9, 1.0, n)])  # up, then all the way back down
        for W in (2048, 1024, 896):
            a, b, st = _both(model, p, W)
            assert a.tobytes() == b.tobytes(), (trial, kind, n, W, a, b)
            n_steps.append(st)
    assert max(n_steps) < 4000


def test_half_way_cases_by_construction(model):
    """s = 2^23 .. 2^24 in units of 1: adding k + 0.5 exactly is a tie every time; both roundings (up and down to even) occur"""
    for start in (8388608.0, 8388609.0, 12345678.0, 16777214.0):
        for inc in (0.5, 1.5, -0.5, 2.5, -1.5):
            p = np.array([start] + [inc] * 200, dtype=np.float32)
            a, b, _ = _both(model, p, 2048)
            assert a.tobytes() == b.tobytes(), (start, inc, a, b)


@pytest.mark.parametrize("sf", [7, 9, 11])
def test_real_product_sequences(model, oracle_mod, sf):
    """ifreq of a (noisy) preamble window times d_upchirp_ifreq, for the shifts around the chirp boundary - what strict::resolve adds up"""
    cfg = synth.TxConfig(sf=sf, cr=4)
    rng = np.random.default_rng(sf)
    st = synth.build_stream([b"0123456789abcdef"], cfg, rng=rng, noise_sigma=synth.awgn_sigma_for_snr(36.0, cfg))
    o = oracle_mod.Oracle(sf=sf, cr=4)
    up = np.asarray(o.table(3), dtype=np.float32)   # d_upchirp_ifreq
    sps = cfg.sps
    x = st.iq[st.frame_starts[0] + 2 * sps - 37: st.frame_starts[0] + 4 * sps + 64]
    ph = np.angle(x).astype(np.float32)
    d = (ph[1:] - ph[:-1]).astype(np.float32)
    d = np.where(d > np.float32(np.pi), d - np.float32(2 * np.pi), d)
    d = np.where(d < -np.float32(np.pi), d + np.float32(2 * np.pi), d).astype(np.float32)
    for shift in range(0, 80, 3):
        p = (d[shift:shift + sps - 1] * up[:sps - 1]).astype(np.float32)
        for W in (2048, 896):
            a, b, steps = _both(model, p, W)
            assert a.tobytes() == b.tobytes(), (sf, shift, W, a, b)
    print(sf, "parallel steps for the last chain:", steps, "of", sps - 1, "taps")
