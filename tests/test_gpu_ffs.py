"""d_fine_sync of the FFT demodulators on hostile windows - the closed-form fine_sync of wave_demod_symbol (FMODE 2: SF7 - SF9 one wavefront per window, round 6) and
whatever SF10 - SF12 run.  d_fine_sync per window against the oracle's fine_sync
(lib/decoder_impl.cc:300-338) on every kind of window tools/ffs_model.py knows - clean and noisy symbols cut early / late, noise,
downchirps, tones, interferers, carrier offsets, partial windows, bursts, clipping - i.e. on windows that take the closed form AND on
windows that must take the exact path, and the same with the closed form switched off (LORA_HIP_NO_FFS)."""
import os
import subprocess
import sys

import numpy as np
import pytest

from gr_lora_amd import synth

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def torch_cuda():
    import torch
    if not torch.cuda.is_available():
        pytest.fail("gpu-marked test needs a GPU: the HIP path has no CPU fallback")
    return torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))


def _windows(sf, n_per_kind, seed):
    import ffs_model as M
    cfg = synth.TxConfig(sf=sf, cr=4)
    up = synth.base_upchirp(cfg)
    rng = np.random.default_rng(seed)
    wins, kinds = [], []
    for snr, kind in M.KINDS:
        for _ in range(n_per_kind):
            wins.append(M.make_window(kind, snr, rng, up, cfg.sps, cfg.nbins))
            kinds.append(kind)
    return cfg, wins, kinds


def _check(sf, mode, wins, kinds, g, gf, oracle_mod):
    import ffs_model as M
    o = oracle_mod.Oracle(sf=sf)
    S, N = o.sps, 1 << sf
    V, alpha, J, tol = M.tables(o)
    vtab = o.table(4).astype(np.float64)
    n_same_bin = n_closed = 0
    for i, w in enumerate(wins):
        sres = o.get_shift_fft(w)
        if int(g[i]) != sres:
            continue            # (the arg-max of a noise window may differ in the last float bit: not this test's subject)
        n_same_bin += 1
        bin_idx = 0 if (sres == 0 and mode == 2) else (sres + N - 1) % N
        wf = o.fine_sync(w, bin_idx, 2)
        n_closed += M.fast(w, bin_idx, S, N, V, alpha, J, tol)[0] is not None
        if int(gf[i]) != wf:    # only a float near-tie of the reference's own sums may differ (as in test_wave_demod_shift_and_fine_sync_vs_oracle)
            fq = oracle_mod.instantaneous_frequency(w).astype(np.float64)
            base = (bin_idx + 1) * 8 + S
            cq = {lag: float(np.dot(fq, vtab[base + lag:base + lag + S])) for lag in (-1, 0, 1)}
            scale = max(abs(c) for c in cq.values()) + 1e-30
            got, want = (cq[-int(gf[i])] if max(cq.values()) > 0 or int(gf[i]) == 0 else None), cq[-wf]
            assert got is not None and abs(got - want) <= 2e-6 * scale, (sf, mode, kinds[i], i, sres, int(gf[i]), wf, cq)
    return n_same_bin, n_closed


@pytest.mark.parametrize("sf,n_per_kind", [(7, 120), (8, 60), (9, 30), (10, 12), (11, 6), (12, 4)])
def test_closed_form_fine_sync_equals_oracle(torch_cuda, oracle_mod, sf, n_per_kind):
    from gr_lora_amd import capi
    cfg, wins, kinds = _windows(sf, n_per_kind, seed=4000 + sf)
    x = np.concatenate(wins + [np.zeros(2 * cfg.sps, np.complex64)])
    offs = np.arange(len(wins)) * cfg.sps
    dev = torch_cuda.from_numpy(x.view(np.float32)).cuda()
    for mode in (1, 2):
        h = capi.Handle(sf=sf, demod=mode)
        g, gf = h.demod_symbols_ex_device(dev.data_ptr(), x.size, offs, mode)
        h.close()
        n_same, n_closed = _check(sf, mode, wins, kinds, g, gf, oracle_mod)
        assert n_same > 0.9 * len(wins) and n_closed > (0.15 if sf <= 8 else 0.03) * len(wins), (n_same, n_closed, len(wins))   # both paths are exercised


_NOFFS = r'''
import os, sys
import numpy as np, torch
sys.path.insert(0, os.environ["LORA_ROOT"]); sys.path.insert(0, os.path.join(os.environ["LORA_ROOT"], "tests"))
from gr_lora_amd import capi
import test_gpu_ffs as T
cfg, wins, kinds = T._windows(7, 40, seed=77)
x = np.concatenate(wins + [np.zeros(2 * cfg.sps, np.complex64)])
offs = np.arange(len(wins)) * cfg.sps
dev = torch.from_numpy(x.view(np.float32)).cuda()
h = capi.Handle(sf=7, demod=2)
g, gf = h.demod_symbols_ex_device(dev.data_ptr(), x.size, offs, 2)
h.close()
np.save(os.environ["LORA_OUT"], np.stack([g.astype(np.int64), gf.astype(np.int64)]))
'''


def test_closed_form_on_and_off_agree(tmp_path):
    """the same windows with the closed form compiled in but switched off (every window through the exact path): identical outputs"""
    outs = []
    for tag, extra in (("on", {}), ("off", {"LORA_HIP_NO_FFS": "1"})):
        env = dict(os.environ)
        env.pop("LORA_HIP_NO_FFS", None)
        env.update(LORA_ROOT=ROOT, LORA_OUT=str(tmp_path / (tag + ".npy")), **extra)
        r = subprocess.run([sys.executable, "-c", _NOFFS], env=env, capture_output=True, text=True, timeout=600)
        assert r.returncode == 0, (r.stdout[-2000:], r.stderr[-4000:])
        outs.append(np.load(tmp_path / (tag + ".npy")))
    assert np.array_equal(outs[0][0], outs[1][0])
    differ = np.flatnonzero(outs[0][1] != outs[1][1])
    assert differ.size == 0, differ[:10]
