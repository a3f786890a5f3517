"""The closed-form fine_sync rule (tools/ffs_model.py = wave_demod_symbol FMODE 2, docs/LAB_NOTEBOOK.md 5.4) against the oracle's fine_sync on CPU: wherever the rule
claims a decision it is the reference's; the table constants are those lora_hip_create derives."""
import os
import sys

import numpy as np
import pytest

from gr_lora_amd import synth

sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools"))


@pytest.mark.parametrize("sf", [7, 8, 9, 10, 11])
def test_rule_never_contradicts_the_oracle(oracle_mod, sf):
    import ffs_model as M
    o = oracle_mod.Oracle(sf=sf)
    S, N = o.sps, 1 << sf
    V, alpha, J, tol = M.tables(o)
    assert abs(alpha - np.pi / 4 / S) < 1e-6 * alpha + 1e-9 and abs(J + np.pi / 4) < 1e-3 and tol < {7: 0.02, 8: 0.08}.get(sf, 0.2)
    up = synth.base_upchirp(synth.TxConfig(sf=sf, cr=4))
    rng = np.random.default_rng(900 + sf)
    n_closed = 0
    for snr, kind in M.KINDS:
        for _ in range({7: 40, 8: 20, 9: 10}.get(sf, 5)):
            w = M.make_window(kind, snr, rng, up, S, N)
            bin_idx = (o.get_shift_fft(w) + N - 1) % N
            lag, _why = M.fast(w, bin_idx, S, N, V, alpha, J, tol)
            if lag is not None:
                n_closed += 1
                assert lag == -o.fine_sync(w, bin_idx, 2), (sf, kind, snr)
    assert n_closed > {7: 150, 8: 70, 9: 12}.get(sf, 5), n_closed   # (the aligned clean windows of every SF, a third of the rest at SF7 / SF8)
