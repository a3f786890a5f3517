"""Decision-flip stress test (tools/decision_flip_sweep.py; the full sweep is profiles/r02_decision_flips.jsonl: 14 400
packets at SF7/8/9 across the 30-38 dB in-band SNR band where the reference's gates go from never to always passing).
Here a reduced sweep with the same assertions: identical frames; the device trace leaves the oracle's only at SYNC, and
only between shifts whose exact (float64) sliding correlations tie at the resolution of the reference's float sum -
no flips at the 0.90 / 0.96 / -0.97 gates (one-pass Pearson variance, product-form ifreq, polynomial atan2 included),
none in fine_sync, none in the bins."""
import os
import sys

import pytest

sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools"))

pytestmark = [pytest.mark.gpu, pytest.mark.slow]


@pytest.mark.parametrize("sf", [7, 8, 9])
@pytest.mark.parametrize("demod", [2, 0])
def test_no_gate_flips_in_the_marginal_band(oracle_mod, sf, demod):
    import decision_flip_sweep as D
    tot_diff = 0
    for snr in (32.5, 34.0, 35.5):
        r = D.run_point(sf, snr, 96, demod, seed=7)
        assert r["frames_differ"] == 0, r
        assert r["oracle_frames"] == r["device_frames"], r
        assert all(k.startswith("SYNC:") for k in r["first_diff"]), r          # nothing but the SYNC first-maximum
        assert r.get("sync_gap_rel_max", 0.0) < 1e-5, r                         # ... and only between tying shifts
        tot_diff += r["streams_with_trace_diff"]
    assert tot_diff <= 6                                                        # (0.2 % in the full sweep)
