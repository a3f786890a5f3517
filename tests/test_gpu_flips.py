"""Decision-flip stress test (tools/decision_flip_sweep.py: packets at SF7-SF11 across the 30-38 dB in-band SNR band where the reference's gates go
from never to always passing).  Round 2 found 30 of 14 400 packets whose device trace left the oracle's, every one at SYNC between two shifts whose
correlations tie at the resolution of the reference's float sum (profiles/r02_decision_flips.jsonl).  Since round 4 those ties are decided with the
reference's own arithmetic (lora_strict_sync.inc.hip): the sweep finds NO differing trace and no differing frame (profiles/r04_decision_flips.jsonl,
4 680 packets), and this reduced sweep requires exactly that - no flips at the 0.90 / 0.96 / -0.97 gates (one-pass Pearson variance, product-form
ifreq, polynomial atan2 included), none at SYNC, none in fine_sync, none in the bins."""
import os
import sys

import pytest

sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools"))

pytestmark = [pytest.mark.gpu, pytest.mark.slow]


@pytest.mark.parametrize("sf", [7, 8, 9])
@pytest.mark.parametrize("demod", [2, 0])
def test_no_gate_flips_in_the_marginal_band(oracle_mod, sf, demod):
    import decision_flip_sweep as D
    for snr in (32.5, 34.0, 35.5):
        r = D.run_point(sf, snr, 96, demod, seed=7)
        assert r["frames_differ"] == 0, r
        assert r["oracle_frames"] == r["device_frames"], r
        assert r["streams_with_trace_diff"] == 0 and not r["first_diff"], r
