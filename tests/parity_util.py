"""Trace comparison shared by the GPU parity tests.

`exact=True`: every work() step must match the oracle's - state, input position, consume_each, bin, d_fine_sync - and the
decision values to 1e-3.

`exact=False` is for decoders created with LORA_HIP_FLAG_FAST_SYNC (the closed-form SYNC maximum alone, the only behaviour up to
round 3; since round 4 the near-tie is decided with the reference's own arithmetic, tests/test_gpu_strict_sync.py, and every test
compares exactly): the operating points where the REFERENCE'S OWN timing decisions are below the resolution of its float
sums: the sliding correlation of SYNC (decoder_impl.cc:399-413) ties between adjacent shifts to 1e-6 .. 2e-8 relative
(SF8 .. SF12: the ideal upchirp's ifreq sums to ~0, so C[t0 + 1] - C[t0] = b sum(u) vanishes), and which shift wins is
decided by the summation order of VOLK's dot product.  tests/test_ref_pin.py::test_sync_shift_depends_on_volk_summation_order
shows the compiled reference disagreeing with ITSELF there when only that order changes.  In those cases the device (double-
precision closed form) may land one sample beside the oracle (sequential float sum); required is then: the same sequence
of states, every position within one sample, identical frames (checked by the caller)."""
import numpy as np


def assert_trace_parity(tr, otr, exact, tag=None):
    assert len(tr) == len(otr), (tag, len(tr), len(otr))
    off = 0
    for i, (a, b) in enumerate(zip(tr, otr)):
        if exact:
            assert tuple(a[:5]) == tuple(b[:5]), (tag, i, a, b)
            if np.isfinite(b[5]):
                assert abs(a[5] - b[5]) <= 1e-3 * max(1.0, abs(b[5])), (tag, i, a, b)
        else:
            assert a[0] == b[0], (tag, i, a, b)                 # same state sequence
            assert abs(a[1] - b[1]) <= 1, (tag, i, a, b)        # positions within one sample
            assert abs(a[2] - b[2]) <= 2, (tag, i, a, b)        # (a step that re-aligns consumes up to 2 more / fewer)
            off += a[1] != b[1]
    return off
