"""the explicit CFO estimate - decoder_impl::experimental_determine_cfo (lib/decoder_impl.cc:730-738), dead code
upstream (its call in SYNC and the ("cfo", value) message to the channeliser are commented out, :774-776).  The oracle
restates it and is pinned bit-for-bit against the compiled reference (tests/test_ref_pin.py::test_cfo_estimate_vs_reference);
here the device kernel is compared with the oracle, the mean-over-the-window variant is checked for what it is for, and the
receiver mirror closes the loop upstream sketched: decoder "control" port -> channelizer.apply_cfo."""
import numpy as np
import pytest

from gr_lora_amd import capi, lora, synth
from oracle import oracle

pytestmark = pytest.mark.gpu


def _windows(sf, cfos, noise, seed):
    cfg = synth.TxConfig(sf=sf)
    up = synth.base_upchirp(cfg)
    rng = np.random.default_rng(seed)
    n = np.arange(up.size)
    ws = []
    for c in cfos:
        x = up * np.exp(2j * np.pi * c * n / 1e6 + 1j * rng.uniform(0, 2 * np.pi))
        x = x + noise * (rng.standard_normal(x.size) + 1j * rng.standard_normal(x.size))
        ws.append(x.astype(np.complex64))
    return cfg, ws


@pytest.mark.parametrize("sf", [7, 9, 12])
def test_cfo_kernel_vs_oracle(sf):
    import torch
    cfos = [0.0, 811.0, -2500.0, 4000.0, -7000.0]
    cfg, ws = _windows(sf, cfos, 0.05, sf)
    iq = np.concatenate(ws)
    offs = [i * cfg.sps for i in range(len(ws))]
    d = torch.from_numpy(iq.view(np.float32)).cuda()
    h = capi.Handle(sf=sf)
    o = oracle.Oracle(sf=sf)
    for mode in (0, 1):
        got = h.estimate_cfo_device(d.data_ptr(), iq.size, offs, mode=mode)
        want = np.array([o.determine_cfo(w, mode) for w in ws], dtype=np.float32)
        # float tolerance: the device's atan2f differs from libm's by an ulp or two of a phase, 1e-6 rad = 0.2 Hz at fs = 1 MHz
        assert np.max(np.abs(got - want)) <= 1.0, (sf, mode, got, want)
        if mode == 1:   # the mean recovers the offset; upstream's single sample does not survive noise
            assert np.max(np.abs(got - np.array(cfos))) <= 40.0, (sf, got)
    with pytest.raises(capi.LoraHipError):
        h.estimate_cfo_device(d.data_ptr(), iq.size, [iq.size - cfg.sps + 1], mode=1)
    h.close()


def test_receiver_applies_the_estimate_to_the_channeliser():
    """Packets 2.2 kHz (2.25 bins at SF7) off the channel centre through channeliser + decoder with cfo_correction.  The
    first packet is demodulated as it comes (at a quarter-bin residue its bytes may be wrong: that is what the correction is
    for); its preamble gives the estimate, the channeliser is retuned, and the packets behind it decode with what is left."""
    cfg = synth.TxConfig(sf=7, cr=4)
    payloads = [bytes([i] * 9) for i in (1, 2, 3, 4)]
    st = synth.build_stream(payloads, cfg, gaps=[60 * cfg.sps] * 4, tail_symbols=60.0)
    n = np.arange(st.iq.size)
    cfo = 2200.0
    x = (st.iq * np.exp(2j * np.pi * (100e3 + cfo) * n / 1e6)).astype(np.complex64)   # channel at +100 kHz from the capture centre
    rx = lora.lora_receiver(1e6, 868.0e6, [868.1e6], 125000, 7, False, 4, True, cfo_correction=True, verbose=False, batch_items=1 << 16)
    got = []
    rx.subscribe("frames", lambda b: got.append(bytes(b)))
    step = 1 << 16
    for i in range(0, x.size, step):
        rx.work(x[i:i + step])
    rx.stop()
    tails = [g[18:18 + g[15]] for g in got]
    assert len(got) == 4, (len(got), rx.cfo_log)
    # estimates taken on samples that were filtered before the previous correction are dropped: what is applied converges
    assert 1 <= len(rx.cfo_log) <= 3 and abs(rx.cfo_log[0] - cfo) < 200.0, rx.cfo_log   # one sample of timing error is 122 Hz at SF7
    assert abs(sum(rx.cfo_log) - cfo) < 200.0, rx.cfo_log
    assert tails[2:] == payloads[2:], (tails, rx.cfo_log)      # (the second packet is in flight when the first estimate lands)
