"""Decoupled passes (gr_lora_amd/csrc/lora_stitch.hpp payload_begin / payload_end; include/lora_hip.h LORA_HIP_FLAG_NO_DECOUPLED; docs/LAB_NOTEBOOK.md 4.13): the
state-machine jobs run the header-only kernel variant (walker3_kernel_sf*_skip, and since round 5 walker2_kernel_sf7/8*_skip; LaunchCfg.skip_payload) - a packet's attempt ends behind its header, the job
goes on where the payload would end had no symbol moved the symbol clock - and the payload pass demodulates every payload symbol of every packet at once
(demod_symbols_w3_kernel, second reads behind symbols that move the clock) and walks each packet's symbols through the integer chain
(payload_chain_kernel).  Required: the frames, header positions and end positions of the ordinary pass - on config-3 cells against the compiled
reference's fixtures, with a drifting transmitter clock (jobs split at packets that end off the zero-drift grid and probed from the true end; packets that
drift through more offsets than the pass reads handed to the complete kernels), with the data ending inside a payload, through the streaming entry point -
and the per-pass choice."""
import numpy as np
import pytest

import bench
from gr_lora_amd import synth

pytestmark = [pytest.mark.gpu]


def _dev(iq):
    import torch
    return torch.from_numpy(np.ascontiguousarray(iq).view(np.float32)).cuda()


def _run(iq, offs, lens, demod, monkeypatch, decoupled, **kw):
    from gr_lora_amd import capi
    monkeypatch.setenv("LORA_HIP_DECOUPLED", decoupled)
    dev = _dev(iq)
    h = capi.Handle(demod=demod, **kw)
    h.decode_device(dev.data_ptr(), iq.size, offs, lens, 0)
    out = [(g, i.stream, i.header_pos, i.end_pos) for g, i in h.drain()]
    info = dict(kernel=h.kernel_name(), t=h.timing(), **h.payload_pass())
    h.close()
    return out, info


@pytest.mark.parametrize("sf", [7, 8, 9, 10, 11, 12])
@pytest.mark.parametrize("demod", [2, 0])
def test_decoupled_equals_ordinary_pass(sf, demod, monkeypatch):
    cfg, iq, offs, lens, expect = bench.make_workload(sf, 4, 48, 32, 8, seed=40 + sf)
    kw = dict(sf=sf, cr=4, reduced_rate=(sf > 10))
    want, wi = _run(iq, offs, lens, demod, monkeypatch, "0", **kw)
    got, gi = _run(iq, offs, lens, demod, monkeypatch, "1", **kw)
    assert len(want) == 48 and wi["packets"] == 0 and not wi["kernel"].endswith("_skip")
    assert gi["kernel"].endswith("_skip"), gi
    # clean signal: a symbol with bin 0 moves the symbol clock by +1 and its successor moves it back - one more round of reads, nothing handed back
    # (SF7 / SF8, round 5: the wave-per-symbol kernels have no second reads - the +1 / -1 pair is a round per move)
    assert gi["packets"] >= 48 and gi["rerun"] == 0 and gi["moved"] == 0 and gi["rounds"] <= (2 if sf >= 9 else 3) and gi["symbols"] > 48 * 20, gi
    assert got == want
    if demod == 2:
        assert [g[0][15:] for g in got if g[1] == 0] == expect[0]
    if sf == 9:     # without the second reads every move of the symbol clock is a round of its own: the +1 / -1 pairs need three
        monkeypatch.setenv("LORA_HIP_NO_SECOND_READS", "1")
        got2, g2 = _run(iq, offs, lens, demod, monkeypatch, "1", **kw)
        assert got2 == want and g2["rounds"] >= gi["rounds"] and g2["rerun"] == 0, (gi, g2)


@pytest.mark.parametrize("sf,cr", [(7, 1), (7, 4), (8, 2), (8, 3), (9, 1), (9, 4), (10, 2), (11, 1), (11, 3), (12, 1), (12, 4)])
def test_decoupled_config3_cell_vs_reference_fixture(sf, cr, monkeypatch):
    """a whole config-3 cell (256 packets, the reference's shipped gradient demodulator) with EVERY payload through the payload pass: the compiled
    reference's frames and header positions (tests/golden/fullsize_ref.json), under the same rules as tests/test_gpu_fullsize.py"""
    import test_gpu_fullsize as F
    monkeypatch.setenv("LORA_HIP_DECOUPLED", "1")
    F._against_reference_fixture("config3-sf%d-cr%d" % (sf, cr))


def _drifting(sf, n, seed, ppm=60e-6, snr_db=42.0):
    cfg = synth.TxConfig(sf=sf, cr=4)
    rng = np.random.default_rng(seed)
    payloads = [bytes(rng.integers(0, 256, int(rng.integers(8, 40)), dtype=np.uint8)) for _ in range(n)]
    st = synth.build_stream(payloads, cfg, rng=rng, noise_sigma=synth.awgn_sigma_for_snr(snr_db, cfg))
    t = np.arange(int(st.iq.size / (1 + ppm)) - 2, dtype=np.float64) * (1 + ppm)
    i0 = t.astype(np.int64)
    fr = (t - i0).astype(np.float32)
    return cfg, (st.iq[i0] * (1 - fr) + st.iq[i0 + 1] * fr).astype(np.complex64)


@pytest.mark.parametrize("sf,demod", [(7, 2), (8, 0), (9, 2), (9, 0), (10, 2)])
def test_packets_that_move_the_symbol_clock_for_good(sf, demod, monkeypatch):
    """a transmitter clock 60 ppm off + noise: fine_sync moves the symbol clock inside most payloads, for good.  Packets the payload pass follows to
    their end leave their job split at the packet's true end (a probe from there decides what stands of the job's scan behind it); packets that drift
    through more offsets than the pass reads are decoded by the complete kernels.  Either way the output is the ordinary pass's, which
    tests/test_gpu_a16.py holds to the oracle's traces."""
    cfg, iq = _drifting(sf, 16, 7 + sf)
    offs, lens = [0], [iq.size]
    want, _ = _run(iq, offs, lens, demod, monkeypatch, "0", sf=sf, cr=4)
    got, gi = _run(iq, offs, lens, demod, monkeypatch, "1", sf=sf, cr=4)
    assert len(want) >= 12
    assert gi["packets"] > 0 and gi["moved"] + gi["rerun"] > 0 and gi["rounds"] >= 2, gi
    assert got == want


def test_data_ending_inside_a_payload_and_streaming(monkeypatch, oracle_mod):
    """the pass ends inside a payload: the packet stays pending exactly as in the ordinary pass (same frames from lora_hip_work over the same chunks)"""
    from gr_lora_amd import capi
    cfg, iq, offs, lens, expect = bench.make_workload(9, 4, 12, 32, 1, seed=5)
    cut = iq[: iq.size - 25 * cfg.sps]
    want, _ = _run(cut, [0], [cut.size], 2, monkeypatch, "0", sf=9, cr=4)
    got, gi = _run(cut, [0], [cut.size], 2, monkeypatch, "1", sf=9, cr=4)
    assert got == want and len(want) == 11 and gi["packets"] >= 12 and gi["rerun"] == 0, gi
    res = {}
    for mode in ("0", "1"):
        monkeypatch.setenv("LORA_HIP_DECOUPLED", mode)
        h = capi.Handle(sf=9, cr=4, demod=2, batch_items=40 * cfg.sps * 4)
        for k in range(0, iq.size, 77777):
            h.work(iq[k:k + 77777])
        h.flush()
        res[mode] = [(g, i.header_pos, i.end_pos) for g, i in h.drain()]
        h.close()
    o = oracle_mod.Oracle(sf=9, cr=4, demod=2)
    o.run(iq)
    assert res["1"] == res["0"] and [r[0] for r in res["1"]] == o.frames()


@pytest.mark.parametrize("sf", [7, 8, 9, 10, 11, 12])
def test_ordinary_kernels_on_small_workloads(oracle_mod, sf, monkeypatch):
    """Small workloads run decoupled when the choice is left to the library - so the suite's small SF9-SF12 cases exercise the header-only kernels and
    the payload pass.  The complete kernels' payload rounds on the same cases: the SF x CR sweep of tests/test_gpu_configs.py with LORA_HIP_DECOUPLED=0."""
    import test_gpu_configs as G
    monkeypatch.setenv("LORA_HIP_DECOUPLED", "0")
    G.test_config3_sf_cr_sweep(oracle_mod, sf)


def test_per_pass_choice(monkeypatch):
    """auto (no LORA_HIP_DECOUPLED): a gateway's pass - 8 continuous channels x 2 s of SF9 - runs decoupled, and so does a sparse bursty one; a config-3
    cell (256 packets: the balanced plan, a job of whole packets per CU) does not; LORA_HIP_FLAG_NO_DECOUPLED keeps the ordinary pass"""
    from gr_lora_amd import capi
    monkeypatch.delenv("LORA_HIP_DECOUPLED", raising=False)
    cfg, iq, offs, lens, expect = bench.make_gateway_workload(list(range(8)), 2.0, 9)
    dev = _dev(iq)
    outs = []
    for flags in (0, capi.FLAG_NO_DECOUPLED):
        h = capi.Handle(sf=9, cr=4, demod=2, flags=flags)
        h.decode_device(dev.data_ptr(), iq.size, offs, lens, 0)
        fr = [(g, i.stream, i.header_pos, i.end_pos) for g, i in h.drain()]
        pp, name = h.payload_pass(), h.kernel_name()
        h.close()
        outs.append(fr)
        if flags == 0:
            assert pp["packets"] >= len(fr) - 8 and name.endswith("_skip"), (pp, name)
        else:
            assert pp["packets"] == 0 and not name.endswith("_skip"), (pp, name)
    assert outs[0] == outs[1]
    by = {}
    for g, s, _, _ in outs[0]:
        by.setdefault(s, []).append(g[15:])
    assert [by.get(s, []) for s in range(8)] == expect
    cfg, iq, offs, lens, expect = bench.make_workload(9, 4, 256, 32, 8, seed=3)
    dev = _dev(iq)
    h = capi.Handle(sf=9, cr=4, demod=2)
    h.decode_device(dev.data_ptr(), iq.size, offs, lens, 0)
    assert h.payload_pass()["packets"] == 0 and len(h.drain()) == 256
    h.close()
