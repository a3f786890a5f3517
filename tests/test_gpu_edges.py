"""Edge cases of the decode path on the GPU, each against the oracle on the same input: empty and too-short streams inside a
batch, the smallest and the largest frames (0, 1 and 255 payload bytes; the reference's own `decode_long` suite is the
255-byte frame, apps/generate_test_suites.py:157-166), a packet cut off by the end of the data, two transmissions that
collide, silence, noise only, and non-finite samples between packets.  Frames bit-exact and in order; positions exact."""
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


def _gpu(torch, iq, streams, **kw):
    from gr_lora_amd import capi
    h = capi.Handle(**kw)
    dev = torch.from_numpy(np.ascontiguousarray(iq).view(np.float32)).to("cuda:0")
    h.decode_device(dev.data_ptr(), iq.size, [s[0] for s in streams], [s[1] for s in streams], torch.cuda.current_stream().cuda_stream)
    out = h.drain()
    h.close()
    per = {}
    for blob, info in out:
        per.setdefault(info.stream, []).append((blob, info.header_pos))
    return per


def _oracle(O, iq, streams, **kw):
    per = {}
    for i, (off, n) in enumerate(streams):
        o = O.Oracle(**kw)
        o.run(iq[off:off + n])
        fr, pos = o.frames(), o.frame_positions()
        if fr:
            per[i] = list(zip(fr, pos))
    return per


def _same(got, want, pos_tol=0):
    """frames bit-exact and in order; header positions exact (pos_tol: only for runs with LORA_HIP_FLAG_FAST_SYNC, tests/parity_util.py)"""
    assert sorted(got) == sorted(want), (sorted(got), sorted(want))
    for s in want:
        assert [b for b, _ in got[s]] == [b for b, _ in want[s]], s
        assert len(got[s]) == len(want[s]) and all(abs(p - q) <= pos_tol for (_, p), (_, q) in zip(got[s], want[s])), (s, [p for _, p in got[s]], [p for _, p in want[s]])


@pytest.mark.parametrize("sf,demod", [(7, 2), (7, 0), (9, 2)])
def test_empty_and_too_short_streams_in_a_batch(torch_cuda, oracle_mod, sf, demod):
    cfg = synth.TxConfig(sf=sf, cr=4)
    rng = np.random.default_rng(40 + sf)
    good = [synth.build_stream([bytes(rng.integers(0, 256, 6, dtype=np.uint8)) for _ in range(2)], cfg, rng=rng).iq for _ in range(3)]
    sps = cfg.sps
    # stream lengths: 0, 1, just under / exactly the block's output multiple (2 sps, decoder_impl.cc:91), and three real ones
    pieces = [np.zeros(0, np.complex64), good[0], np.ones(1, np.complex64), good[1][:2 * sps - 1], good[1], good[2][:2 * sps], good[2]]
    streams, off = [], 0
    for p in pieces:
        streams.append((off, p.size)); off += p.size
    iq = np.concatenate(pieces).astype(np.complex64)
    kw = dict(sf=sf, cr=4, demod=demod)
    got, want = _gpu(torch_cuda, iq, streams, **kw), _oracle(oracle_mod, iq, streams, **kw)
    _same(got, want)
    assert sorted(want) == [1, 4, 6] and all(len(v) == 2 for v in want.values())


@pytest.mark.parametrize("sf,cr,n", [(7, 4, 255), (7, 1, 255), (12, 1, 255), (10, 3, 255), (7, 4, 0), (8, 2, 1), (11, 4, 0)])
def test_smallest_and_largest_frames(torch_cuda, oracle_mod, sf, cr, n):
    cfg = synth.TxConfig(sf=sf, cr=cr, reduced_rate=(sf > 10))
    payload = bytes(range(255))[:n]
    iq = synth.build_stream([payload, payload], cfg, rng=np.random.default_rng(n + sf)).iq
    kw = dict(sf=sf, cr=cr, reduced_rate=(sf > 10), demod=2)
    got, want = _gpu(torch_cuda, iq, [(0, iq.size)], **kw), _oracle(oracle_mod, iq, [(0, iq.size)], **kw)
    _same(got, want)
    assert len(want[0]) == 2 and all(len(b) == 18 + n + 2 for b, _ in want[0])
    # 257 bytes at CR 4/5 and SF7 run past the reference's 516-entry de-whitening table (lib/tables.h; read out of bounds
    # upstream at decoder_impl.cc:643, pinned to "no whitening" in the oracle): the tail of that frame is not the payload
    if (sf, cr, n) != (7, 1, 255):
        assert all(b[18:18 + n] == payload for b, _ in want[0])


@pytest.mark.parametrize("sf", [7, 9])
def test_packet_cut_off_by_the_end_of_the_data(torch_cuda, oracle_mod, sf):
    cfg = synth.TxConfig(sf=sf, cr=4)
    st = synth.build_stream([b"first one", bytes(range(40))], cfg, rng=np.random.default_rng(sf))
    kw = dict(sf=sf, cr=4, demod=2)
    for cut_symbols in (3.5, 14.0, 30.25):   # inside the second packet's preamble, header, payload
        n = st.frame_starts[1] + int(cut_symbols * cfg.sps)
        iq = st.iq[:n]
        got, want = _gpu(torch_cuda, iq, [(0, n)], **kw), _oracle(oracle_mod, iq, [(0, n)], **kw)
        _same(got, want)
        assert len(want[0]) == 1


@pytest.mark.parametrize("sf", [7, 10])
def test_colliding_transmissions(torch_cuda, oracle_mod, sf):
    """A second, weaker transmission starts in the middle of the first one's payload: whatever the reference makes of it."""
    cfg = synth.TxConfig(sf=sf, cr=4)
    a = synth.build_stream([bytes(range(30))], cfg, gaps=[3 * cfg.sps], tail_symbols=40.0).iq
    b = synth.build_stream([b"collider"], cfg, gaps=[0], tail_symbols=3.0).iq
    iq = a.copy()
    start = 3 * cfg.sps + int(31.3 * cfg.sps)
    m = min(b.size, iq.size - start)
    iq[start:start + m] += (0.5 * b[:m]).astype(np.complex64)
    kw = dict(sf=sf, cr=4, demod=2)
    _same(_gpu(torch_cuda, iq, [(0, iq.size)], **kw), _oracle(oracle_mod, iq, [(0, iq.size)], **kw))


def test_silence_noise_and_non_finite_samples(torch_cuda, oracle_mod):
    cfg = synth.TxConfig(sf=7, cr=4)
    rng = np.random.default_rng(3)
    kw = dict(sf=7, cr=4, demod=2)
    silence = np.zeros(50 * cfg.sps, np.complex64)
    noise = (rng.standard_normal(80 * cfg.sps) + 1j * rng.standard_normal(80 * cfg.sps)).astype(np.complex64)
    for x in (silence, noise):
        _same(_gpu(torch_cuda, x, [(0, x.size)], **kw), _oracle(oracle_mod, x, [(0, x.size)], **kw))
    st = synth.build_stream([b"before", b"after!"], cfg, gaps=[2 * cfg.sps, 12 * cfg.sps])
    iq = st.iq.copy()
    g0 = st.frame_starts[1] - 9 * cfg.sps       # inside the gap between the packets
    iq[g0:g0 + 64] = np.nan
    iq[g0 + 2 * cfg.sps:g0 + 2 * cfg.sps + 64] = np.inf
    got = _gpu(torch_cuda, iq, [(0, iq.size)], **kw)                     # must terminate; the packets around the burst are intact
    assert [b[18:24] for b, _ in got[0]] == [b"before", b"after!"]
    want = _oracle(oracle_mod, iq, [(0, iq.size)], **kw)
    assert [b for b, _ in got[0]] == [b for b, _ in want[0]]
