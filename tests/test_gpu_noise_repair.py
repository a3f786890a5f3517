"""Noise over the stream, idle gaps included (round 6).  With any noise detect_upchirp's tie between adjacent shifts (decoder_impl.cc:392-413) is
decided by the noise, differently for two DETECT alignments: at about every second cut of the speculative segments the job's header and the true trajectory's are ONE
sample apart.  The stitch refuses such a job (the output is the serial decoder's, bit for bit) - and repairs the cut in the probe launch (the rest of the segment from
the true header; walker3's early-stopped probes walk the segment) instead of walking it serially: frames and header positions against the oracle, and the serial
fall-backs (lora_hip_timing_t.slow_path_relaunches) stay a handful."""
import concurrent.futures as cf

import numpy as np
import pytest

from gr_lora_amd import synth

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("sf,demod,per_stream", [(7, 2, 48), (8, 0, 32), (9, 2, 12), (10, 1, 8)])
def test_noisy_segments_are_repaired_not_walked_serially(oracle_mod, sf, demod, per_stream):
    import torch
    from gr_lora_amd import capi
    cfg = synth.TxConfig(sf=sf, cr=4)
    rng = np.random.default_rng(11000 + sf)
    pieces, offs, lens = [], [], []
    off = 0
    for s in range(8):
        payloads = [bytes(rng.integers(0, 256, 24, dtype=np.uint8)) for _ in range(per_stream)]
        st = synth.build_stream(payloads, cfg, rng=rng, gap_symbols=(2.0, 6.0), noise_sigma=synth.awgn_sigma_for_snr(45.0, cfg))
        pieces.append(st.iq); offs.append(off); lens.append(st.iq.size); off += st.iq.size

    def ora(k):
        o = oracle_mod.Oracle(sf=sf, cr=4, demod=demod)
        o.run(pieces[k])
        return o.frames(), o.frame_positions()
    with cf.ThreadPoolExecutor(8) as ex:
        want = list(ex.map(ora, range(8)))
    iq = np.concatenate(pieces)
    dev = torch.from_numpy(iq.view(np.float32)).cuda()
    h = capi.Handle(sf=sf, cr=4, demod=demod)
    h.decode_device(dev.data_ptr(), iq.size, offs, lens, torch.cuda.current_stream().cuda_stream)
    got, tm = h.drain(), h.timing()
    h.close()
    by_stream, pos_by_stream = {}, {}
    for g, i in got:
        by_stream.setdefault(i.stream, []).append(g)
        pos_by_stream.setdefault(i.stream, []).append(i.header_pos)
    for s in range(8):
        assert by_stream.get(s, []) == want[s][0], s
        assert pos_by_stream.get(s, []) == want[s][1], s
    assert sum(len(w[0]) for w in want) >= 6 * per_stream
    assert tm.jobs >= 4 * 8, tm.jobs                      # (the streams were cut into speculation segments)
    # (without the repairs: about one serial walk per second cut.  walker3's early-stopped probes walk the target segment from the true state; where that trajectory
    # does not END on the speculative job's sample the next cut still takes the serial path - a few per cent of the cuts)
    assert tm.slow_path_relaunches <= max(4, tm.jobs // 8), (tm.jobs, tm.probes, tm.slow_path_relaunches)
