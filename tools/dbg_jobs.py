import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from gr_lora_amd import capi, synth
from oracle import oracle as O
cfg = synth.TxConfig(sf=7, cr=4)
for seg in [int(a) for a in sys.argv[1:]]:
    rng = np.random.default_rng(4242 + seg)
    payloads = [bytes(rng.integers(0, 256, int(rng.integers(1, 48)), dtype=np.uint8)) for _ in range(40)]
    gaps = [int(g) for g in rng.integers(0, 7 * cfg.sps, len(payloads))]
    gaps[5] = 0; gaps[6] = 1; gaps[7] = cfg.sps // 2; gaps[8] = 2 * cfg.sps + 3
    st = synth.build_stream(payloads, cfg, gaps=gaps)
    d = torch.from_numpy(st.iq.view(np.float32)).cuda()
    for demod in (2, 0):
        o = O.Oracle(sf=7, cr=4, demod=demod); o.run(st.iq); want = o.frames()
        for rep in range(4):
            h = capi.Handle(sf=7, cr=4, demod=demod, segment_symbols=seg)
            h.decode_device(d.data_ptr(), st.iq.size, [0], [st.iq.size], 0)
            got = [g for g, _ in h.drain()]
            tm = h.timing()
            print("seg", seg, "demod", demod, "rep", rep, "frames", len(got), len(want), "equal", got == want, "jobs", tm.jobs, "probes", tm.probes, "slow", tm.slow_path_relaunches)
            h.close()
