import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from gr_lora_amd import capi, synth
n = 1024
cfg = synth.TxConfig(sf=7, cr=4)
rng = np.random.default_rng(3)
streams, offs, lens, expect, shifts = [], [], [], [], []
off = 0
for s in range(n):
    p = bytes(rng.integers(0, 256, 32, dtype=np.uint8))
    st = synth.build_stream([p], cfg, gaps=[int(rng.integers(2, 6) * cfg.sps)], tail_symbols=2.5)
    streams.append(st.iq); offs.append(off); lens.append(st.iq.size); off += st.iq.size
    expect.append(synth.expected_frame_tail(p, cfg)); shifts.append(st.shifts[0][0] + st.shifts[0][1])
iq = np.concatenate(streams)
d = torch.from_numpy(iq.view(np.float32)).cuda()
h = capi.Handle(sf=7, cr=4, demod=2, flags=capi.FLAG_TRACE)
shown = 0
for it in range(int(sys.argv[1]) if len(sys.argv) > 1 else 30):
    h.trace_clear()
    h.decode_device(d.data_ptr(), iq.size, offs, lens, 0)
    fr = h.drain()
    got = {i.stream: b[15:] for b, i in fr}
    wrong = [s for s in range(n) if got.get(s) != expect[s]]
    if wrong and shown < 4:
        tr = h.trace()
        for s in wrong[:2]:
            steps = [t for t in tr if t[6] == s and t[0] in (4, 5)]
            bins = [t[3] for t in steps]
            want = [(x - 1) % 128 if x else 0 for x in shifts[s]]
            bad_idx = [k for k in range(min(len(bins), len(want))) if bins[k] != want[k]]
            fines = [t[4] for t in steps]
            print("iter", it, "stream", s, "n_sym", len(bins), len(want), "bad bin idx", bad_idx[:10],
                  "got", [bins[k] for k in bad_idx[:6]], "want", [want[k] for k in bad_idx[:6]], "nonzero fines", [(k, f) for k, f in enumerate(fines) if f][:6])
            shown += 1
print("done")
