"""Per-job timeline of one walker launch on the bench workload (LORA_HIP_JOB_TIMELINE dump): where the spread between
jobs of equal work comes from -- start skew, XCD, CU sharing."""
import os, sys, collections
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import bench
from gr_lora_amd import capi
path = "/tmp/job_timeline.txt"
os.environ["LORA_HIP_JOB_TIMELINE"] = path
cfg, iq, offs, lens, expect = bench.make_workload(7, 4, 1024, 32, 8, 2)
d = torch.from_numpy(iq.view(np.float32)).cuda()
h = capi.Handle(sf=7, cr=4, demod=2)
cu_hist = []
for i in range(4):
    if os.path.exists(path): os.remove(path)
    h.decode_device(d.data_ptr(), iq.size, offs, lens, 0); h.drain()
    r_ = np.array([list(map(int, l.split())) for l in open(path) if not l.startswith("#")], dtype=np.int64)
    k_ = (r_[:, 2] & 0xf) * 256 + ((r_[:, 1] >> 8) & 0xff)
    e_ = ((r_[:, 4] - r_[:, 3].min()) & 0xffffffff) / 100.0
    cu_hist.append({int(k): float(e_[k_ == k].max()) for k in set(k_.tolist())})
h.close()
ks = sorted(cu_hist[-1])
m = np.array([[c.get(k, np.nan) for k in ks] for c in cu_hist[1:]])
print("per-CU end time, correlation between consecutive passes: %.2f %.2f" % (np.corrcoef(m[0], m[1])[0, 1], np.corrcoef(m[1], m[2])[0, 1]))
rows = [list(map(int, l.split())) for l in open(path) if not l.startswith("#")]
a = np.array(rows, dtype=np.int64)
hw, xcc, t0, t1 = a[:, 1], a[:, 2] & 0xf, a[:, 3], a[:, 4]
dur = ((t1 - t0) & 0xffffffff) / 100.0  # us
start = ((t0 - t0.min()) & 0xffffffff) / 100.0
end = start + dur
cu = (hw >> 8) & 0xf; se = (hw >> 13) & 0x7; sh = (hw >> 12) & 1
print("jobs %d; start skew us: min %.1f p50 %.1f max %.1f; dur us: min %.1f p50 %.1f p90 %.1f max %.1f; end max %.1f" %
      (len(a), start.min(), np.median(start), start.max(), dur.min(), np.median(dur), np.percentile(dur, 90), dur.max(), end.max()))
for x in sorted(set(xcc)):
    m = xcc == x
    print("xcc %d: n %d dur mean %.1f max %.1f start mean %.1f end max %.1f" % (x, m.sum(), dur[m].mean(), dur[m].max(), start[m].mean(), end[m].max()))
key = xcc * 1000 + se * 100 + sh * 20 + cu
cnt = collections.Counter(key.tolist())
print("workgroups per CU: ", collections.Counter(cnt.values()))
for n in sorted(set(cnt.values())):
    m = np.array([cnt[k] == n for k in key.tolist()])
    print("  jobs on CUs holding %d: n %d dur mean %.1f max %.1f" % (n, m.sum(), dur[m].mean(), dur[m].max()))
loop = a[:, 5:11].copy(); loop[:, 4] = 0
lc = loop.sum(axis=1) * 64 / 2400.0
print("state-loop us (at 2.4 GHz): mean %.1f max %.1f; dur - loop: mean %.1f" % (lc.mean(), lc.max(), (dur - lc).mean()))
clk = a[:, 14] * 64.0
print("shader clock over the job: mean %.3f GHz (min %.3f max %.3f)" % ((clk / dur).mean() / 1e3, (clk / dur).min() / 1e3, (clk / dur).max() / 1e3))
print("state loop / whole job (clock64): mean %.3f" % ((loop.sum(axis=1) * 64.0) / clk).mean())
pair = collections.defaultdict(list)
for j, k in enumerate(key.tolist()): pair[k].append(j)
pp = np.array([[dur[v[0]], dur[v[1]]] for v in pair.values() if len(v) == 2])
print("correlation of the two jobs sharing a CU: %.2f" % np.corrcoef(pp[:, 0], pp[:, 1])[0, 1])
first_fast = 0; cu_end = []; gap = []
for v in pair.values():
    if len(v) != 2: continue
    a0, a1 = (v[0], v[1]) if (start[v[0]], a[v[0], 0]) <= (start[v[1]], a[v[1], 0]) else (v[1], v[0])
    first_fast += dur[a0] < dur[a1]
    cu_end.append(max(end[a0], end[a1])); gap.append(abs(end[a0] - end[a1]))
cu_end = np.array(cu_end); gap = np.array(gap)
print("pairs where the workgroup that started first is the faster one: %d of %d" % (first_fast, len(cu_end)))
print("per-CU end us: min %.1f p50 %.1f max %.1f; |end difference| within a CU: mean %.1f max %.1f" % (cu_end.min(), np.median(cu_end), cu_end.max(), gap.mean(), gap.max()))
wid = hw & 0xf; simd = (hw >> 4) & 3
print("HW wave slot of the reporting wave vs duration: ", {int(w): round(float(dur[wid == w].mean()), 1) for w in sorted(set(wid.tolist()))})
names = ["DET", "SYNC", "SFD", "PAUSE", "-", "PAYLOAD"]
srt = np.argsort(dur)
for tag, sel in (("fastest", srt[8:16]), ("slowest", srt[-8:])):
    print(tag, "mean kcycles per state:", " ".join("%s %.0f" % (names[i], loop[sel, i].mean() * 64 / 1e3) for i in (0, 1, 2, 3, 5)), "dur %.1f" % dur[sel].mean())
print("payload rounds: fastest %.1f slowest %.1f; all: min %d max %d" % (a[srt[8:16], 15].mean(), a[srt[-8:], 15].mean(), a[:, 15].min(), a[:, 15].max()))
cuk = xcc * 256 + ((hw >> 8) & 0xff)
sums = {int(k): (loop[cuk == k, 5].sum() * 64 / 1e3, a[cuk == k, 15].sum(), end[cuk == k].max()) for k in set(cuk.tolist())}
vals = np.array(list(sums.values()))
print("per CU: payload kcycles vs end time corr %.2f; payload rounds per CU min %d max %d; kcycles per payload round min %.1f p50 %.1f max %.1f" %
      (np.corrcoef(vals[:, 0], vals[:, 2])[0, 1], vals[:, 1].min(), vals[:, 1].max(), (vals[:, 0] / vals[:, 1]).min(), np.median(vals[:, 0] / vals[:, 1]), (vals[:, 0] / vals[:, 1]).max()))
for x in sorted(set(xcc)):
    sel = [k for k in sums if k // 256 == x]
    print("  xcc %d: CU end min %.1f max %.1f" % (x, min(sums[k][2] for k in sel), max(sums[k][2] for k in sel)))
o = np.argsort(-dur)[:8]
for j in o: print("  slow job %d xcc %d se %d cu %d: start %.1f dur %.1f n_att %d" % (a[j, 0], xcc[j], se[j], cu[j], start[j], dur[j], a[j, 13]))
