// lora_stitch.hpp -- the speculation scheduler: segments, probes and the stitch, independent of HIP.
//
// decode_streams<Env>() is instantiated twice: by lora_runtime.cpp over the real device (jobs run by the
// walker kernels) and by tests/host_sim/stitch_sim.cpp over a CPU environment whose jobs are run by the
// oracle's state machine -- so the scheduling logic is exercised by the CPU test-suite as well.
//
// Env must provide:  uint32_t sps(), ctor_cr(), segment_symbols(), resident_slots();  bool tracing(), implicit();
//   int  run_jobs(const std::vector<Job> &, uint32_t recs_per_job, uint32_t trace_cap, RunOut &);   (0 = ok; launch and wait)
//   int  run_jobs_begin(same) / run_jobs_end(RunOut &);   (the main launch of a pass, split: launch / wait and fetch)
//   RunOut &run_out(int which);   (two reusable result holders)
//   bool quiet_edges(const std::vector<StreamDesc> &, std::vector<std::vector<int64_t>> &);   (gap starts per stream, false = none)
//   void publish(const AttemptRec &, StreamDesc &);   void append_trace(const RunOut &, uint32_t job, uint32_t cap, int64_t base);
//   void count_jobs(uint32_t), count_probes(uint32_t), count_slow_path(), count_repair(), note_plan(bool burst_aware, size_t n_segments);   double walker_ms();
//   uint32_t resident_slots_alt();   (a second, smaller slot count when the kernel exists in two workgroup sizes; 0: none)
//   bool early_probe();   (the jobs record their FIND_SFD entry states and their tail probes may stop behind the first one: Job.tail_stop_sfd)
//   bool decoupled(size_t n_jobs);   (run this pass's segment jobs header-only and the payloads in the symbol-parallel payload pass)
//   void set_skip_payload(bool);     (the launches that follow run the kernels' header-only variant, LaunchCfg.skip_payload)
//   void abort_payload();   (an error return between the two calls below: wait for whatever the payload pass has in flight and forget it)
//   int  run_payload_begin(std::vector<PayloadReq> &) / run_payload_end(same);   (0 = ok; launch / wait: status, end_shift and frame of every request filled in)   void count_payload(uint32_t packets, uint32_t moved, uint32_t rerun);
#pragma once
#include <algorithm>
#include <chrono>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <memory>
#include <vector>

#include "lora_device.h"

namespace lora_hip {

// d_pwr_queue (boost::circular_buffer<float>(4)) + d_snr, kept on the host: the
// device reports the values each DETECT step pushed (:360), the host replays them.
struct PwrState {
    float q[4];
    int n = 0;
    float snr = 0.0f; // uninitialised upstream (decoder_impl.h:101); pinned to 0
    void push(float v)
    {
        if (n < 4) q[n++] = v;
        else { q[0] = q[1]; q[1] = q[2]; q[2] = q[3]; q[3] = v; }
    }
    void apply(uint32_t npush, const float tail[4])
    {
        const uint32_t k = npush < 4u ? npush : 4u;
        for (uint32_t i = 0; i < k; i++) push(tail[i]);
    }
    void determine_snr() // :377-383
    {
        if (n >= 2) snr = q[n - 1] / q[0];
    }
};

struct StreamDesc {
    uint64_t off, len;
    uint32_t id;
    uint32_t cr_in;       // d_phdr.cr carried in
    PwrState pwr;         // power queue / snr carried in
    int64_t  abs_base;    // added to reported positions
    // results
    int64_t  final_pos = 0;
    uint32_t cr_out = 0;
    bool incomplete = false;
};

// Attempt records of one launch: storage that is neither cleared nor reallocated between launches (a 58-deep
// std::vector<AttemptRec> per job would be megabytes to zero-fill and to page in on every call).
struct RecStore {
    std::unique_ptr<AttemptRec[]> p;
    size_t n = 0, cap = 0;
    AttemptRec *data() { return p.get(); }
    const AttemptRec *data() const { return p.get(); }
    AttemptRec &operator[](size_t i) { return p[i]; }
    const AttemptRec &operator[](size_t i) const { return p[i]; }
    size_t size() const { return n; }
    void clear() { n = 0; }
    void resize_uninit(size_t k) // contents undefined: the caller writes every element it later reads
    {
        if (k > cap) { p.reset(new AttemptRec[k + k / 4]); cap = k + k / 4; }
        n = k;
    }
    void assign(size_t k, const AttemptRec &v)
    {
        resize_uninit(k);
        for (size_t i = 0; i < k; i++) p[i] = v;
    }
};

struct RunOut {
    std::vector<JobResult> res;
    RecStore recs;
    uint32_t rpj = 0; // records stored per job (stride of `recs`): every record the jobs wrote, at most `cap`
    uint32_t cap = 0; // record capacity the jobs ran with
    const AttemptRec &rec(size_t job, uint32_t a) const { return recs[job * rpj + a]; }
    // attempts that ran to completion (the last one is pending when JobResult.pad is set)
    uint32_t n_done(size_t job) const
    {
        const JobResult &r = res[job];
        const uint32_t n = r.pad ? r.n_attempts - 1u : r.n_attempts;
        return n < cap ? n : cap;
    }
};

// One packet of the payload pass: a kAttemptHeaderOnly record of a header-only job, and what the pass made of it.
enum PayloadStatus : uint32_t {
    kPayloadDecoded = 0,   // frame[] is the serial decoder's; its symbols moved the symbol clock by end_shift samples in all (0: the job's scan behind it stands)
    kPayloadOutOfData = 1, // the data ends inside the payload (:91): the packet is pending
    kPayloadUnresolved = 2 // the pass gave up (too many moves of the symbol clock): the complete kernels decode it again from its header
};
struct PayloadReq {
    uint64_t stream_off, stream_len;
    int64_t  start;          // first payload symbol, relative to the stream (symbol k is read at start + k sps while nothing moves the symbol clock)
    int64_t  hdr_pos;        // (for an environment that decodes the packet from its header instead)
    uint32_t n_walk;         // symbols DECODE_PAYLOAD demodulates
    uint32_t cr_prev;
    SkippedPayload sk;
    // results
    uint32_t status;         // PayloadStatus
    int32_t  end_shift;
    uint32_t frame_len;
    uint8_t  frame[kMaxFrame + 4];
};

// A completed attempt of the TRUE trajectory: replay its DETECT pushes, take the
// SNR at the trigger (:756), publish the frame if there is one.
template <class Env>
void adopt(Env &env, const AttemptRec &r, StreamDesc &sd)
{
    sd.pwr.apply(r.npush, r.push_tail);
    sd.pwr.determine_snr();
    if (r.status == kAttemptFrame) env.publish(r, sd);
}

struct Cursor { int64_t pos; uint32_t cr; }; // DETECT state between attempts: position + carried d_phdr.cr

inline uint32_t recs_for(uint64_t span_items, uint32_t sps)
{ // a completed packet spans >= 13 symbols, a lost-sync attempt >= 5 (+ its DETECT steps)
    return (uint32_t)std::min<uint64_t>(span_items / (5ull * sps) + 6ull, 4096ull);
}

// The truth by construction: one job from `cur` up to `limit`, everything adopted.
template <class Env>
int run_serial(Env &env, StreamDesc &sd, Cursor &cur, int64_t limit, bool tracing)
{
    while (true) {
        std::vector<Job> jobs(1);
        Job &j = jobs[0];
        j.stream_off = sd.off; j.stream_len = sd.len; j.start = cur.pos; j.scan_limit = limit; j.stream_id = sd.id;
        j.cr_prev = cur.cr; j.max_attempts = 0; j.stop_at_header = 0;
        const uint64_t span = (uint64_t)std::max<int64_t>(limit - cur.pos, 0);
        const uint32_t rpj = recs_for(span, env.sps());
        const uint32_t trace_cap = tracing ? (uint32_t)std::min<uint64_t>(2ull * (span / env.sps()) + 64ull, 1ull << 22) : 0u;
        RunOut out;
        int s = env.run_jobs(jobs, rpj, trace_cap, out);
        if (s != 0) return s;
        const JobResult &jr = out.res[0];
        static const bool dbg_jobs = getenv("LORA_HIP_DEBUG_JOBS") != nullptr;
        if (dbg_jobs) {
            fprintf(stderr, "[serial] start %lld limit %lld cr %u | final_pos %lld cr %u n_att %u stop %u pad %u npush %u\n", (long long)j.start, (long long)j.scan_limit, j.cr_prev,
                    (long long)jr.final_pos, jr.final_cr, jr.n_attempts, jr.stop_reason, jr.pad, jr.npush);
            for (uint32_t a = 0; a < jr.n_attempts && a < out.cap; a++) {
                const AttemptRec &t = out.rec(0, a);
                fprintf(stderr, "[serial]    rec %u status %u start %lld trig %lld hdr %lld end %lld nsym %u npush %u\n", a, t.status, (long long)t.start_pos, (long long)t.trig_pos, (long long)t.hdr_pos,
                        (long long)t.end_pos, t.n_symbols, t.npush);
            }
        }
        for (uint32_t a = 0; a < out.n_done(0); a++) adopt(env, out.rec(0, a), sd);
        if (tracing) env.append_trace(out, 0, trace_cap, sd.abs_base);
        cur.cr = jr.final_cr;
        cur.pos = jr.final_pos; // start of the pending attempt when one was cut short
        if (jr.pad) { sd.incomplete = true; return 0; }
        sd.pwr.apply(jr.npush, jr.push_tail);
        if (jr.stop_reason != 2u) return 0;
    }
}

inline int cr_class(uint32_t cr) { return cr >= 3u ? 2 : (cr >= 1u ? 1 : 0); }

// Burst-aware segment plan.  `edges[i]` holds the positions where stream i goes quiet (the start of every gap between
// bursts, ascending).  Cuts are placed only at such positions and chosen greedily so that every segment carries about
// the same load (items plus a fixed acquisition cost per burst) and the whole plan is at most `slots` segments: one
// resident wave of workgroups, each with the same number of packets.  Compared with the fixed grid this removes the
// spread between jobs that happen to hold one packet more than their neighbours, the scan from a cut in mid-packet
// to the next preamble, and the failed attempt on the truncated packet.  Returns false (fixed grid) when the streams
// do not look like dense burst traffic.  Only where the cuts are changes, never what is decoded.
inline bool plan_burst_segments(const std::vector<StreamDesc> &streams, const std::vector<std::vector<int64_t>> &edges, uint32_t sps,
                                uint32_t slots, uint64_t nominal, std::vector<std::vector<int64_t>> &cuts)
{
    if (edges.size() != streams.size() || slots == 0) return false;
    size_t n_edges = 0;
    for (const auto &e : edges) n_edges += e.size();
    if (n_edges < slots) return false;                       // fewer bursts than workgroups: the fixed grid spreads the scan better
    const int64_t acq = 40ll * sps;                          // DETECT + SYNC + FIND_SFD of one packet, in items of payload work
    auto weight = [&](const std::vector<int64_t> &e, size_t k, int64_t len) { // burst k (up to its gap), k == size(): the rest
        const int64_t a = k ? e[k - 1] : 0;
        return k < e.size() ? (e[k] - a) + acq : len - a;
    };
    double total = 0;
    for (size_t i = 0; i < streams.size(); i++)
        for (size_t k = 0; k <= edges[i].size(); k++) total += (double)weight(edges[i], k, (int64_t)streams[i].len);
    double target = total / slots;
    for (int attempt = 0; attempt < 12; attempt++, target *= 1.07) {
        size_t n_segs = 0;
        cuts.assign(streams.size(), {});
        for (size_t i = 0; i < streams.size(); i++) {
            const std::vector<int64_t> &e = edges[i];
            const int64_t len = (int64_t)streams[i].len;
            double acc = 0;
            for (size_t k = 0; k < e.size(); k++) {
                acc += (double)weight(e, k, len);
                if (acc >= target - 0.5 * (double)weight(e, k + 1, len) && len - e[k] > 8ll * sps && e[k] > (cuts[i].empty() ? 0 : cuts[i].back())) {
                    cuts[i].push_back(e[k]);
                    acc = 0;
                }
            }
            // stretches without a gap (back-to-back or weak packets): fall back to the grid inside them
            std::vector<int64_t> c;
            int64_t prev = 0;
            for (size_t k = 0; k <= cuts[i].size(); k++) {
                const int64_t b = k < cuts[i].size() ? cuts[i][k] : len;
                if ((uint64_t)(b - prev) > 3u * nominal) {
                    const uint64_t parts = ((uint64_t)(b - prev) + 2u * nominal - 1u) / (2u * nominal);
                    for (uint64_t q = 1; q < parts; q++) c.push_back(prev + (int64_t)((uint64_t)(b - prev) * q / parts));
                }
                if (k < cuts[i].size()) c.push_back(b);
                prev = b;
            }
            cuts[i].swap(c);
            n_segs += cuts[i].size() + 1u;
        }
        if (n_segs <= slots) return true;
    }
    return false;
}

// Decodes a set of independent streams.
//
// Every stream is cut into fixed segments.  Round 1 runs one walker job per
// segment, each starting in DETECT at its segment boundary (a guess: the true
// decoder arrives there with some other window phase and possibly mid-packet).
// Round 2 is, for every segment, a probe that starts from the END state of the
// preceding segment's job and walks DETECT/SYNC/FIND_SFD up to the first header.
// Normally the preceding job has run it itself (Job.probe_limit: having reached its
// own limit the workgroup runs that probe as a second phase and reports it as its
// "tail"); a separate probe job is launched only where no such tail exists
// (header-less segments in between, kernels without the feature).
// The host then stitches: if the probe enters DECODE_HEADER at the same sample
// as one of the segment job's attempts, the two trajectories are identical from
// there on (the decoder state at header entry is position + d_phdr.cr), and the
// job's remaining attempts are the true ones.  Anything else is re-run serially
// from the true state.  The result is exactly the serial state machine's.
struct Seg { uint32_t stream; int64_t b0, b1; };

// What the first half of a pass (plan + main launch, decode_begin) hands to the second (decode_end: results, probes,
// stitch).  The two halves exist so that a caller can put another pass's launch between them: while the device runs
// one batch the host plans the next and stitches the previous.
struct PassCtx {
    std::vector<Seg> segs;
    std::vector<size_t> first_seg;
    std::vector<Job> jobs;
    uint32_t rpj1 = 0, rpj2 = 8, trace_cap = 0;
    bool segmenting = false, tracing = false, launched = false;
    bool decoupled = false;  // the segment jobs run header-only (LaunchCfg.skip_payload), payload_round() decodes the payloads
    std::chrono::steady_clock::time_point tp_in, tp0;
};

template <class Env>
int decode_begin(Env &env, std::vector<StreamDesc> &streams, PassCtx &ctx)
{
    const uint32_t sps = env.sps();
    const bool tracing = env.tracing();
    ctx = PassCtx{};
    ctx.tp_in = std::chrono::steady_clock::now();
    uint64_t total = 0;
    for (const StreamDesc &sd : streams) total += sd.len;
    // auto: as many segments as workgroups fit on the device at once (one wave of workgroups per launch;
    // kernel time is ceil(jobs / resident slots) x job latency), never shorter than 64 symbols
    const uint32_t slots = env.resident_slots();
    const uint64_t want_jobs = std::max<uint64_t>(slots - slots / 16u, (uint64_t)streams.size());
    uint64_t seg = env.segment_symbols() ? (uint64_t)env.segment_symbols() * sps
                                          : std::max<uint64_t>(64ull * sps, (total + want_jobs - 1) / want_jobs);
    if (seg < 16ull * sps) seg = 16ull * sps;
    const bool segmenting = !tracing && !env.implicit();
    ctx.segmenting = segmenting; ctx.tracing = tracing;

    // auto mode on a batch worth cutting up: look for the gaps between bursts and plan the cuts around them
    std::vector<std::vector<int64_t>> cuts;
    bool planned = false, balanced = false; // balanced: plan_burst_segments' plan - at least as many bursts as workgroups, every job whole packets
    static const bool no_plan = getenv("LORA_HIP_NO_BURST_PLAN") != nullptr;
    if (segmenting && env.segment_symbols() == 0 && !no_plan && total > 2ull * seg) {
        std::vector<std::vector<int64_t>> edges;
        if (env.quiet_edges(streams, edges)) {
            // a pass that will run decoupled wants one job per burst: header-only jobs cost an acquisition and a header each, whatever the packet's
            // length, so nothing is gained by balancing them - and cuts at the gaps need no explicit probes (plan_burst_segments refuses with fewer
            // bursts than workgroups, which is exactly when a pass is decoupled)
            if (env.decoupled(1)) {
                size_t nb = streams.size();
                for (const auto &e : edges) nb += e.size();
                if (edges.size() == streams.size() && env.decoupled(nb)) {
                    // ... and stretches without a gap (back-to-back packets) are cut on the grid, as plan_burst_segments does
                    cuts.assign(streams.size(), {});
                    size_t n_segs = 0;
                    for (size_t i = 0; i < streams.size(); i++) {
                        int64_t prev = 0;
                        constexpr uint64_t dsym = 24u; // (config 4 at 2 s per pass: 24 symbols 48.0 Gsamples/s, 32: 46.1, 16: 38.3 - the scan for preambles is what a header-only job spends most of its time on, and short segments spread it over the idle CUs)
                        const uint64_t dseg = std::max<uint64_t>(dsym, 8u) * sps;
                        auto grid_to = [&](int64_t b) {
                            if (2u * (uint64_t)(b - prev) > 3u * dseg) {
                                const uint64_t parts = ((uint64_t)(b - prev) + dseg - 1u) / dseg;
                                for (uint64_t q = 1; q < parts; q++) cuts[i].push_back(prev + (int64_t)((uint64_t)(b - prev) * q / parts));
                            }
                        };
                        for (int64_t c : edges[i]) {
                            if (!(c > prev && (int64_t)streams[i].len - c > 8ll * sps)) continue;
                            grid_to(c);
                            cuts[i].push_back(c);
                            prev = c;
                        }
                        grid_to((int64_t)streams[i].len);
                        n_segs += cuts[i].size() + 1u;
                    }
                    planned = env.decoupled(n_segs);
                    if (!planned) cuts.clear();
                }
            }
            if (!planned) balanced = planned = plan_burst_segments(streams, edges, sps, slots, seg, cuts);
            // fewer bursts than slots, but the kernel also exists with half as many, larger workgroups (walker3 SF9 / SF10 as one or two per CU):
            // one wave of those
            const uint32_t alt = env.resident_slots_alt();
            if (!planned && alt && alt < slots) {
                const uint64_t seg_alt = std::max<uint64_t>(64ull * sps, (total + alt - alt / 16u - 1) / (alt - alt / 16u));
                balanced = planned = plan_burst_segments(streams, edges, sps, alt, seg_alt, cuts);
            }
        }
    }

    // the fixed grid (no burst-aware plan) is sized for the LARGER workgroups where the kernel exists in two sizes: its jobs start inside packets, and
    // each pays a scan to the next preamble, an acquisition and a tail probe - twice as many jobs cost more than two workgroups per CU give back
    // (BASELINE config 4, 32 s per pass: 11.7 % of HBM peak with 244 jobs against 10.8 % with 484)
    if (!planned && segmenting && env.segment_symbols() == 0 && env.resident_slots_alt() && env.resident_slots_alt() < slots) {
        const uint32_t alt = env.resident_slots_alt();
        const uint64_t wj = std::max<uint64_t>(alt - alt / 16u, (uint64_t)streams.size());
        seg = std::max<uint64_t>(64ull * sps, (total + wj - 1) / wj);
    }
    std::vector<Seg> &segs = ctx.segs;
    std::vector<size_t> &first_seg = ctx.first_seg;
    first_seg.assign(streams.size() + 1, 0);
    for (size_t i = 0; i < streams.size(); i++) {
        first_seg[i] = segs.size();
        StreamDesc &sd = streams[i];
        sd.cr_out = sd.cr_in; sd.final_pos = 0; sd.incomplete = false;
        if (planned) { // burst-aware cuts
            int64_t prev = 0;
            for (int64_t c : cuts[i]) { segs.push_back(Seg{(uint32_t)i, prev, c}); prev = c; }
            segs.push_back(Seg{(uint32_t)i, prev, (int64_t)sd.len});
            continue;
        }
        uint64_t n = 1;
        if (segmenting && sd.len > seg + seg / 2) n = (sd.len + seg - 1) / seg;
        for (uint64_t k = 0; k < n; k++)
            segs.push_back(Seg{(uint32_t)i, (int64_t)(k * seg), (k + 1 == n) ? (int64_t)sd.len : (int64_t)((k + 1) * seg)});
    }
    first_seg[streams.size()] = segs.size();
    if (segs.empty()) return 0;
    env.note_plan(planned, segs.size());

    // ---- round 1: every segment speculatively
    std::vector<Job> &jobs = ctx.jobs;
    jobs.assign(segs.size(), Job{});
    // Decoupled pass: the jobs stop behind every header and skip the payload, whose symbols the payload pass demodulates all at once.  It pays
    // wherever the ordinary pass leaves the device short of work or makes it do work twice: few packets (a packet's symbols are a serial chain on ONE
    // CU, the others idle), or continuous traffic cut on the grid (every job first scans the rest of a packet another job decodes).  It does not pay
    // for the balanced plan - bursty traffic with at least a burst per workgroup, every job whole packets: there all CUs demodulate payload already
    // (BASELINE config 3, 256 packets: SF9 -8 %, SF12 -1 % decoupled; config 4: +100 % at 4 s per pass, +40 % at 8 s, +14 % at 32 s).
    ctx.decoupled = segmenting && !balanced && env.decoupled(jobs.size());

    uint64_t max_span = 0;
    for (size_t k = 0; k < segs.size(); k++) {
        const StreamDesc &sd = streams[segs[k].stream];
        Job &j = jobs[k];
        j.stream_off = sd.off; j.stream_len = sd.len; j.start = segs[k].b0; j.scan_limit = segs[k].b1;
        j.stream_id = sd.id; j.cr_prev = (k == first_seg[segs[k].stream]) ? sd.cr_in : env.ctor_cr();
        j.max_attempts = 0; j.stop_at_header = 0;
        // tail probe: past its own limit the job continues as the next segment's probe (same limit an explicit probe gets)
        const bool has_next = k + 1 < segs.size() && segs[k + 1].stream == segs[k].stream;
        static const bool no_tail = getenv("LORA_HIP_NO_TAIL") != nullptr; // diagnostics: separate probe jobs, as the generic kernels need
        // (a decoupled pass probes in a launch of its own: it runs beside the payload pass, on CUs that would idle, instead of lengthening every
        // header-only job by a second acquisition - config 4 at 2 s per pass 49.4 -> 54.0 Gsamples/s, at 8 s the same either way)
        j.probe_limit = (segmenting && has_next && !no_tail && !ctx.decoupled) ? std::min<int64_t>((int64_t)sd.len, segs[k + 1].b1 + 16ll * sps) : 0;
        j.tail_stop_sfd = (j.probe_limit && env.early_probe()) ? 1u : 0u;
        max_span = std::max<uint64_t>(max_span, (uint64_t)(segs[k].b1 - segs[k].b0));
    }
    ctx.rpj1 = recs_for(max_span, sps) + (segmenting ? ctx.rpj2 : 0u);
    ctx.trace_cap = tracing ? (uint32_t)std::min<uint64_t>(2ull * (max_span / sps) + 64ull, 1ull << 22) : 0u;
    env.count_jobs((uint32_t)jobs.size());
    env.set_skip_payload(ctx.decoupled);
    ctx.tp0 = std::chrono::steady_clock::now();
    const int s = env.run_jobs_begin(jobs, ctx.rpj1, ctx.trace_cap, env.run_out(0)); // (result holders are kept by the environment between calls)
    if (s != 0) return s;
    ctx.launched = true;
    return 0;
}

// The payload round of a decoupled pass.  Every kAttemptHeaderOnly record of the segment jobs goes through the payload pass (env.run_payload), which
// returns the packet's frame and by how many samples its symbols moved the symbol clock in all.
//  * Not at all: the record becomes the frame (kAttemptFrame); the job's scan behind it - started where the payload ends under that very assumption -
//    stands.
//  * By some samples: the frame stands, the scan behind it started beside the true position.  That is the situation at every segment boundary, and it
//    is handled the same way: the job is split there into two - the second half's records are a speculation like any later segment's, and a probe from
//    the TRUE end of the packet decides whether the true trajectory enters one of its headers (it does when SYNC / FIND_SFD pull both onto the next
//    preamble, as they do for a few samples' difference).
//  * The data ends inside the payload: the packet is pending, the job ends there.
//  * Unresolved: the rest of the job is run again from that packet's header by the complete kernels (Job.start_at_header, same limits, same tail
//    probe), all such jobs in one launch, and spliced in.
// What the stitch then sees are ordinary records, jobs and segments.
// (In two halves, so that the pass's explicit probes run while the device works on the payloads: payload_begin hands the requests to the payload
// pass and returns; payload_end takes the results and re-lays the pass.  It returns 1 when jobs were split, cut short or run again - the probes
// planned meanwhile are then planned again.)
struct PayloadRound {
    struct Ref { size_t k; uint32_t a; };
    std::vector<Ref> refs;
    std::vector<PayloadReq> reqs;
    bool open = false;
};

template <class Env>
int payload_begin(Env &env, const std::vector<StreamDesc> &streams, const PassCtx &ctx, const RunOut &R1, PayloadRound &pr)
{
    const uint32_t sps = env.sps();
    typedef PayloadRound::Ref Ref;
    std::vector<Ref> &refs = pr.refs;
    std::vector<PayloadReq> &reqs = pr.reqs;
    refs.clear(); reqs.clear(); pr.open = false;
    for (size_t k = 0; k < ctx.jobs.size(); k++) {
        const uint32_t nall = std::min(std::min(R1.res[k].n_attempts, R1.cap), R1.rpj);
        const StreamDesc &sd = streams[ctx.segs[k].stream];
        for (uint32_t a = 0; a < nall; a++) {
            const AttemptRec &r = R1.rec(k, a);
            if (r.status != kAttemptHeaderOnly) continue;
            PayloadReq q{};
            std::memcpy(&q.sk, r.frame, sizeof q.sk);
            q.n_walk = (uint32_t)(q.sk.payload_symbols > 0 ? q.sk.payload_symbols : 1);
            q.stream_off = sd.off; q.stream_len = sd.len;
            q.start = r.end_pos - (int64_t)q.n_walk * (int64_t)sps;
            q.hdr_pos = r.hdr_pos; q.cr_prev = r.cr_prev;
            refs.push_back(Ref{k, a});
            reqs.push_back(q);
        }
    }
    if (reqs.empty()) return 0;
    pr.open = true;
    return env.run_payload_begin(reqs);
}

template <class Env>
int payload_end(Env &env, const std::vector<StreamDesc> &streams, PassCtx &ctx, RunOut &R1, PayloadRound &pr)
{
    if (!pr.open) return 0;
    pr.open = false;
    typedef PayloadRound::Ref Ref;
    std::vector<Ref> &refs = pr.refs;
    std::vector<PayloadReq> &reqs = pr.reqs;
    int s = env.run_payload_end(reqs);
    if (s != 0) return s < 0 ? s : -1;

    // in place: frames and true end positions; per job, where it is split and where it ends early
    struct Plan { std::vector<uint32_t> split; int term = -1; uint32_t term_kind = 0; }; // split: behind these attempts; term: the job ends with this attempt (1 pending, 2 run again)
    std::vector<Plan> plan(ctx.jobs.size());
    uint32_t n_moved = 0, n_rerun = 0, n_pending = 0;
    for (size_t i = 0; i < reqs.size(); i++) {
        const size_t k = refs[i].k;
        Plan &pl = plan[k];
        if (pl.term >= 0) continue; // behind the attempt this job ends with
        AttemptRec &r = R1.recs[k * R1.rpj + refs[i].a];
        const PayloadReq &q = reqs[i];
        if (q.status == kPayloadDecoded) {
            r.status = kAttemptFrame;
            r.frame_len = q.frame_len;
            std::memcpy(r.frame, q.frame, std::min<size_t>(q.frame_len, sizeof r.frame));
            r.n_symbols += q.n_walk;
            r.end_pos += q.end_shift;
            if (q.end_shift != 0) { pl.split.push_back(refs[i].a); n_moved++; }
        } else {
            pl.term = (int)refs[i].a; pl.term_kind = q.status == kPayloadOutOfData ? 1u : 2u;
            if (pl.term_kind == 1u) n_pending++; else n_rerun++;
        }
    }
    env.count_payload((uint32_t)reqs.size(), n_moved, n_rerun);
    if (n_moved == 0 && n_rerun == 0 && n_pending == 0) return 0;

    // jobs to run again (complete kernels), one launch
    std::vector<Job> rjobs; // (in job order: the re-lay below takes their results in the same order)
    for (size_t k = 0; k < plan.size(); k++) {
        if (plan[k].term < 0 || plan[k].term_kind != 2u) continue;
        const AttemptRec &r = R1.rec(k, (uint32_t)plan[k].term);
        Job j = ctx.jobs[k];
        j.start = r.hdr_pos; j.start_at_header = 1; j.cr_prev = r.cr_prev;
        rjobs.push_back(j);
    }
    RunOut &R3 = env.run_out(1);
    R3.res.clear(); R3.recs.clear();
    if (!rjobs.empty()) {
        env.set_skip_payload(false);
        s = env.run_jobs(rjobs, ctx.rpj1, 0, R3);
        if (s != 0) return s;
    }

    // the pass re-laid: every job cut into its pieces
    struct Row { JobResult res; std::vector<AttemptRec> recs; };
    std::vector<Seg> nsegs;
    std::vector<Job> njobs;
    std::vector<Row> nrows;
    std::vector<size_t> nfirst(streams.size() + 1, 0);
    size_t rq = 0;
    for (size_t i = 0; i < streams.size(); i++) {
        nfirst[i] = nsegs.size();
        for (size_t k = ctx.first_seg[i]; k < ctx.first_seg[i + 1]; k++) {
            const Plan &pl = plan[k];
            const JobResult &jr = R1.res[k];
            const uint32_t n_rows = std::min(std::min(jr.n_attempts + (jr.tail_valid ? jr.tail_n_attempts : 0u), R1.cap), R1.rpj);
            uint32_t lo = 0;
            int64_t b0 = ctx.segs[k].b0;
            std::vector<uint32_t> cuts;
            for (uint32_t a : pl.split) if (pl.term < 0 || (int)a < pl.term) cuts.push_back(a);
            for (size_t c = 0; c <= cuts.size(); c++) {
                const bool last = c == cuts.size();
                Row row;
                Job j = ctx.jobs[k];
                Seg sg = ctx.segs[k];
                j.start = b0; sg.b0 = b0;
                if (!last) { // [lo, cuts[c]]: ends with a packet that moved the symbol clock; the scan behind it becomes the next piece
                    const uint32_t hi = cuts[c];
                    const AttemptRec &e = R1.rec(k, hi);
                    int32_t shift = 0;
                    for (size_t t = 0; t < reqs.size(); t++) if (refs[t].k == k && refs[t].a == hi) shift = reqs[t].end_shift;
                    const int64_t assumed_end = e.end_pos - shift;
                    for (uint32_t a = lo; a <= hi; a++) row.recs.push_back(R1.rec(k, a));
                    row.res = JobResult{};
                    row.res.final_pos = e.end_pos; row.res.n_attempts = hi - lo + 1u; row.res.final_cr = (uint32_t)e.frame[1] >> 5;
                    row.res.stop_reason = 0; row.res.pad = 0; row.res.npush = 0; row.res.tail_valid = 0;
                    j.scan_limit = assumed_end; j.probe_limit = 0; j.tail_stop_sfd = 0;
                    sg.b1 = assumed_end;
                    b0 = assumed_end; lo = hi + 1u;
                } else if (pl.term >= 0 && pl.term_kind == 1u) { // ends with a pending packet
                    const uint32_t hi = (uint32_t)pl.term;
                    for (uint32_t a = lo; a <= hi; a++) row.recs.push_back(R1.rec(k, a));
                    AttemptRec &e = row.recs.back();
                    e.status = kAttemptOutOfData; e.frame_len = 0;
                    row.res = JobResult{};
                    row.res.final_pos = e.start_pos; row.res.n_attempts = hi - lo + 1u; row.res.final_cr = (uint32_t)e.frame[1] >> 5;
                    row.res.stop_reason = 1; row.res.pad = 1; row.res.npush = 0; row.res.tail_valid = 0;
                } else if (pl.term >= 0) { // ends with the re-run: the packet (its scan and acquisition are the original attempt's) and whatever the complete kernels found behind it
                    const uint32_t hi = (uint32_t)pl.term;
                    for (uint32_t a = lo; a < hi; a++) row.recs.push_back(R1.rec(k, a));
                    JobResult nr = R3.res[rq];
                    const uint32_t n_new = std::min(nr.n_attempts + (nr.tail_valid ? nr.tail_n_attempts : 0u), R3.rpj);
                    const AttemptRec &old = R1.rec(k, hi);
                    for (uint32_t t = 0; t < n_new; t++) row.recs.push_back(R3.rec(rq, t));
                    if (nr.n_attempts >= 1u && n_new >= 1u) {
                        AttemptRec &m = row.recs[hi - lo];
                        m.start_pos = old.start_pos; m.trig_pos = old.trig_pos; m.npush = old.npush;
                        for (int t = 0; t < 4; t++) m.push_tail[t] = old.push_tail[t];
                        m.n_sfd = old.n_sfd;
                        for (int t = 0; t < kMaxSfdRec; t++) { m.sfd_pos[t] = old.sfd_pos[t]; m.sfd_fails[t] = old.sfd_fails[t]; }
                        if (nr.pad && nr.n_attempts == 1u) nr.final_pos = old.start_pos; // pending: the stream resumes at the start of this attempt's scan
                    }
                    nr.n_attempts += hi - lo;
                    nr.tail_first_rec += hi - lo;
                    row.res = nr;
                    rq++;
                } else { // the rest of the job as it ran
                    for (uint32_t a = lo; a < n_rows; a++) row.recs.push_back(R1.rec(k, a));
                    row.res = jr;
                    row.res.n_attempts = jr.n_attempts >= lo ? jr.n_attempts - lo : 0u;
                    row.res.tail_first_rec = jr.tail_first_rec >= lo ? jr.tail_first_rec - lo : 0u;
                }
                nsegs.push_back(sg);
                njobs.push_back(j);
                nrows.push_back(std::move(row));
            }
        }
    }
    nfirst[streams.size()] = nsegs.size();
    uint32_t stride = 1;
    for (const Row &r : nrows) stride = std::max<uint32_t>(stride, (uint32_t)r.recs.size());
    R1.res.resize(nrows.size());
    R1.recs.resize_uninit(nrows.size() * (size_t)stride);
    R1.rpj = stride; R1.cap = std::max(R1.cap, stride);
    for (size_t k = 0; k < nrows.size(); k++) {
        R1.res[k] = nrows[k].res;
        for (size_t a = 0; a < nrows[k].recs.size(); a++) R1.recs[k * stride + a] = nrows[k].recs[a];
    }
    ctx.segs.swap(nsegs); ctx.jobs.swap(njobs); ctx.first_seg.swap(nfirst);
    return 1;
}

template <class Env>
int decode_end(Env &env, std::vector<StreamDesc> &streams, PassCtx &ctx)
{
    if (!ctx.launched) return 0; // nothing to decode
    ctx.launched = false;
    const uint32_t sps = env.sps();
    const bool tracing = ctx.tracing, segmenting = ctx.segmenting;
    const std::vector<Seg> &segs = ctx.segs;
    const std::vector<size_t> &first_seg = ctx.first_seg;
    const std::vector<Job> &jobs = ctx.jobs;
    const uint32_t rpj1 = ctx.rpj1, rpj2 = ctx.rpj2, trace_cap = ctx.trace_cap;
    const auto tp_in = ctx.tp_in, tp0 = ctx.tp0;
    (void)segmenting;
    RunOut &R1 = env.run_out(0);
    static const bool dbg_t = getenv("LORA_HIP_DEBUG") != nullptr;
    int s = env.run_jobs_end(R1);
    if (s != 0) return s;
    const auto tp1 = std::chrono::steady_clock::now();
    static const bool dbg_jobs = getenv("LORA_HIP_DEBUG_JOBS") != nullptr;
    if (dbg_jobs) { // diagnostics: every segment job's result, comparable between the device and the CPU simulation
        for (size_t k = 0; k < jobs.size(); k++) {
            const JobResult &r = R1.res[k];
            fprintf(stderr, "[job] %zu start %lld limit %lld probe_limit %lld | final_pos %lld cr %u n_att %u stop %u pad %u npush %u | tail %u: first %u n_att %u final_pos %lld cr %u stop %u pad %u npush %u\n",
                    k, (long long)jobs[k].start, (long long)jobs[k].scan_limit, (long long)jobs[k].probe_limit, (long long)r.final_pos, r.final_cr, r.n_attempts, r.stop_reason, r.pad,
                    r.npush, r.tail_valid, r.tail_first_rec, r.tail_n_attempts, (long long)r.tail_final_pos, r.tail_final_cr, r.tail_stop_reason, r.tail_pad, r.tail_npush);
            const uint32_t n = r.n_attempts + (r.tail_valid ? r.tail_n_attempts : 0u);
            for (uint32_t a = 0; a < n && a < R1.cap; a++) {
                const AttemptRec &t = R1.rec(k, a);
                fprintf(stderr, "[job]    rec %u status %u start %lld trig %lld hdr %lld end %lld nsym %u npush %u cr_prev %u ambig %u len %u\n", a, t.status, (long long)t.start_pos,
                        (long long)t.trig_pos, (long long)t.hdr_pos, (long long)t.end_pos, t.n_symbols, t.npush, t.cr_prev, t.hdr_ambig, t.frame_len);
            }
        }
    }
    // ---- round 2: probes along the speculative chain.  Only segments whose own job
    // reached a header get a probe; it starts from the end state of the previous
    // header-bearing job and scans through any header-less segments in between.
    auto has_header = [&](size_t k) {
        const uint32_t nall = std::min(R1.res[k].n_attempts, R1.cap);
        for (uint32_t a = 0; a < nall; a++) {
            const AttemptRec &r = R1.rec(k, a);
            if (r.hdr_pos >= 0 && (r.status == kAttemptFrame || r.status == kAttemptOutOfData || r.status == kAttemptHeaderOnly)) return true;
        }
        return false;
    };
    auto job_ok = [&](size_t k) { return !R1.res[k].pad && R1.res[k].stop_reason != 2u; };
    // ---- round 1b: segment jobs that guessed the wrong header FEC branch.  A later segment's job runs with the constructor's d_phdr.cr; the true
    // one is its predecessor's last header's (:655).  Where the two are of different Hamming classes AND the two decodes of the job's first header
    // disagree (bit errors: hdr_ambig), its frames are not the true decoder's - at CR 4/5 / 4/6 under noise that is most packets - and used to cost
    // one serial launch per segment (32 ms for 16 of SF12's 256 packets).  The predecessor's tail probe has already reported the true value
    // (cr_prev of its pending record): all such jobs are run again, together, with it, and take their originals' place before anything is
    // stitched.  (The choice is checked like any other speculation: the stitch compares branch classes again.)
    if (segmenting) {
        std::vector<Job> rjobs;
        std::vector<size_t> rk;
        for (size_t i = 0; i < streams.size(); i++) {
            for (size_t k = first_seg[i] + 1; k < first_seg[i + 1]; k++) {
                const JobResult &pj = R1.res[k - 1];
                if (!pj.tail_valid || !pj.tail_pad || pj.tail_n_attempts == 0u) continue;
                const uint32_t li = std::min(pj.tail_first_rec, R1.cap) + pj.tail_n_attempts - 1u;
                if (li >= R1.cap) continue;
                const AttemptRec &L = R1.rec(k - 1, li);
                if (L.status != kAttemptAtHeader && L.status != kAttemptAtSfd) continue;
                const bool at_sfd = L.status == kAttemptAtSfd;
                if (at_sfd && L.n_sfd == 0u) continue;
                bool wrong = false, right = false;
                const uint32_t nall = std::min(R1.res[k].n_attempts, R1.cap);
                for (uint32_t a = 0; a < nall; a++) {
                    const AttemptRec &r = R1.rec(k, a);
                    if (r.hdr_pos < 0 || (r.status != kAttemptFrame && r.status != kAttemptOutOfData && r.status != kAttemptHeaderOnly)) continue;
                    bool same = !at_sfd && r.hdr_pos == L.hdr_pos;
                    if (at_sfd)
                        for (uint32_t z = 0; z < r.n_sfd && z < (uint32_t)kMaxSfdRec; z++)
                            same = same || (r.sfd_pos[z] == L.sfd_pos[L.n_sfd - 1u] && r.sfd_fails[z] == L.sfd_fails[L.n_sfd - 1u]);
                    if (!same) continue;
                    const int cj = cr_class(r.cr_prev), cl = cr_class(L.cr_prev);
                    if (cj != cl && (r.hdr_ambig || cj == 0 || cl == 0)) wrong = true; else right = true;
                    break;
                }
                if (!wrong || right) continue;
                Job j = jobs[k];
                j.cr_prev = L.cr_prev;
                rjobs.push_back(j);
                rk.push_back(k);
            }
        }
        if (!rjobs.empty()) {
            RunOut &R3 = env.run_out(1);
            R3.res.clear(); R3.recs.clear();
            env.count_slow_path();
            env.set_skip_payload(ctx.decoupled);
            s = env.run_jobs(rjobs, rpj1, 0, R3);
            if (s != 0) return s;
            for (size_t q = 0; q < rk.size(); q++) {
                const JobResult &nr = R3.res[q];
                const uint32_t n = nr.n_attempts + (nr.tail_valid ? nr.tail_n_attempts : 0u);
                if (n > R1.rpj || n > R3.cap) continue; // (does not fit the original's row: the original stands, the stitch falls back)
                R1.res[rk[q]] = nr;
                for (uint32_t a = 0; a < n; a++) R1.recs[rk[q] * R1.rpj + a] = R3.rec(q, a);
            }
            if (dbg_t) fprintf(stderr, "[lora_hip] %zu segment job(s) run again with the header FEC branch their predecessor's tail probe reported\n", rjobs.size());
        }
    }
    // the payloads of the header-only jobs: the payload pass starts here and is collected behind the launch of the explicit probes below - which are
    // planned on the records as they stand (a header-only record says where its packet ends if nothing moves the symbol clock) and planned again in the
    // rare pass whose payloads change that (payload_end: jobs split, cut short or run again)
    PayloadRound pround;
    // whatever goes wrong between payload_begin and payload_end (a failed probe launch, a failed round of the payload pass itself), the pass's kernels may
    // still be reading and writing the handle's page-locked staging buffers on their own stream: every early return drains it first
    struct PayloadGuard { Env &env; bool armed; ~PayloadGuard() { if (armed) env.abort_payload(); } } pguard{env, false};
    if (ctx.decoupled) {
        pguard.armed = true;
        s = payload_begin(env, streams, ctx, R1, pround);
        if (s != 0) return s;
        pguard.armed = pround.open; // (a pass without a header-only record launched nothing: no stream to drain, and the staging vectors keep their capacity)
    }
    env.set_skip_payload(false);
    struct Probe { uint32_t stream; size_t target; Cursor start; int job; int tail_of; }; // job: index into pjobs, or -1 with tail_of = the job whose tail it is
    std::vector<Probe> probes;
    std::vector<size_t> first_probe(streams.size() + 1, 0);
    std::vector<Job> pjobs;
    std::vector<int> repair_of; // per probe: index into pjobs of the job that runs the rest of its target segment again from the true header, or -1
    static const bool no_repair = getenv("LORA_HIP_NO_REPAIR") != nullptr; // (A/B: every mismatching cut takes the serial path, as until round 6)
    RunOut &R2 = env.run_out(1);
    for (int planning = 0; planning < 2; planning++) {
    probes.clear(); pjobs.clear();
    first_probe.assign(streams.size() + 1, 0);
    for (size_t i = 0; i < streams.size(); i++) {
        first_probe[i] = probes.size();
        const size_t f = first_seg[i], e = first_seg[i + 1];
        if (e - f < 2) continue;
        const StreamDesc &sd = streams[i];
        bool chain = job_ok(f), pending = false;
        Cursor cur{R1.res[f].final_pos, R1.res[f].final_cr};
        size_t cur_job = f; // the job whose end state `cur` is
        auto add_probe = [&](size_t target, int64_t limit) {
            if (R1.res[cur_job].tail_valid && jobs[cur_job].probe_limit == limit) { // that job has already run this probe
                probes.push_back(Probe{(uint32_t)i, target, cur, -1, (int)cur_job});
                return;
            }
            Job j{};
            j.stream_off = sd.off; j.stream_len = sd.len; j.start = cur.pos; j.scan_limit = limit;
            j.stream_id = sd.id; j.cr_prev = cur.cr; j.max_attempts = 0; j.stop_at_header = 1;
            probes.push_back(Probe{(uint32_t)i, target, cur, (int)pjobs.size(), -1});
            pjobs.push_back(j);
        };
        for (size_t k = f + 1; k < e && chain; k++) {
            if (cur.pos >= segs[k].b1) continue; // a packet ran across this whole segment
            if (!has_header(k)) { pending = true; continue; }
            add_probe(k, std::min<int64_t>((int64_t)sd.len, segs[k].b1 + 16ll * sps));
            chain = job_ok(k);
            cur = Cursor{R1.res[k].final_pos, R1.res[k].final_cr};
            cur_job = k;
            pending = false;
        }
        if (chain && pending) add_probe(e - 1, (int64_t)sd.len); // header-less tail still has to be walked
    }
    first_probe[streams.size()] = probes.size();
    // Tail probes that stopped behind their first FIND_SFD step (Job.tail_stop_sfd) in a state NO header-bearing attempt of their successors
    // passed through - the successor triggered two or more chirps later, as happens when a cut falls inside a packet train: they are run to the
    // header after all, as explicit probe jobs in the one launch below (a mismatch must cost a probe, not a serial walk of the segment).
    bool through = false; // some explicit probe of this launch walks its whole target segment (it needs a segment job's record capacity)
    for (size_t q = 0; q < probes.size(); q++) {
        if (probes[q].job >= 0) continue;
        const size_t tj = (size_t)probes[q].tail_of;
        const JobResult &jr = R1.res[tj];
        if (!jr.tail_pad || jr.tail_n_attempts == 0u) continue;
        const uint32_t li = std::min(jr.tail_first_rec, R1.cap) + jr.tail_n_attempts - 1u;
        if (li >= R1.cap) continue;
        const AttemptRec &L = R1.rec(tj, li);
        if (L.status != kAttemptAtSfd || L.n_sfd == 0u) continue;
        const int64_t ppos = L.sfd_pos[L.n_sfd - 1u];
        const uint32_t pfails = L.sfd_fails[L.n_sfd - 1u];
        bool shared = false;
        for (size_t k = first_seg[probes[q].stream] + 1; k <= probes[q].target && !shared; k++) {
            const uint32_t nall = std::min(R1.res[k].n_attempts, R1.cap);
            for (uint32_t a = 0; a < nall && !shared; a++) {
                const AttemptRec &r = R1.rec(k, a);
                if (r.hdr_pos < 0 || (r.status != kAttemptFrame && r.status != kAttemptOutOfData && r.status != kAttemptHeaderOnly)) continue;
                for (uint32_t z = 0; z < r.n_sfd && z < (uint32_t)kMaxSfdRec; z++) shared = shared || (r.sfd_pos[z] == ppos && r.sfd_fails[z] == pfails);
            }
        }
        if (shared) continue;
        const StreamDesc &sd = streams[probes[q].stream];
        Job j{};
        j.stream_off = sd.off; j.stream_len = sd.len; j.start = probes[q].start.pos; j.scan_limit = jobs[tj].probe_limit;
        j.stream_id = sd.id; j.cr_prev = probes[q].start.cr; j.max_attempts = 0; j.stop_at_header = 1;
        if (!tracing && !no_repair) {
            // (round 6) ... and past it: the job walks the whole target segment from the true state - the true trajectory itself, adopted by the stitch as a probe that
            // "ended without a header" at its limit; with noise over the stream such cuts are every second one, and a probe that stops at a header one sample beside the
            // speculative job's would send the segment down the serial path
            j.scan_limit = jobs[probes[q].target].scan_limit; j.stop_at_header = 0;
            through = true;
        }
        probes[q].job = (int)pjobs.size(); probes[q].tail_of = -1;
        pjobs.push_back(j);
    }
    // Tail probes that reached a header NO successor job entered at the same sample.  With any noise over the stream detect_upchirp's tie between adjacent
    // shifts is decided by the noise, differently for two DETECT alignments: the speculative job then sits ONE sample beside the true trajectory, at about
    // every second cut.  Walking such a segment serially, one workgroup at a time, is what a pass would then spend its time on (round 6: 263 -> 5 Gsamples/s
    // at a noise floor 60 dB down); instead the REST of the target segment's job is run again from the true header - Job.start_at_header, the true d_phdr.cr, the
    // job's own limits and tail probe - all such cuts in the one launch below.  The stitch adopts a repair's records as the true trajectory (which they are); where
    // its end state is the speculative job's - fine_sync pulls the two together within a packet - the chain goes on, else the next cut's check falls back as before.
    repair_of.assign(probes.size(), -1);
    uint32_t rpj_launch = through ? std::max(rpj2, rpj1) : rpj2;
    if (!tracing && !no_repair) {
        for (size_t q = 0; q < probes.size(); q++) {
            if (probes[q].job >= 0) continue; // (an explicit probe's result is not known yet: it keeps the serial path)
            const size_t tj = (size_t)probes[q].tail_of;
            const JobResult &jr = R1.res[tj];
            if (!jr.tail_pad || jr.tail_n_attempts == 0u) continue;
            const uint32_t li = std::min(jr.tail_first_rec, R1.cap) + jr.tail_n_attempts - 1u;
            if (li >= R1.cap) continue;
            const AttemptRec &L = R1.rec(tj, li);
            if (L.status != kAttemptAtHeader || L.hdr_pos < 0) continue;
            bool entered = false;
            for (size_t k = first_seg[probes[q].stream] + 1; k <= probes[q].target && !entered; k++) {
                const uint32_t nall = std::min(R1.res[k].n_attempts, R1.cap);
                for (uint32_t a = 0; a < nall && !entered; a++) {
                    const AttemptRec &r = R1.rec(k, a);
                    entered = r.hdr_pos == L.hdr_pos && (r.status == kAttemptFrame || r.status == kAttemptOutOfData || r.status == kAttemptHeaderOnly); // (header-only: a decoupled pass ahead of its payload_end)
                }
            }
            if (entered) continue; // (the stitch's own match - with its FEC-branch condition - decides)
            const StreamDesc &sd = streams[probes[q].stream];
            const Job &tjob = jobs[probes[q].target];
            Job j{};
            j.stream_off = sd.off; j.stream_len = sd.len; j.start = L.hdr_pos; j.start_at_header = 1; j.cr_prev = L.cr_prev;
            j.scan_limit = tjob.scan_limit; j.probe_limit = tjob.probe_limit; j.tail_stop_sfd = 0;
            j.stream_id = sd.id; j.max_attempts = 0; j.stop_at_header = 0;
            repair_of[q] = (int)pjobs.size();
            pjobs.push_back(j);
            rpj_launch = std::max(rpj_launch, rpj1);
        }
    }
    R2.res.clear(); R2.recs.clear();
    if (!pjobs.empty()) {
        env.count_probes((uint32_t)pjobs.size());
        s = env.run_jobs(pjobs, rpj_launch, 0, R2);
        if (s != 0) return s;
    }
    if (!pround.open) break;
    s = payload_end(env, streams, ctx, R1, pround); // from here on the records are ordinary ones
    if (s < 0) return s;
    pguard.armed = false;
    if (s == 0) break; // (1: the pass was re-laid - its probes are planned again)
    }
    // ... and the same repair for EXPLICIT probes (a decoupled pass has no tail probes; the generic kernels none either): that a probe job reached a header no
    // segment job entered is known only now, so these repairs take a launch of their own - one for all of them
    RunOut R4;
    std::vector<int> repair4_of(probes.size(), -1);
    if (!tracing && !no_repair) {
        std::vector<Job> rjobs4;
        for (size_t q = 0; q < probes.size(); q++) {
            if (probes[q].job < 0) continue;
            const size_t pj = (size_t)probes[q].job;
            if (!pjobs[pj].stop_at_header) continue; // (a probe that walks its whole target segment: nothing to match)
            const JobResult &pr = R2.res[pj];
            if (!pr.pad || pr.n_attempts == 0u || pr.n_attempts > R2.cap) continue;
            const AttemptRec &L = R2.rec(pj, pr.n_attempts - 1u);
            if (L.status != kAttemptAtHeader || L.hdr_pos < 0) continue;
            bool entered = false;
            for (size_t k = first_seg[probes[q].stream] + 1; k <= probes[q].target && !entered; k++) {
                const uint32_t nall = std::min(R1.res[k].n_attempts, R1.cap);
                for (uint32_t a = 0; a < nall && !entered; a++) {
                    const AttemptRec &r = R1.rec(k, a);
                    entered = r.hdr_pos == L.hdr_pos && (r.status == kAttemptFrame || r.status == kAttemptOutOfData || r.status == kAttemptHeaderOnly);
                }
            }
            if (entered) continue;
            const StreamDesc &sd = streams[probes[q].stream];
            const Job &tjob = jobs[probes[q].target];
            Job j{};
            j.stream_off = sd.off; j.stream_len = sd.len; j.start = L.hdr_pos; j.start_at_header = 1; j.cr_prev = L.cr_prev;
            j.scan_limit = tjob.scan_limit; j.probe_limit = tjob.probe_limit; j.tail_stop_sfd = 0;
            j.stream_id = sd.id; j.max_attempts = 0; j.stop_at_header = 0;
            repair4_of[q] = (int)rjobs4.size();
            rjobs4.push_back(j);
        }
        if (!rjobs4.empty()) {
            env.count_probes((uint32_t)rjobs4.size());
            s = env.run_jobs(rjobs4, rpj1, 0, R4);
            if (s != 0) return s;
        }
    }
    if (dbg_jobs) {
        for (size_t k = 0; k < pjobs.size(); k++) {
            const JobResult &r = R2.res[k];
            fprintf(stderr, "[probe] %zu start %lld limit %lld cr %u | final_pos %lld cr %u n_att %u stop %u pad %u npush %u\n", k, (long long)pjobs[k].start, (long long)pjobs[k].scan_limit,
                    pjobs[k].cr_prev, (long long)r.final_pos, r.final_cr, r.n_attempts, r.stop_reason, r.pad, r.npush);
            for (uint32_t a = 0; a < r.n_attempts && a < R2.cap; a++) {
                const AttemptRec &t = R2.rec(k, a);
                fprintf(stderr, "[probe]    rec %u status %u start %lld trig %lld hdr %lld end %lld nsym %u npush %u cr_prev %u ambig %u\n", a, t.status, (long long)t.start_pos, (long long)t.trig_pos,
                        (long long)t.hdr_pos, (long long)t.end_pos, t.n_symbols, t.npush, t.cr_prev, t.hdr_ambig);
            }
        }
    }
    // one view per probe, whether it ran as its own job or as the tail of the preceding segment's job
    struct ProbeView { JobResult res; const AttemptRec *recs; uint32_t cap; int64_t limit; };
    std::vector<ProbeView> pv(probes.size());
    for (size_t q = 0; q < probes.size(); q++) {
        ProbeView &v = pv[q];
        if (probes[q].job >= 0) {
            const size_t pj = (size_t)probes[q].job;
            v.res = R2.res[pj]; v.recs = R2.recs.data() + pj * R2.rpj; v.cap = R2.cap; v.limit = pjobs[pj].scan_limit;
        } else {
            const size_t tj = (size_t)probes[q].tail_of;
            const JobResult &jr = R1.res[tj];
            v.res = JobResult{};
            v.res.final_pos = jr.tail_final_pos; v.res.n_attempts = jr.tail_n_attempts; v.res.final_cr = jr.tail_final_cr;
            v.res.npush = jr.tail_npush;
            for (int t = 0; t < 4; t++) v.res.push_tail[t] = jr.tail_push_tail[t];
            v.res.stop_reason = jr.tail_stop_reason; v.res.pad = jr.tail_pad;
            const uint32_t first = std::min(jr.tail_first_rec, R1.cap);
            v.recs = R1.recs.data() + tj * R1.rpj + first; v.cap = R1.cap - first; v.limit = jobs[tj].probe_limit;
        }
    }
    auto pv_done = [&](const ProbeView &v) { // attempts that ran to completion (cf. RunOut::n_done)
        const uint32_t n = v.res.pad ? v.res.n_attempts - 1u : v.res.n_attempts;
        return n < v.cap ? n : v.cap;
    };

    const auto tp2 = std::chrono::steady_clock::now();
    // ---- stitch, stream by stream, in stream order
    for (size_t i = 0; i < streams.size(); i++) {
        StreamDesc &sd = streams[i];
        const size_t f = first_seg[i];
        // the first segment starts from the true state: adopt it wholesale
        for (uint32_t a = 0; a < R1.n_done(f); a++) adopt(env, R1.rec(f, a), sd);
        if (tracing) env.append_trace(R1, (uint32_t)f, trace_cap, sd.abs_base);
        Cursor cur{R1.res[f].final_pos, R1.res[f].final_cr};
        int64_t covered = segs[f].b1; // the true trajectory is known up to here
        if (R1.res[f].pad) sd.incomplete = true;
        else sd.pwr.apply(R1.res[f].npush, R1.res[f].push_tail);
        static const bool dbg = getenv("LORA_HIP_DEBUG") != nullptr;
        auto serial_to = [&](int64_t limit, const char *why) -> int {
            if (dbg) fprintf(stderr, "[lora_hip] serial fallback stream %u pos %lld -> %lld: %s\n", sd.id, (long long)cur.pos, (long long)limit, why);
            env.count_slow_path();
            int r = run_serial(env, sd, cur, limit, false);
            covered = std::max(covered, limit);
            return r;
        };
        if (!sd.incomplete && R1.res[f].stop_reason == 2u) {
            s = serial_to(segs[f].b1, "first segment out of records");
            if (s != 0) return s;
        }
        for (size_t q = first_probe[i]; q < first_probe[i + 1] && !sd.incomplete; q++) {
            const Probe &pb = probes[q];
            const int64_t b1 = segs[pb.target].b1;
            if (cur.pos >= b1) continue;
            if (pb.start.pos != cur.pos || pb.start.cr != cur.cr) { // speculation chain broken: redo serially
                s = serial_to(b1, "probe start differs from the true state");
                if (s != 0) return s;
                continue;
            }
            const ProbeView &view = pv[q];
            const JobResult &pr = view.res;
            // lost-sync attempts the true trajectory went through before the header
            for (uint32_t a = 0; a < pv_done(view); a++) adopt(env, view.recs[a], sd);
            if (!pr.pad) { // no header before the probe's limit
                cur = Cursor{pr.final_pos, pr.final_cr};
                sd.pwr.apply(pr.npush, pr.push_tail);
                covered = std::max(covered, std::min<int64_t>(view.limit, b1));
                if (pr.stop_reason == 2u || (cur.pos < b1 && cur.pos + 2 * (int64_t)sps <= (int64_t)sd.len)) {
                    s = serial_to(b1, "probe ended without a header");
                    if (s != 0) return s;
                }
                continue;
            }
            if (pr.n_attempts > view.cap) { // the pending attempt's record did not fit
                s = serial_to(b1, "probe out of records");
                if (s != 0) return s;
                continue;
            }
            const AttemptRec &L = view.recs[pr.n_attempts - 1u];
            if (L.status != kAttemptAtHeader && L.status != kAttemptAtSfd) { // ran out of data before reaching a header
                cur.pos = L.start_pos;
                sd.incomplete = true;
                break;
            }
            // which segment job entered a header at the same sample?  (A probe that stopped behind its first FIND_SFD step, Job.tail_stop_sfd:
            // which job's header-bearing attempt STARTED a FIND_SFD step in the state the probe stopped in - position and d_corr_fails, all
            // that decoder_impl.cc:785-818 read: from that step on the two trajectories are one, header entry included.)
            const bool at_sfd = L.status == kAttemptAtSfd;
            const int64_t probe_sfd_pos = (at_sfd && L.n_sfd) ? L.sfd_pos[L.n_sfd - 1u] : -1;
            const uint32_t probe_sfd_fails = (at_sfd && L.n_sfd) ? L.sfd_fails[L.n_sfd - 1u] : 0u;
            int match = -1;
            size_t mk = 0;
            for (size_t k = f + 1; k <= pb.target && match < 0; k++) {
                if (segs[k].b1 <= cur.pos) continue;
                const uint32_t nall = std::min(R1.res[k].n_attempts, R1.cap);
                for (uint32_t a = 0; a < nall; a++) {
                    const AttemptRec &r = R1.rec(k, a);
                    if (r.hdr_pos < 0 || (r.status != kAttemptFrame && r.status != kAttemptOutOfData)) continue;
                    if (!at_sfd) { if (r.hdr_pos != L.hdr_pos) continue; }
                    else {
                        bool shared = false;
                        for (uint32_t q = 0; q < r.n_sfd && q < (uint32_t)kMaxSfdRec; q++) shared = shared || (r.sfd_pos[q] == probe_sfd_pos && r.sfd_fails[q] == probe_sfd_fails);
                        if (!shared) continue;
                    }
                    { // the header FEC branch follows the carried-in d_phdr.cr (:655): the job's speculative decode only stands
                      // if its branch is the true one, or the two Hamming branches agree on this header - and class 0
                      // (no switch case upstream: the header reads as zeros) never agrees with either
                        const int cj = cr_class(r.cr_prev), cl = cr_class(L.cr_prev);
                        if (cj != cl && (r.hdr_ambig || cj == 0 || cl == 0)) continue;
                    }
                    match = (int)a; mk = k;
                    break;
                }
            }
            const bool rep2 = q < repair_of.size() && repair_of[q] >= 0, rep4 = repair4_of[q] >= 0;
            if (match < 0 && (rep2 || rep4) && !at_sfd) { // the rest of the target segment, run again from the true header
                const RunOut &RR = rep2 ? R2 : R4;
                const size_t rj = (size_t)(rep2 ? repair_of[q] : repair4_of[q]);
                const JobResult &rr = RR.res[rj];
                const uint32_t nd = RR.n_done(rj);
                if (nd >= 1u && RR.rec(rj, 0).hdr_pos == L.hdr_pos && rr.stop_reason != 2u) {
                    sd.pwr.apply(L.npush, L.push_tail); // the true DETECT scan is the probe's, everything from the header on the repair's
                    sd.pwr.determine_snr();
                    if (RR.rec(rj, 0).status == kAttemptFrame) env.publish(RR.rec(rj, 0), sd);
                    for (uint32_t a = 1; a < nd; a++) adopt(env, RR.rec(rj, a), sd);
                    cur = Cursor{rr.final_pos, rr.final_cr};
                    if (rr.pad) { sd.incomplete = true; break; }
                    sd.pwr.apply(rr.npush, rr.push_tail);
                    covered = std::max(covered, segs[pb.target].b1);
                    env.count_repair();
                    continue;
                }
            }
            if (match < 0) {
                if (dbg) {
                    fprintf(stderr, "[lora_hip] probe for segment %zu [%lld, %lld): start %lld cr %u -> trig %lld hdr %lld (stopped %s, FIND_SFD state %lld / %u)\n", pb.target, (long long)segs[pb.target].b0, (long long)segs[pb.target].b1,
                            (long long)pb.start.pos, pb.start.cr, (long long)L.trig_pos, (long long)L.hdr_pos, at_sfd ? "behind its first FIND_SFD step" : "at the header", (long long)probe_sfd_pos, probe_sfd_fails);
                    for (size_t k = pb.target ? pb.target - 1 : 0; k <= pb.target; k++)
                        for (uint32_t a = 0; a < std::min(R1.res[k].n_attempts, R1.cap); a++) {
                            const AttemptRec &r = R1.rec(k, a);
                            fprintf(stderr, "[lora_hip]    job %zu [%lld, %lld) rec %u status %u start %lld trig %lld hdr %lld end %lld\n", k, (long long)segs[k].b0, (long long)segs[k].b1, a, r.status,
                                    (long long)r.start_pos, (long long)r.trig_pos, (long long)r.hdr_pos, (long long)r.end_pos);
                        }
                }
                cur = Cursor{L.start_pos, L.cr_prev};
                s = serial_to(b1, at_sfd ? "no segment job passed through the probe's FIND_SFD state" : "no segment job entered the same header");
                if (s != 0) return s;
                continue;
            }
            const AttemptRec &m = R1.rec(mk, (uint32_t)match);
            if (m.status == kAttemptOutOfData) {
                cur = Cursor{L.start_pos, L.cr_prev};
                sd.incomplete = true;
                break;
            }
            // merged: the true DETECT scan is the probe's, everything after the header is the job's
            sd.pwr.apply(L.npush, L.push_tail);
            sd.pwr.determine_snr();
            env.publish(m, sd);
            for (uint32_t a = (uint32_t)match + 1u; a < R1.n_done(mk); a++) adopt(env, R1.rec(mk, a), sd);
            const JobResult &jr = R1.res[mk];
            cur = Cursor{jr.final_pos, jr.final_cr};
            if (jr.pad) { sd.incomplete = true; break; }
            sd.pwr.apply(jr.npush, jr.push_tail);
            covered = std::max(covered, segs[mk].b1);
            if (jr.stop_reason == 2u) {
                s = serial_to(segs[mk].b1, "segment job out of records");
                if (s != 0) return s;
            }
        }
        // whatever the probes did not cover is walked serially (exactness before speed)
        if (!sd.incomplete && cur.pos < (int64_t)sd.len && covered < (int64_t)sd.len &&
            cur.pos + 2 * (int64_t)sps <= (int64_t)sd.len) {
            s = serial_to((int64_t)sd.len, "uncovered tail");
            if (s != 0) return s;
        }
        sd.final_pos = cur.pos;
        sd.cr_out = cur.cr;
    }
    if (dbg_t) {
        const auto tp3 = std::chrono::steady_clock::now();
        auto ms = [](auto a, auto b) { return std::chrono::duration<double, std::milli>(b - a).count(); };
        fprintf(stderr, "[lora_hip] plan %.3f ms, round1 %.3f ms (%zu jobs, rpj %u), round2 %.3f ms (%zu probe jobs, %zu tail probes), stitch %.3f ms, walker %.3f ms\n",
                ms(tp_in, tp0), ms(tp0, tp1), jobs.size(), rpj1, ms(tp1, tp2), pjobs.size(), probes.size() - pjobs.size(), ms(tp2, tp3), env.walker_ms());
    }
    return 0;
}

// One pass, start to finish (the synchronous entry points and the CPU simulation).
template <class Env>
int decode_streams(Env &env, std::vector<StreamDesc> &streams)
{
    PassCtx ctx;
    const int s = decode_begin(env, streams, ctx);
    if (s != 0) return s;
    return decode_end(env, streams, ctx);
}


} // namespace lora_hip
