// stitch_sim.cpp -- TEST-ONLY environment for the product's speculation scheduler (gr_lora_amd/csrc/lora_stitch.hpp):
// jobs are executed by the CPU oracle's state machine instead of the walker kernels, so that segmenting,
// probing and stitching are exercised by the CPU test-suite.  Built by tests/test_stitch_sim.py with g++.
#include <cstring>
#include <vector>

#include "../../gr_lora_amd/csrc/lora_stitch.hpp"
#include "../../oracle/lora_oracle.h"

using namespace lora_hip;

namespace {
struct SimFrame { std::vector<uint8_t> blob; int64_t hdr_pos; };

struct SimEnv {
    lora_oracle_t *o;
    const float *iq;
    size_t n_items;
    uint32_t sps_, ctor_cr_, seg_symbols, slots;
    std::vector<SimFrame> frames;
    uint32_t n_jobs = 0, n_probes = 0, n_slow = 0, n_tails = 0, n_early = 0;
    RunOut outs[2];
    RunOut &run_out(int which) { return outs[which & 1]; }
    bool burst_plan = false;  // offer the scheduler the gaps between bursts (the device's envelope pre-pass)
    uint32_t planned = 0;
    bool tail_probes = true; // emulate Job.probe_limit (walker2); false: the generic kernels' behaviour (explicit probe jobs only)
    bool early = false;      // emulate walker3: FIND_SFD entry states in the attempt records, tail probes stop behind their first FIND_SFD step
    bool early_probe() const { return early; }
    int decoupled_mode = 0;  // 1: every pass decoupled (header-only segment jobs + payload pass)
    bool skip = false;
    uint32_t n_payload = 0, n_rerun = 0;
    bool decoupled(size_t n) const { return decoupled_mode == 1 || (decoupled_mode == 2 && n <= (two_per_cu ? slots / 2u : slots)); }
    void set_skip_payload(bool on) { skip = on; }
    uint32_t n_moved = 0, n_pending = 0;
    void count_payload(uint32_t p, uint32_t m, uint32_t r) { n_payload += p; n_moved += m; n_rerun += r; }
    // the payload pass, by the oracle: the packet decoded from its header by the complete state machine - its frame, and how far from the zero-drift
    // end it ended (the device finds the same two things by demodulating the symbols on their own and following their d_fine_sync)
    void abort_payload() {}
    int run_payload_begin(std::vector<PayloadReq> &) { return 0; }
    int run_payload_end(std::vector<PayloadReq> &reqs)
    {
        std::vector<oracle_attempt_t> tmp(2);
        for (PayloadReq &q : reqs) {
            oracle_job_result_t r{};
            lora_oracle_run_job(o, iq + 2 * q.stream_off, (size_t)q.stream_len, q.hdr_pos, q.hdr_pos + 1, q.cr_prev, 1, 8, 1, tmp.data(), &r);
            const oracle_attempt_t &t = tmp[0];
            q.frame_len = 0; q.end_shift = 0;
            if (r.n_attempts == 1u && !r.pad && t.status == 1u) {
                q.status = kPayloadDecoded;
                q.end_shift = (int32_t)(t.end_pos - (q.start + (int64_t)q.n_walk * (int64_t)sps_));
                q.frame_len = t.frame_len;
                std::memcpy(q.frame, t.frame, t.frame_len);
            } else {
                q.status = kPayloadOutOfData;
                n_pending++;
            }
            if (payload_force_rerun && ((n_forced++ % payload_force_rerun) == 0u)) q.status = kPayloadUnresolved;
        }
        return 0;
    }
    uint32_t payload_force_rerun = 0, n_forced = 0; // tests: every n-th packet is reported as not clean

    uint32_t sps() const { return sps_; }
    uint32_t ctor_cr() const { return ctor_cr_; }
    uint32_t segment_symbols() const { return seg_symbols; }
    uint32_t resident_slots() const { return slots; }
    uint32_t resident_slots_alt() const { return two_per_cu ? slots / 2u : 0u; } // (the device: walker_resident_slots_full - one workgroup per CU where the kernel fits a CU twice)
    bool two_per_cu = false;
    bool tracing() const { return false; }
    bool implicit() const { return false; }
    bool quiet_edges(const std::vector<StreamDesc> &streams, std::vector<std::vector<int64_t>> &edges)
    { // same rule as envelope_kernel / edges_kernel, from every sample of the block
        if (!burst_plan) return false;
        edges.assign(streams.size(), {});
        for (size_t i = 0; i < streams.size(); i++) {
            const float *x = iq + 2 * streams[i].off;
            const size_t nb = streams[i].len / sps_;
            std::vector<double> E(nb);
            double sum = 0;
            for (size_t b = 0; b < nb; b++) {
                double e = 0;
                for (size_t k = 2 * b * sps_; k < 2 * (b + 1) * sps_; k++) e += (double)x[k] * x[k];
                E[b] = e; sum += e;
            }
            (void)sum;
            auto quiet = [&](size_t b) { // under half the largest energy within 8 blocks
                double m = 0;
                for (size_t q = b >= 8 ? b - 8 : 0; q <= b + 8 && q < nb; q++) m = std::max(m, E[q]);
                return E[b] < 0.5 * m;
            };
            for (size_t b = 1; b < nb; b++)
                if (quiet(b) && !quiet(b - 1)) edges[i].push_back((int64_t)(b * sps_));
        }
        return true;
    }
    void note_plan(bool ok, size_t) { planned += ok ? 1u : 0u; }
    int run_jobs_begin(const std::vector<Job> &jobs, uint32_t rpj, uint32_t tc, RunOut &out) { return run_jobs(jobs, rpj, tc, out); } // (no device: runs at once)
    int run_jobs_end(RunOut &) { return 0; }
    int run_jobs(const std::vector<Job> &jobs, uint32_t rpj, uint32_t, RunOut &out)
    {
        out.rpj = rpj; out.cap = rpj;
        out.res.assign(jobs.size(), JobResult{});
        out.recs.assign(jobs.size() * (size_t)rpj, AttemptRec{});
        std::vector<oracle_attempt_t> tmp(rpj + 1);
        auto put_recs = [&](size_t j, uint32_t first, uint32_t n) {
            for (uint32_t a = 0; a < n && first + a < rpj; a++) {
                AttemptRec &d = out.recs[j * (size_t)rpj + first + a];
                const oracle_attempt_t &s = tmp[a];
                d.start_pos = s.start_pos; d.trig_pos = s.trig_pos; d.hdr_pos = s.hdr_pos; d.end_pos = s.end_pos;
                d.status = s.status; d.npush = s.npush; std::memcpy(d.push_tail, s.push_tail, sizeof d.push_tail);
                d.cr_prev = s.cr_prev; d.hdr_ambig = s.hdr_ambig; d.frame_len = s.frame_len; d.n_symbols = s.n_symbols;
                d.n_sfd = early ? s.n_sfd : 0u; std::memcpy(d.sfd_pos, s.sfd_pos, sizeof d.sfd_pos); std::memcpy(d.sfd_fails, s.sfd_fails, sizeof d.sfd_fails);
                std::memcpy(d.frame, s.frame, s.frame_len); // (status 6: the 16 bytes of SkippedPayload, same layout)
            }
        };
        for (size_t j = 0; j < jobs.size(); j++) {
            const Job &jb = jobs[j];
            oracle_job_result_t r{};
            lora_oracle_run_job(o, iq + 2 * jb.stream_off, (size_t)jb.stream_len, jb.start, jb.scan_limit, jb.cr_prev,
                                jb.max_attempts, (int)jb.stop_at_header | (skip ? 4 : 0) | (jb.start_at_header ? 8 : 0), rpj, tmp.data(), &r);
            JobResult &jr = out.res[j];
            jr.final_pos = r.final_pos; jr.n_attempts = r.n_attempts; jr.final_cr = r.final_cr; jr.npush = r.npush;
            std::memcpy(jr.push_tail, r.push_tail, sizeof jr.push_tail);
            jr.stop_reason = r.stop_reason; jr.pad = r.pad;
            put_recs(j, 0, r.n_attempts);
            // tail probe, with the device's semantics: having reached its scan limit the job goes on as a fresh probe job
            if (tail_probes && jb.probe_limit > jb.scan_limit && r.stop_reason == 0u && !r.pad) {
                oracle_job_result_t t{};
                const uint32_t first = r.n_attempts, cap = rpj > first ? rpj - first : 0u;
                lora_oracle_run_job(o, iq + 2 * jb.stream_off, (size_t)jb.stream_len, r.final_pos, jb.probe_limit, r.final_cr, 0, jb.tail_stop_sfd ? 3 : 1, cap,
                                    tmp.data(), &t);
                n_early += (t.pad && t.n_attempts && t.n_attempts <= cap && tmp[t.n_attempts - 1].status == 5u) ? 1u : 0u;
                jr.tail_valid = 1; jr.tail_first_rec = first; jr.tail_final_pos = t.final_pos; jr.tail_n_attempts = t.n_attempts;
                jr.tail_final_cr = t.final_cr; jr.tail_npush = t.npush; std::memcpy(jr.tail_push_tail, t.push_tail, sizeof jr.tail_push_tail);
                jr.tail_stop_reason = t.stop_reason; jr.tail_pad = t.pad;
                put_recs(j, first, t.n_attempts);
                n_tails++;
            }
        }
        return 0;
    }
    void publish(const AttemptRec &r, StreamDesc &sd)
    { // same blob as lora_runtime.cpp::publish
        SimFrame f;
        f.blob.assign(15 + r.frame_len, 0);
        f.blob[13] = lora_oracle_snr_byte(sd.pwr.snr);
        std::memcpy(f.blob.data() + 15, r.frame, r.frame_len);
        f.hdr_pos = sd.abs_base + r.hdr_pos;
        frames.push_back(std::move(f));
    }
    void append_trace(const RunOut &, uint32_t, uint32_t, int64_t) {}
    void count_jobs(uint32_t n) { n_jobs += n; }
    void count_probes(uint32_t n) { n_probes += n; }
    void count_slow_path() { n_slow++; }
    void count_repair() {}
    double walker_ms() const { return 0.0; }
};
} // namespace

extern "C" int stitch_sim_decode(const float *iq, size_t n_items, int sf, int ctor_cr, int crc, int reduced_rate, int demod,
                                 uint32_t segment_symbols, uint32_t resident_slots, int tail_probes, uint8_t *out, size_t cap, int *lens,
                                 long long *hdr_pos, int max_frames, uint32_t *stats)
{
    lora_oracle_t *o = lora_oracle_create(1e6f, 125000, (uint8_t)sf, 0, (uint8_t)ctor_cr, crc, reduced_rate, 0, demod);
    if (!o) return -1;
    SimEnv env{o, iq, n_items, lora_oracle_sps(o), (uint32_t)ctor_cr, segment_symbols, resident_slots};
    env.tail_probes = (tail_probes & 1) != 0;
    env.burst_plan = (tail_probes & 2) != 0;
    env.early = (tail_probes & 4) != 0;
    env.decoupled_mode = (tail_probes & 8) ? 1 : (tail_probes & 16) ? 2 : 0; // 2: the device's per-pass rule (the jobs fit the device at once)
    env.two_per_cu = (tail_probes & 32) != 0;
    env.payload_force_rerun = (uint32_t)(tail_probes >> 8) & 0xffu;
    std::vector<StreamDesc> sds(1);
    sds[0].off = 0; sds[0].len = n_items; sds[0].id = 0; sds[0].cr_in = (uint32_t)ctor_cr; sds[0].abs_base = 0;
    const int rc = decode_streams(env, sds);
    lora_oracle_destroy(o);
    if (rc != 0) return -2;
    size_t used = 0;
    int n = 0;
    for (const SimFrame &f : env.frames) {
        if (n >= max_frames || used + f.blob.size() > cap) return -3;
        std::memcpy(out + used, f.blob.data(), f.blob.size());
        lens[n] = (int)f.blob.size(); hdr_pos[n] = f.hdr_pos; used += f.blob.size(); n++;
    }
    stats[0] = env.n_jobs; stats[1] = env.n_probes; stats[2] = env.n_slow; stats[3] = sds[0].incomplete ? 1u : 0u; stats[4] = env.n_tails; stats[5] = env.planned; stats[6] = env.n_early; stats[7] = env.n_payload; stats[8] = env.n_rerun; stats[9] = env.n_moved; stats[10] = env.n_pending;
    return n;
}

// plan_burst_segments on its own: edges[] holds the gap starts of all streams back to back (n_edges[i] of them for stream i);
// cuts_out receives the cuts the same way, n_cuts[i] per stream.  Returns 1 when a burst-aware plan was made, 0 for the fixed grid.
extern "C" int stitch_sim_plan(const long long *lens, const int *n_edges, const long long *edges, int n_streams, uint32_t sps, uint32_t slots,
                               unsigned long long nominal, long long *cuts_out, int cap, int *n_cuts)
{
    std::vector<StreamDesc> sds(n_streams);
    std::vector<std::vector<int64_t>> e(n_streams), cuts;
    size_t k = 0;
    for (int i = 0; i < n_streams; i++) {
        sds[i].off = 0; sds[i].len = (uint64_t)lens[i]; sds[i].id = (uint32_t)i;
        for (int q = 0; q < n_edges[i]; q++) e[i].push_back(edges[k++]);
    }
    const bool ok = plan_burst_segments(sds, e, sps, slots, nominal, cuts);
    int used = 0;
    for (int i = 0; i < n_streams; i++) {
        n_cuts[i] = ok ? (int)cuts[i].size() : 0;
        if (!ok) continue;
        for (int64_t c : cuts[i]) { if (used >= cap) return -1; cuts_out[used++] = c; }
    }
    return ok ? 1 : 0;
}
