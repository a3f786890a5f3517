// lora_runtime.cpp -- host side below the C ABI (include/lora_hip.h):
// table construction (the reference constructor's work), job scheduling over
// streams / stream segments, speculation stitching, frame assembly, streaming.
//
// Citations `:NNN` are lines of the reference's lib/decoder_impl.cc.

#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <cstddef>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <chrono>
#include <deque>
#include <new>
#include <string>
#include <vector>

#include "../../include/lora_hip.h"
#include "lora_device.h"
#include "lora_stitch.hpp"

using namespace lora_hip;

namespace {

constexpr int kLoratapLen = 15; // sizeof(loratap_header_t), include/lora/loratap.h:35-55
constexpr uint32_t kWorkBudget = 64u * 1024u + 64u; // LDS work area cap (bytes)

// Published frames wait here for poll/drain: blobs back to back in one byte arena (no allocation per frame).
struct FrameQueue {
    struct Ref { size_t off; lora_hip_frame_info_t info; };
    std::vector<uint8_t> bytes;
    std::vector<Ref> refs;
    size_t head = 0; // first frame not yet handed out
    size_t size() const { return refs.size() - head; }
    bool empty() const { return head == refs.size(); }
    uint8_t *push(uint32_t len, const lora_hip_frame_info_t &info)
    {
        const size_t off = bytes.size();
        bytes.resize(off + len, 0);
        refs.push_back(Ref{off, info});
        return bytes.data() + off;
    }
    const Ref &front() const { return refs[head]; }
    const uint8_t *front_bytes() const { return bytes.data() + refs[head].off; }
    void pop()
    {
        if (++head == refs.size()) { refs.clear(); bytes.clear(); head = 0; }
    }
};

// Page-locked host staging buffer, grown on demand.
template <typename T>
struct PinnedBuf {
    T *p = nullptr;
    size_t cap = 0;
    hipError_t reserve(size_t n)
    {
        if (n <= cap) return hipSuccess;
        if (p) (void)hipHostFree(p);
        p = nullptr; cap = 0;
        const size_t want = n + n / 4 + 16;
        hipError_t e = hipHostMalloc((void **)&p, want * sizeof(T), hipHostMallocDefault);
        if (e == hipSuccess) cap = want;
        return e;
    }
    void release() { if (p) (void)hipHostFree(p); p = nullptr; cap = 0; }
};

// loratap_header.rssi.snr = (uint8_t)(10.0f * log10(d_snr) + 0.5) (:597).  The
// narrowing is UB out of range upstream; pinned to the x86-64 result (cvttsd2si).
uint8_t snr_byte(float snr)
{
    const double v = (double)(10.0f * log10f(snr)) + 0.5;
    if (!(v > -2147483649.0 && v < 2147483648.0)) return 0;
    return (uint8_t)(int32_t)v;
}

template <typename T>
struct DevBuf {
    T *p = nullptr;
    size_t cap = 0;
    hipError_t reserve(size_t n)
    {
        if (n <= cap) return hipSuccess;
        if (p) (void)hipFree(p);
        p = nullptr; cap = 0;
        const size_t want = n + n / 4 + 16;
        hipError_t e = hipMalloc((void **)&p, want * sizeof(T));
        if (e == hipSuccess) cap = want;
        return e;
    }
    void release() { if (p) (void)hipFree(p); p = nullptr; cap = 0; }
};

} // namespace

struct lora_hip_decoder {
    lora_hip_config_t cfg{};
    DevParams P{};
    int device = 0;
    // device tables
    float2 *d_down = nullptr, *d_twN = nullptr, *d_tws = nullptr;
    float *d_wave_tabs = nullptr;
    float2 *d_w3_tw = nullptr, *d_w3_ctab = nullptr;
    float2 *d_team_tabs = nullptr;
    float *d_up_ifreq = nullptr, *d_down_ifreq = nullptr, *d_up_ifreq_v = nullptr;
    std::vector<float2> h_up;          // d_upchirp (:160): no kernel reads it; kept for lora_hip_get_table
    size_t n_up_ifreq_v = 0;           // floats in d_up_ifreq_v (3 sps + the guard tail)
    // per-pass buffers
    DevBuf<Job> d_jobs;
    DevBuf<JobResult> d_results;
    DevBuf<AttemptRec> d_recs;
    DevBuf<float> d_scratch;
    DevBuf<StepRec> d_trace;
    DevBuf<float2> d_staging;
    DevBuf<int64_t> d_offsets;
    DevBuf<uint32_t> d_bins;
    RunOut run_out[2];                 // results of the scheduler's two launches, reused between calls
    std::vector<JobResult> h_results;
    std::vector<AttemptRec> h_recs;
    std::vector<StepRec> h_trace;
    hipEvent_t ev0 = nullptr, ev1 = nullptr, ev_done = nullptr;
    hipEvent_t ev_pre0 = nullptr, ev_pre1 = nullptr, ev_dep = nullptr; // envelope pre-pass timing; dependency of the pre-pass stream on the caller's
    hipStream_t pre_stream = nullptr;  // the envelope pre-pass runs here, beside whatever the caller's stream is busy with
    // the main launch of a pass between run_jobs_begin and run_jobs_end
    struct Pending { uint32_t nj = 0, recs_per_job = 0, eager = 0, trace_cap = 0; bool direct = true, open = false; hipStream_t st = nullptr;
                     std::chrono::steady_clock::time_point hp0, hp1; } pending;
    // a pass between lora_hip_decode_device_begin and _end
    PassCtx pass;
    std::vector<StreamDesc> pass_streams;
    const float2 *pass_iq = nullptr;
    hipStream_t pass_st = nullptr;
    bool pass_open = false, iq_ready = false;
    // an envelope pre-pass between quiet_edges_enqueue and quiet_edges_collect
    bool pre_issued = false;
    const float2 *pre_iq = nullptr;
    uint64_t pre_sig = 0;
    uint32_t pre_ns = 0;
    std::chrono::steady_clock::time_point pre_hq0;
    // outputs
    FrameQueue frames;
    PinnedBuf<Job> p_jobs;             // staging for run_jobs: jobs up, results and the first attempt records down
    PinnedBuf<JobResult> p_res;
    PinnedBuf<AttemptRec> p_recs;
    // burst-envelope pre-pass (segment planning)
    DevBuf<uint32_t> d_balance;             // LaunchCfg::balance, zeroed when allocated
    DevBuf<float> d_env_E;
    DevBuf<unsigned long long> d_env_buf;   // gap-start bitmap, one bit per block
    PinnedBuf<EnvStream> p_env_streams;
    PinnedBuf<unsigned long long> p_env_buf;
    float envelope_ms = 0.0f;
    std::vector<lora_hip_step_t> trace;
    lora_hip_timing_t timing{};
    std::string err;
    // streaming state (lora_hip_work): see the pipeline description above stream_rotate()
    struct StreamPipe {
        DevBuf<float2> dbuf[2];       // [ tail (up to tailcap items, right-aligned) | chunk (batch items) ]
        size_t tailcap = 0;
        int cur = 0;                  // buffer whose chunk is being filled
        size_t fill = 0;              // items of the current chunk uploaded (or queued for upload) so far
        size_t tail_len = 0;          // items carried over in front of the current chunk
        hipStream_t copy_st = nullptr, comp_st = nullptr;
        hipEvent_t up_ev = nullptr, tail_ev = nullptr;
        bool inflight = false;        // a pass is running on dbuf[cur ^ 1]
        size_t fl_off = 0, fl_len = 0;
        PinnedBuf<float2> stage[2];   // bounce buffers for caller memory that is not page-locked
        hipEvent_t stage_ev[2] = {nullptr, nullptr};
        bool stage_busy[2] = {false, false};
        int stage_i = 0;
        std::vector<std::pair<uintptr_t, uintptr_t>> pinned; // caller ranges registered with hipHostRegister (DMA straight from them)
        std::vector<std::pair<uintptr_t, uintptr_t>> refused; // ranges the runtime would not register: do not ask again
        uint64_t bytes_direct = 0, bytes_staged = 0;
        // latency bound (lora_hip_set_stream_latency): a pass is launched when the oldest unlaunched sample has waited this long
        float max_latency_ms = 50.0f;
        std::chrono::steady_clock::time_point t_first; // arrival of the first item of the chunk being filled
        bool have_first = false;
        uint64_t passes = 0, passes_by_latency = 0;
    } sp;
    int64_t host_base = 0;      // absolute item index of the first item of the stream region of the next pass
    uint32_t stream_cr = 0;
    PwrState stream_pwr;
    size_t batch_items = 0;
    const char *last_kernel = nullptr; // name of the walker kernel the last pass's main launch ran
    uint32_t last_kernel_jobs = 0;     // ... and its job count (reset when a pass begins)
    uint32_t resident_slots = 0;
    uint32_t resident_slots_alt = ~0u; // walker_resident_slots_full: the one-workgroup-per-CU slot count where the kernel family has a second geometry (walker3 SF9 / SF10: the full-size kernel beside the half-size one; walker2 SF7 / SF8: the CU count - the plan for passes with fewer bursts than two-per-CU slots), else 0
    uint32_t eager_recs = 4;
    uint32_t last_plan_burst = 0, last_plan_segments = 0;
    // decoupled passes (lora_stitch.hpp payload_round): header-only segment jobs + the payload pass
    int decoupled_policy = -1;         // -1 auto (few jobs for the device, few re-runs lately), 0 never, 1 whenever the kernels allow it
    bool launch_skip = false;          // the next launches run the header-only kernel variant (LaunchCfg.skip_payload)
    uint32_t dec_backoff = 0;          // auto: passes to sit out after one whose packets mostly had to be run again
    uint32_t last_payload_packets = 0, last_payload_rerun = 0, last_payload_symbols = 0, last_payload_moved = 0, last_payload_rounds = 0;
    float last_payload_ms = 0.0f;
    PinnedBuf<int64_t> p_pay_off;
    PinnedBuf<PayloadDesc> p_pay_desc;
    PinnedBuf<PayloadOut> p_pay_out;
    DevBuf<int32_t> d_fine, d_alt_shift, d_alt_fine;
    DevBuf<uint32_t> d_alt_bins;
    hipStream_t pay_stream = nullptr;  // the payload pass runs here, beside the explicit probes of the same pass on the caller's stream
    hipEvent_t ev_pay0 = nullptr, ev_pay1 = nullptr, ev_pay_done = nullptr;
    struct PayState { std::vector<uint32_t> active; size_t used = 0, cap_sym = 0, n_sym = 0; int round = 0; float ms = 0.0f; bool open = false; } pay;
};

namespace {

thread_local std::string g_create_err; // lora_hip_last_error(NULL): why the last lora_hip_create failed

lora_hip_status fail(lora_hip_decoder *h, lora_hip_status s, const char *fmt, ...)
{
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    if (h) h->err = buf;
    g_create_err = buf;
    return s;
}

#define HIP_TRY(h, expr)                                                                                   \
    do {                                                                                                   \
        hipError_t e__ = (expr);                                                                           \
        if (e__ != hipSuccess) return fail(h, LORA_HIP_ERR_HIP, "%s: %s", #expr, hipGetErrorString(e__)); \
    } while (0)

// instantaneous_frequency (:224-244), used for the constructor's tables
void host_ifreq(const float2 *x, float *out, uint32_t window)
{
    for (uint32_t i = 1; i < window; i++) {
        const float p1 = std::atan2(x[i - 1].y, x[i - 1].x);
        float p2 = std::atan2(x[i].y, x[i].x);
        while ((double)(p2 - p1) > M_PI) p2 = (float)((double)p2 - 2.0 * M_PI);
        while ((double)(p2 - p1) < -M_PI) p2 = (float)((double)p2 + 2.0 * M_PI);
        out[i - 1] = p2 - p1;
    }
    out[window - 1] = out[window - 2];
}

template <typename T>
lora_hip_status upload(lora_hip_decoder *h, T **dst, const std::vector<T> &src)
{
    HIP_TRY(h, hipMalloc((void **)dst, src.size() * sizeof(T)));
    HIP_TRY(h, hipMemcpy(*dst, src.data(), src.size() * sizeof(T), hipMemcpyHostToDevice));
    return LORA_HIP_OK;
}

// decoder_impl constructor (:49-122) + build_ideal_chirps (:141-175)
lora_hip_status build_tables(lora_hip_decoder *h)
{
    const lora_hip_config_t &c = h->cfg;
    DevParams &P = h->P;
    const uint32_t samples_per_second = (uint32_t)c.samp_rate;          // :74
    const double dt = (double)(1.0f / (float)samples_per_second);       // :77 (float division)
    const double symbols_per_second = (double)c.bandwidth / (double)(1u << c.sf); // :80
    const uint32_t sps = (uint32_t)((double)samples_per_second / symbols_per_second); // :83
    const uint32_t N = 1u << c.sf;
    if (sps < N || (sps & (sps - 1u)) != 0u)
        return fail(h, LORA_HIP_ERR_BAD_CONFIG, "samples per symbol (%u) must be a power-of-two multiple of 2^sf (%u)", sps, N);
    const uint32_t D = sps / N;
    if (D > 16u) return fail(h, LORA_HIP_ERR_BAD_CONFIG, "decimation %u > 16 is not supported", D);
    P.sf = c.sf; P.nbins = N; P.nbins_hdr = 1u << (c.sf - 2); P.sps = sps; P.decim = D; P.samples_per_second = samples_per_second;
    P.log_nbins = c.sf; P.delay_after_sync = sps / 4u;
    P.implicit = c.implicit ? 1u : 0u; P.reduced_rate = c.reduced_rate ? 1u : 0u;
    P.enable_fine_sync = c.disable_drift_correction ? 0u : 1u;
    P.demod_mode = (uint32_t)c.demod;
    P.ctor_cr = c.cr & 7u; P.ctor_crc = c.crc ? 1u : 0u;
    // walker2 (wave-per-symbol rounds) where it applies; LORA_HIP_NO_FAST=1 forces the generic kernels (diagnostics)
    P.use_fast = getenv("LORA_HIP_NO_FAST") ? 0u : 1u;
    // get_shift_fft working set: D/G polyphase rows of N points (+1 pad) must fit the LDS budget
    uint32_t G = 1;
    while ((size_t)(D / G) * (N + 1u) * sizeof(float2) > kWorkBudget && G < D) G <<= 1;
    P.fft_groups = G;
    P.fft_stride = (D / G > 1u) ? N + 1u : N;
    uint32_t work = (uint32_t)((size_t)(D / G) * P.fft_stride * sizeof(float2));
    if (work < 2u * sps * sizeof(float) && 2u * sps * sizeof(float) <= kWorkBudget) work = 2u * sps * (uint32_t)sizeof(float);
    if (work < sps * sizeof(float) && sps * sizeof(float) <= kWorkBudget) work = sps * (uint32_t)sizeof(float);
    work = (work + 15u) & ~15u;
    P.lds_work_bytes = work;
    P.ifreq_in_lds_1 = (sps * sizeof(float) <= work) ? 1u : 0u;
    P.ifreq_in_lds_2 = (2u * sps * sizeof(float) <= work) ? 1u : 0u;

    std::vector<float2> down(sps), up(sps), twN(N / 2u > 0 ? N / 2u : 1u), tws(sps);
    const double T = -0.5 * c.bandwidth * symbols_per_second, f0 = c.bandwidth / 2.0, pre_dir = 2.0 * M_PI;
    for (uint32_t i = 0; i < sps; i++) {
        const double t = dt * i;
        const float pd = (float)(pre_dir * t * (f0 + T * t));          // gr_expj takes a float (:159)
        const float pu = (float)(pre_dir * t * (f0 + T * t) * -1.0f);  // :160
        const float cd = std::cos(pd), sd = std::sin(pd), cu = std::cos(pu), su = std::sin(pu);
        down[i] = make_float2(cd - sd, sd + cd);                       // (1+1j) * expj
        up[i] = make_float2(cu - su, su + cu);
    }
    std::vector<float> down_ifreq(sps), up_ifreq(sps), up3(3u * (size_t)sps + 4u * D + 8u);
    host_ifreq(down.data(), down_ifreq.data(), sps);
    host_ifreq(up.data(), up_ifreq.data(), sps);
    {
        std::vector<float2> tmp(3u * (size_t)sps);
        for (int r = 0; r < 3; r++) std::copy(up.begin(), up.end(), tmp.begin() + (size_t)r * sps);
        host_ifreq(tmp.data(), up3.data(), 3u * sps);
        // guard tail for bin_idx = N-1 (reference indexes past the vector there, :301,:310)
        for (size_t i = 3u * (size_t)sps; i < up3.size(); i++) up3[i] = up3[3u * (size_t)sps - 1u];
    }
    { // closed-form fine_sync of the wave demodulators (DevParams::ffs_*): slope, step and noise bound of d_upchirp_ifreq_v as built above
        P.ffs_on = 0u; P.ffs_alpha = 0.0f; P.ffs_jump = 0.0f; P.ffs_tol = 0.0f;
        if (D == 8u && c.sf >= 7u && c.sf <= 12u && !getenv("LORA_HIP_NO_FFS")) {
            const size_t S = sps, lo = S + 7u, hi = 3u * S - 8u; // every dV index a lag difference can touch for bin_idx < N-1: [S+7, 3S-9]
            auto dV = [&](size_t m) { return (double)up3[m + 1u] - (double)up3[m]; };
            double sum = 0.0;
            bool ok = std::fabs(dV(2u * S - 1u)) > 0.5;
            for (size_t m = lo; m < hi; m++) if (m != 2u * S - 1u) { sum += dV(m); ok = ok && std::fabs(dV(m)) < 0.01; }
            const double alpha = sum / (double)(hi - lo - 1u);
            std::vector<double> cs(hi - lo + 1u, 0.0); // prefix sums of the squared deviations from the slope
            for (size_t m = lo; m < hi; m++) { const double e = (m == 2u * S - 1u) ? 0.0 : dV(m) - alpha; cs[m - lo + 1u] = cs[m - lo] + e * e; }
            double worst = 0.0;
            for (size_t o = lo; o + S <= hi; o++) worst = std::max(worst, cs[o + S - lo] - cs[o - lo]);
            // what the table's float noise e[] adds to a lag difference is sum_k ifreq[k] e[o + k].  SF7 / SF8: Cauchy-Schwarz with |ifreq| <= pi - a bound for
            // ANY window.  SF9 and up: the kernel vouches only for windows whose every |ifreq[k]| stays below pi / 2 (SF9, SF10) or atan(1/2) (SF11, SF12;
            // a clean chirp at decimation 8 stays below pi / 8): the same bound with that ceiling - and where even that exceeds what a decision has to spare
            // (SF10 and up: the table's own rounding noise grows with the chirp's phase) twelve standard deviations of the sum under random signs of e[], which
            // is what the model and the decision-flip sweeps hold to the oracle (tools/ffs_model.py, tools/decision_flip_sweep.py); + the evaluation's own rounding
            const double enorm = std::sqrt(worst), fmax = c.sf >= 11u ? std::atan(0.5) : c.sf >= 9u ? M_PI / 2.0 : M_PI; // (kFfsClass, lora_wave_demod.inc.hip)
            const double fnorm = c.sf >= 9u ? std::sqrt((double)S * fmax * fmax + 8.0 * M_PI * M_PI) : fmax * std::sqrt((double)S); // (the four products at either end of a window: any value)
            double tol = 1.05 * fnorm * enorm + 1.0e-4;
            if (c.sf >= 9u && tol > 0.15) tol = 12.0 * fmax * enorm + 1.0e-4;
            if (ok && tol < 0.2) { P.ffs_on = 1u; P.ffs_alpha = (float)alpha; P.ffs_jump = (float)(dV(2u * S - 1u) - alpha); P.ffs_tol = (float)tol; }
            if (getenv("LORA_HIP_DEBUG")) fprintf(stderr, "[lora_hip] closed-form fine_sync: SF%u alpha %.6g jump %.6g |e| %.4g tol %.4g -> %s\n", (unsigned)c.sf, alpha, dV(2u * S - 1u) - alpha, enorm, tol, P.ffs_on ? "on" : "off");
        }
    }
    { // chirp_avg and stddev of the ideal downchirp ifreq over sps-1 points (:287-289, :415-425)
        const uint32_t n = sps - 1u;
        float s = 0.0f;
        for (uint32_t i = 0; i < n; i++) s += down_ifreq[i];
        const float avg = s / (float)n;
        float var = 0.0f;
        for (uint32_t i = 0; i < n; i++) { const float t = down_ifreq[i] - avg; var += t * t; }
        var /= (float)n;
        P.down_ifreq_avg = avg;
        P.down_ifreq_sd = std::sqrt(var);
        float ds = 0.0f;
        for (uint32_t i = 0; i < n; i++) ds += down_ifreq[i] - avg;
        P.down_ifreq_dsum = ds;
        // line through the ideal upchirp ifreq (a linear ramp up to float noise) for the closed-form SYNC
        double sk = 0, su = 0, skk = 0, sku = 0;
        for (uint32_t k = 0; k < n; k++) { sk += k; su += up_ifreq[k]; skk += (double)k * k; sku += (double)k * up_ifreq[k]; }
        P.sync_b = (n * sku - sk * su) / (n * skk - sk * sk);
        P.sync_a = (su - P.sync_b * sk) / n;
        P.sync_closed_form = sps >= 4096u ? 1u : 0u;
        // near-tied SYNC shifts decided by the reference's own float sums (lora_strict_sync.inc.hip): on unless the caller opts out
        P.strict_sync = (c.flags & LORA_HIP_FLAG_FAST_SYNC) ? 0u : 1u;
        if (const char *e = getenv("LORA_HIP_STRICT_SYNC")) P.strict_sync = (e[0] != '0') ? 1u : 0u; // (A/B runs)
        h->decoupled_policy = (c.flags & LORA_HIP_FLAG_NO_DECOUPLED) ? 0 : -1;
        if (const char *e = getenv("LORA_HIP_DECOUPLED")) h->decoupled_policy = e[0] == '0' ? 0 : e[0] == '1' ? 1 : -1; // 0 never, 1 always, anything else: auto
    }
    for (uint32_t t = 0; t < N / 2u; t++) {
        const double a = -2.0 * M_PI * (double)t / (double)N;
        twN[t] = make_float2((float)std::cos(a), (float)std::sin(a));
    }
    for (uint32_t m = 0; m < sps; m++) {
        const double a = -2.0 * M_PI * (double)m / (double)sps;
        tws[m] = make_float2((float)std::cos(a), (float)std::sin(a));
    }
    lora_hip_status s;
    if ((s = upload(h, &h->d_down, down)) != LORA_HIP_OK) return s;
    if ((s = upload(h, &h->d_twN, twN)) != LORA_HIP_OK) return s;
    if ((s = upload(h, &h->d_tws, tws)) != LORA_HIP_OK) return s;
    if ((s = upload(h, &h->d_up_ifreq, up_ifreq)) != LORA_HIP_OK) return s;
    if ((s = upload(h, &h->d_down_ifreq, down_ifreq)) != LORA_HIP_OK) return s;
    if ((s = upload(h, &h->d_up_ifreq_v, up3)) != LORA_HIP_OK) return s;
    h->h_up = up; h->n_up_ifreq_v = up3.size();
    P.down = h->d_down; P.twN = h->d_twN; P.tws = h->d_tws;
    P.wave_tabs = nullptr;
    if (D == 8u && wave_tables_floats(c.sf) != 0u) { // packed twiddle block of the wave demodulator
        std::vector<float> wt(wave_tables_floats(c.sf));
        build_wave_tables(c.sf, down.data(), wt.data());
        if ((s = upload(h, &h->d_wave_tabs, wt)) != LORA_HIP_OK) return s;
        P.wave_tabs = h->d_wave_tabs;
    }
    if (wave_tables_floats_d(c.sf, D) != 0u) { // ... and of its decimation 2 / 4 siblings
        std::vector<float> wt(wave_tables_floats_d(c.sf, D));
        build_wave_tables_d(c.sf, D, down.data(), wt.data());
        if ((s = upload(h, &h->d_wave_tabs, wt)) != LORA_HIP_OK) return s;
        P.wave_tabs = h->d_wave_tabs;
    }
    P.w3_tw = nullptr; P.w3_ctab = nullptr;
    if (D == 8u && walker3_covers(c.sf)) { // tables of the workgroup-per-symbol walker (SF9-12)
        std::vector<float2> tw(w3_tw_entries(c.sf)), ct(sps);
        build_w3_tables(c.sf, tw.data(), ct.data());
        if ((s = upload(h, &h->d_w3_tw, tw)) != LORA_HIP_OK) return s;
        if ((s = upload(h, &h->d_w3_ctab, ct)) != LORA_HIP_OK) return s;
        P.w3_tw = h->d_w3_tw; P.w3_ctab = h->d_w3_ctab;
    }
    P.team_tabs = nullptr;
    if (D == 8u && team_tables_entries(c.sf) != 0u) { // SF10-12: the team demodulator's table block
        std::vector<float2> tt(team_tables_entries(c.sf));
        build_team_tables(c.sf, down.data(), dt, (double)c.bandwidth, tt.data());
        if ((s = upload(h, &h->d_team_tabs, tt)) != LORA_HIP_OK) return s;
        P.team_tabs = reinterpret_cast<const float *>(h->d_team_tabs);
    }
    P.up_ifreq = h->d_up_ifreq; P.down_ifreq = h->d_down_ifreq; P.up_ifreq_v = h->d_up_ifreq_v;
    return LORA_HIP_OK;
}

// Publishes the frame of one completed attempt (msg_lora_frame, :588-609).
void publish(lora_hip_decoder *h, const AttemptRec &r, StreamDesc &sd)
{
    lora_hip_frame_info_t info;
    info.stream = sd.id;
    info.length = (uint32_t)kLoratapLen + r.frame_len;
    info.header_pos = sd.abs_base + r.hdr_pos;
    info.end_pos = sd.abs_base + r.end_pos;
    uint8_t *blob = h->frames.push(info.length, info); // zero-filled loratap header
    blob[13] = snr_byte(sd.pwr.snr);                   // loratap_header.rssi.snr, byte offset 13
    std::memcpy(blob + kLoratapLen, r.frame, r.frame_len);
}

// Launches the walker over a set of jobs (run_jobs_begin) and brings the results back to the host (run_jobs_end).
lora_hip_status run_jobs_begin(lora_hip_decoder *h, const float2 *d_iq, const std::vector<Job> &jobs, uint32_t recs_per_job,
                               uint32_t trace_cap, hipStream_t st, RunOut &out)
{
    const uint32_t nj = (uint32_t)jobs.size();
    const auto hp0 = std::chrono::steady_clock::now();
    if (h->pending.open) return fail(h, LORA_HIP_ERR_INTERNAL, "a launch is already in flight on this handle");
    h->pending = lora_hip_decoder::Pending{};
    h->pending.nj = nj; h->pending.recs_per_job = recs_per_job; h->pending.trace_cap = trace_cap; h->pending.st = st; h->pending.hp0 = hp0;
    out.rpj = recs_per_job; out.cap = recs_per_job;
    out.res.resize(nj); // (overwritten from the landing buffer below)
    out.recs.clear();
    if (nj == 0) return LORA_HIP_OK;
    static const bool staged_bufs = getenv("LORA_HIP_STAGED") != nullptr;
    if (staged_bufs) {
        HIP_TRY(h, h->d_jobs.reserve(nj));
        HIP_TRY(h, h->d_results.reserve(nj));
        HIP_TRY(h, h->d_recs.reserve((size_t)nj * recs_per_job));
    }
    const bool need_scratch = !h->P.ifreq_in_lds_2;
    if (need_scratch) HIP_TRY(h, h->d_scratch.reserve((size_t)nj * 2u * h->P.sps));
    if (trace_cap) HIP_TRY(h, h->d_trace.reserve((size_t)nj * trace_cap));
    // Jobs, results and attempt records live in page-locked host memory that the kernel reads and writes directly:
    // 32 KB of jobs are read over PCIe once per workgroup, the records trickle out as they are produced, and nothing is
    // left to copy when the kernel ends (staged copies: one H2D before and two D2H after the launch, ~40 us of a 0.6 ms
    // pass).  LORA_HIP_STAGED=1 keeps the staged path (diagnostics).
    static const bool direct = getenv("LORA_HIP_STAGED") == nullptr;
    // staged only: attempt records per job fetched together with the results (what the busiest job of the previous launch needed)
    const uint32_t eager = direct ? recs_per_job : std::min(std::max(h->eager_recs, 2u), recs_per_job);
    HIP_TRY(h, h->p_jobs.reserve(nj));
    HIP_TRY(h, h->p_res.reserve(nj));
    HIP_TRY(h, h->p_recs.reserve((size_t)nj * eager));
    std::memcpy(h->p_jobs.p, jobs.data(), nj * sizeof(Job));
    if (!direct) HIP_TRY(h, hipMemcpyAsync(h->d_jobs.p, h->p_jobs.p, nj * sizeof(Job), hipMemcpyHostToDevice, st));
    LaunchCfg c{};
    c.iq = d_iq; c.recs_per_job = recs_per_job;
    c.jobs = direct ? h->p_jobs.p : h->d_jobs.p; c.results = direct ? h->p_res.p : h->d_results.p; c.recs = direct ? h->p_recs.p : h->d_recs.p;
    c.scratch = need_scratch ? h->d_scratch.p : nullptr;
    c.trace = trace_cap ? h->d_trace.p : nullptr; c.trace_cap = trace_cap; c.n_jobs = nj;
    static const bool no_balance = getenv("LORA_HIP_NO_BALANCE") != nullptr;
    if (!h->d_balance.p && !no_balance) {
        HIP_TRY(h, h->d_balance.reserve(kBalanceWords));
        HIP_TRY(h, hipMemsetAsync(h->d_balance.p, 0, kBalanceWords * sizeof(uint32_t), st));
    }
    c.balance = no_balance ? nullptr : h->d_balance.p;
    c.skip_payload = h->launch_skip ? 1u : 0u;
    HIP_TRY(h, hipEventRecord(h->ev0, st));
    if (nj >= h->last_kernel_jobs) { h->last_kernel = walker_kernel_name_for(h->P, nj, h->launch_skip); h->last_kernel_jobs = nj; } // (the pass's main launch: probe and fix-up launches are smaller)
    if (launch_walker(h->P, c, st) != 0) return fail(h, LORA_HIP_ERR_HIP, "walker launch failed: %s", hipGetErrorString(hipGetLastError()));
    HIP_TRY(h, hipEventRecord(h->ev1, st));
    if (!direct) {
        HIP_TRY(h, hipMemcpyAsync(h->p_res.p, h->d_results.p, nj * sizeof(JobResult), hipMemcpyDeviceToHost, st));
        HIP_TRY(h, hipMemcpy2DAsync(h->p_recs.p, eager * sizeof(AttemptRec), h->d_recs.p, recs_per_job * sizeof(AttemptRec), eager * sizeof(AttemptRec), nj,
                                    hipMemcpyDeviceToHost, st));
    }
    HIP_TRY(h, hipEventRecord(h->ev_done, st)); // (waited for instead of the stream: the caller may have queued another pass behind this one)
    h->pending.eager = eager; h->pending.direct = direct; h->pending.open = true;
    h->pending.hp1 = std::chrono::steady_clock::now();
    return LORA_HIP_OK;
}

lora_hip_status run_jobs_end(lora_hip_decoder *h, RunOut &out)
{
    if (!h->pending.open) return LORA_HIP_OK; // nothing was launched (no jobs)
    h->pending.open = false;
    const uint32_t nj = h->pending.nj, recs_per_job = h->pending.recs_per_job, eager = h->pending.eager, trace_cap = h->pending.trace_cap;
    const hipStream_t st = h->pending.st;
    const auto hp0 = h->pending.hp0, hp1 = h->pending.hp1;
    const Job *jobs = h->p_jobs.p;
    HIP_TRY(h, hipEventSynchronize(h->ev_done));
    const auto hp2 = std::chrono::steady_clock::now();
    std::memcpy(out.res.data(), h->p_res.p, nj * sizeof(JobResult));
    static const bool dbg_stats = getenv("LORA_HIP_DEBUG") != nullptr;
    if (dbg_stats && h->P.use_fast) {
        double cyc[6] = {0}, rnd[6] = {0};
        for (uint32_t j = 0; j < nj; j++) for (int i = 0; i < 6; i++) { cyc[i] += 64.0 * out.res[j].cyc[i]; rnd[i] += out.res[j].rounds[i]; }
        fprintf(stderr, "[lora_hip] per-job avg kcycles (rounds): DETECT %.0f (%.1f) SYNC %.0f (%.1f) SFD %.0f (%.1f) PAUSE %.0f (%.1f) HDR %.0f (%.1f) PAYLOAD %.0f (%.1f)\n",
                cyc[0] / nj / 1e3, rnd[0] / nj, cyc[1] / nj / 1e3, rnd[1] / nj, cyc[2] / nj / 1e3, rnd[2] / nj, cyc[3] / nj / 1e3, rnd[3] / nj, cyc[4] / nj / 1e3, rnd[4] / nj, cyc[5] / nj / 1e3, rnd[5] / nj);
        {
            double na = 0, nt = 0, ntv = 0; uint32_t sr[4] = {0, 0, 0, 0}, tsr[4] = {0, 0, 0, 0}, tpad = 0;
            for (uint32_t j = 0; j < nj; j++) {
                const JobResult &r = out.res[j];
                na += r.n_attempts; sr[r.stop_reason & 3u]++;
                if (r.tail_valid) { ntv++; nt += r.tail_n_attempts; tsr[r.tail_stop_reason & 3u]++; tpad += r.tail_pad; }
            }
            fprintf(stderr, "[lora_hip] attempts/job %.2f, stop reasons %u/%u/%u/%u; tails %.0f (attempts/tail %.2f, stop reasons %u/%u/%u/%u, pending %u)\n", na / nj, sr[0], sr[1],
                    sr[2], sr[3], ntv, ntv ? nt / ntv : 0.0, tsr[0], tsr[1], tsr[2], tsr[3], tpad);
        }
        double ctl[4] = {0};
        for (uint32_t j = 0; j < nj; j++) for (int i = 0; i < 4; i++) ctl[i] += 64.0 * out.res[j].ctl[i];
        fprintf(stderr, "[lora_hip] control per job, kcycles: copy-in %.0f loop %.0f plan %.0f copy-out %.0f (walker3: demodulation %.0f, replay %.0f of the decode rounds; with LORA_W3_REPLAY_STATS=2 copy-in and symbol loop in the 3rd / 4th figure)\n", ctl[0] / nj / 1e3, ctl[1] / nj / 1e3, ctl[2] / nj / 1e3, ctl[3] / nj / 1e3, ctl[0] / nj / 1e3, ctl[1] / nj / 1e3);
        std::vector<double> tot(nj);
        for (uint32_t j = 0; j < nj; j++) { double t = 0; for (int i = 0; i < 6; i++) t += 64.0 * out.res[j].cyc[i]; tot[j] = t; }
        std::sort(tot.begin(), tot.end());
        double mean = 0; for (double t : tot) mean += t; mean /= nj;
        fprintf(stderr, "[lora_hip] job kcycles over %u jobs: min %.0f p50 %.0f p90 %.0f max %.0f mean %.0f\n", nj, tot[0] / 1e3, tot[nj / 2] / 1e3,
                tot[(size_t)(nj * 0.9)] / 1e3, tot[nj - 1] / 1e3, mean / 1e3);
    }
    if (const char *tl = getenv("LORA_HIP_JOB_TIMELINE")) { // diagnostics: one line per job, appended to the named file
        if (FILE *f = fopen(tl, "a")) {
            fprintf(f, "# launch of %u jobs: job hw_id xcc_id t_start t_end(100MHz) cyc[6] x64 start scan_limit n_att\n", nj);
            for (uint32_t j = 0; j < nj; j++) {
                const JobResult &r = out.res[j];
                fprintf(f, "%u %u %u %u %u %u %u %u %u %u %u %lld %lld %u %u %u %u %u\n", j, r.dbg[0], r.dbg[1], r.dbg[2], r.dbg[3], r.cyc[0], r.cyc[1], r.cyc[2], r.cyc[3], r.cyc[4], r.cyc[5],
                        (long long)jobs[j].start, (long long)jobs[j].scan_limit, r.n_attempts, r.dbg[5] - r.dbg[4], r.rounds[5], r.rounds[0], r.rounds[2]);
            }
            fclose(f);
        }
    }
    // copy back only the attempt records that were written
    uint32_t max_att = 0;
    for (uint32_t j = 0; j < nj; j++) max_att = std::max(max_att, out.res[j].n_attempts + (out.res[j].tail_valid ? out.res[j].tail_n_attempts : 0u));
    if (max_att > recs_per_job) max_att = recs_per_job;
    // host copy of the records: only as many per job as the busiest job wrote (a 58-deep array per job would be 10 MB to clear)
    const uint32_t stride = std::max(max_att, 1u);
    out.rpj = stride;
    out.recs.resize_uninit((size_t)nj * stride); // every element is written below
    // the landed records: only the ones a job wrote, and of each only the header and the frame bytes it holds
    // (the landing buffer has just been written by DMA: every byte read from it comes from DRAM)
    for (uint32_t j = 0; j < nj; j++) {
        const uint32_t used = std::min(std::min(eager, max_att), out.res[j].n_attempts + (out.res[j].tail_valid ? out.res[j].tail_n_attempts : 0u));
        for (uint32_t a = 0; a < used; a++) {
            const AttemptRec *src = h->p_recs.p + (size_t)j * eager + a;
            const size_t n = offsetof(AttemptRec, frame) + std::min<size_t>(src->frame_len, sizeof src->frame);
            std::memcpy(&out.recs[(size_t)j * stride + a], src, n);
        }
    }
    if (dbg_stats && h->P.use_fast) { // acquisitions (SYNC steps) per published frame; tail probes that stopped behind their first FIND_SFD step
        double sync_rounds = 0; uint32_t frames = 0, tails = 0, early = 0;
        for (uint32_t j = 0; j < nj; j++) {
            const JobResult &r = out.res[j];
            sync_rounds += r.rounds[1];
            const uint32_t used = std::min(std::min(eager, max_att), r.n_attempts + (r.tail_valid ? r.tail_n_attempts : 0u));
            for (uint32_t a = 0; a < used; a++) {
                const AttemptRec &t = out.recs[(size_t)j * stride + a];
                frames += (a < r.n_attempts && t.status == kAttemptFrame) ? 1u : 0u;
                if (r.tail_valid && a + 1u == r.n_attempts + r.tail_n_attempts && r.tail_pad) { tails++; early += t.status == kAttemptAtSfd ? 1u : 0u; }
            }
        }
        fprintf(stderr, "[lora_hip] acquisitions: %.0f SYNC rounds for %u frames of this launch (%.2f per frame); %u tail probes reached a packet, %u of them stopped behind their first FIND_SFD step\n",
                sync_rounds, frames, frames ? sync_rounds / frames : 0.0, tails, early);
    }
    h->eager_recs = std::min(std::max(max_att, 2u), 8u);
    bool more = false;
    if (max_att > eager) { // rare: some job made more attempts than were fetched with the results
        HIP_TRY(h, hipMemcpy2DAsync(out.recs.data() + eager, stride * sizeof(AttemptRec), h->d_recs.p + eager,
                                    recs_per_job * sizeof(AttemptRec), (max_att - eager) * sizeof(AttemptRec), nj, hipMemcpyDeviceToHost, st));
        more = true;
    }
    if (trace_cap) {
        h->h_trace.resize((size_t)nj * trace_cap);
        HIP_TRY(h, hipMemcpyAsync(h->h_trace.data(), h->d_trace.p, (size_t)nj * trace_cap * sizeof(StepRec), hipMemcpyDeviceToHost, st));
        more = true;
    }
    if (more) HIP_TRY(h, hipStreamSynchronize(st));
    float ms = 0.0f;
    HIP_TRY(h, hipEventElapsedTime(&ms, h->ev0, h->ev1));
    if (dbg_stats) {
        const auto hp3 = std::chrono::steady_clock::now();
        auto us = [](auto a, auto b) { return std::chrono::duration<double, std::micro>(b - a).count(); };
        fprintf(stderr, "[lora_hip] run_jobs host: enqueue %.0f us, wait %.0f us (kernel %.0f us), unpack %.0f us; results %zu B + records %zu B\n", us(hp0, hp1), us(hp1, hp2),
                ms * 1e3, us(hp2, hp3), nj * sizeof(JobResult), (size_t)nj * eager * sizeof(AttemptRec));
    }
    h->timing.walker_ms += ms;
    h->timing.total_device_ms += ms; // (the walker launches only: the envelope pre-pass runs on its own stream beside the previous pass and is not part of this figure)
    h->timing.walker_launches++;
    return LORA_HIP_OK;
}

lora_hip_status run_jobs(lora_hip_decoder *h, const float2 *d_iq, const std::vector<Job> &jobs, uint32_t recs_per_job, uint32_t trace_cap,
                         hipStream_t st, RunOut &out)
{
    const lora_hip_status s = run_jobs_begin(h, d_iq, jobs, recs_per_job, trace_cap, st, out);
    return s == LORA_HIP_OK ? run_jobs_end(h, out) : s;
}

void append_trace(lora_hip_decoder *h, const RunOut &out, uint32_t job_index, uint32_t trace_cap, int64_t abs_base)
{
    const JobResult &jr = out.res[job_index];
    for (uint32_t i = 0; i < jr.n_steps; i++) {
        const StepRec &s = h->h_trace[(size_t)job_index * trace_cap + i];
        lora_hip_step_t o;
        o.state = s.state; o.consumed = s.consumed; o.pos = abs_base + s.pos; o.bin = s.bin; o.fine = s.fine;
        o.value = s.value; o.stream = s.stream; o.cycles = s.cycles; o.reserved = 0;
        h->trace.push_back(o);
    }
}

// Gap starts of every stream (item positions, ascending) from the energy envelope; see envelope_kernel.
uint64_t streams_signature(const std::vector<StreamDesc> &streams)
{ // FNV-1a over (offset, length) of every stream: tells whether a pre-pass issued ahead belongs to the pass now begun
    uint64_t x = 1469598103934665603ull;
    for (const StreamDesc &sd : streams)
        for (uint64_t v : {sd.off, sd.len}) { x ^= v; x *= 1099511628211ull; }
    return x ^ streams.size();
}

// ... in two steps: the kernels are put on the handle's pre-pass stream (quiet_edges_enqueue), the gap list is read once they
// are done (quiet_edges_collect).  A caller that pipelines passes issues the first step a pass ahead
// (lora_hip_decode_device_prepass): the kernels then run in the tail of the walker launch before the previous one.
lora_hip_status quiet_edges_enqueue(lora_hip_decoder *h, const float2 *d_iq, const std::vector<StreamDesc> &streams, hipStream_t st)
{
    const uint32_t sps = h->P.sps, ns = (uint32_t)streams.size();
    h->pre_issued = false;
    h->pre_hq0 = std::chrono::steady_clock::now();
    HIP_TRY(h, h->p_env_streams.reserve(ns));
    uint64_t nb = 0;
    for (uint32_t i = 0; i < ns; i++) {
        const uint64_t n = streams[i].len / sps;
        nb = (nb + 63u) & ~63ull;                                        // every stream starts a bitmap word
        if (n == 0 || nb + n > 0xffffffffull) return LORA_HIP_ERR_ARG; // caller falls back to the fixed grid
        h->p_env_streams.p[i] = EnvStream{streams[i].off, (uint32_t)nb, (uint32_t)n};
        nb += n;
    }
    const size_t words = (size_t)((nb + 63u) / 64u);                 // one bit per block
    HIP_TRY(h, h->d_env_E.reserve(nb));
    HIP_TRY(h, h->d_env_buf.reserve(words));
    HIP_TRY(h, h->p_env_buf.reserve(words));
    // The pre-pass runs on the handle's own stream, ordered after what the caller's stream holds so far (whoever
    // produced the IQ) unless the caller has declared the IQ ready (LORA_HIP_FLAG... see lora_hip_decode_device_begin):
    // then it runs beside whatever that stream is busy with - the previous pass's walker, when passes are pipelined.
    const hipStream_t ps = h->pre_stream;
    if (!h->iq_ready) {
        HIP_TRY(h, hipEventRecord(h->ev_dep, st));
        HIP_TRY(h, hipStreamWaitEvent(ps, h->ev_dep, 0));
    }
    static const bool dbg = getenv("LORA_HIP_DEBUG") != nullptr;
    if (dbg) HIP_TRY(h, hipEventRecord(h->ev_pre0, ps));
    static const bool direct = getenv("LORA_HIP_STAGED") == nullptr; // the bitmap is written straight into page-locked host memory
    if (launch_envelope(d_iq, h->p_env_streams.p, ns, sps, h->d_env_E.p, direct ? h->p_env_buf.p : h->d_env_buf.p, ps) != 0)
        return fail(h, LORA_HIP_ERR_HIP, "envelope launch failed: %s", hipGetErrorString(hipGetLastError()));
    if (dbg) HIP_TRY(h, hipEventRecord(h->ev_pre1, ps));
    if (!direct) HIP_TRY(h, hipMemcpyAsync(h->p_env_buf.p, h->d_env_buf.p, words * 8u, hipMemcpyDeviceToHost, ps));
    h->pre_issued = true; h->pre_iq = d_iq; h->pre_sig = streams_signature(streams); h->pre_ns = ns;
    return LORA_HIP_OK;
}

lora_hip_status quiet_edges_collect(lora_hip_decoder *h, std::vector<std::vector<int64_t>> &edges)
{
    if (!h->pre_issued) return LORA_HIP_ERR_ARG;
    h->pre_issued = false;
    const uint32_t sps = h->P.sps, ns = h->pre_ns;
    static const bool dbg = getenv("LORA_HIP_DEBUG") != nullptr;
    const auto hq0 = h->pre_hq0, hq1 = std::chrono::steady_clock::now();
    HIP_TRY(h, hipStreamSynchronize(h->pre_stream));
    const auto hq2 = std::chrono::steady_clock::now();
    if (dbg) { float ms = 0; if (hipEventElapsedTime(&ms, h->ev_pre0, h->ev_pre1) == hipSuccess) h->envelope_ms = ms; }
    // the set bits, stream by stream, in block order
    edges.resize(ns);
    uint32_t n_found = 0;
    const unsigned long long *bm = h->p_env_buf.p;
    for (uint32_t i = 0; i < ns; i++) {
        std::vector<int64_t> &e = edges[i];
        e.clear();
        const uint64_t b0 = h->p_env_streams.p[i].first_block, b1 = b0 + h->p_env_streams.p[i].n_blocks;
        for (uint64_t w = b0 / 64u; w * 64u < b1; w++) {
            unsigned long long bits = bm[w];
            while (bits) {
                const uint64_t b = w * 64u + (uint64_t)__builtin_ctzll(bits);
                bits &= bits - 1ull;
                if (b >= b0 && b < b1) e.push_back((int64_t)(b - b0) * (int64_t)sps);
            }
        }
        n_found += (uint32_t)e.size();
    }
    if (dbg) {
        auto us = [](auto a, auto b) { return std::chrono::duration<double, std::micro>(b - a).count(); };
        fprintf(stderr, "[lora_hip] envelope host: enqueue %.0f us, wait %.0f us, sort %.0f us (%u edges)\n", us(hq0, hq1), us(hq1, hq2), us(hq2, std::chrono::steady_clock::now()), n_found);
    }
    return LORA_HIP_OK;
}

// (a pre-pass issued ahead for exactly these streams is taken up - `allow_reuse`: only by the decode path, where a stale gap list
// costs speed, never correctness; lora_hip_gap_starts_device REPORTS the positions and always computes them afresh - anything
// else is issued now)
lora_hip_status quiet_edges(lora_hip_decoder *h, const float2 *d_iq, const std::vector<StreamDesc> &streams, hipStream_t st,
                            std::vector<std::vector<int64_t>> &edges, bool allow_reuse = true)
{
    if (!(allow_reuse && h->pre_issued && h->pre_iq == d_iq && h->pre_sig == streams_signature(streams))) {
        if (h->pre_issued) { (void)hipStreamSynchronize(h->pre_stream); h->pre_issued = false; } // stale: let it finish before its buffers are reused
        const lora_hip_status s = quiet_edges_enqueue(h, d_iq, streams, st);
        if (s != LORA_HIP_OK) return s;
    }
    return quiet_edges_collect(h, edges);
}

// The payload pass of a decoupled pass (lora_stitch.hpp payload_round): every payload symbol of every header-only packet demodulated at its
// zero-drift position by the symbol-level kernels - all symbols of all packets in one launch, whatever CU is free - then one small workgroup per
// packet walks its symbols in order (payload_chain_kernel).  A walk that meets a symbol whose predecessors moved the symbol clock by a shift nobody
// has read it at yet stops there; the host has the packet's remaining symbols read at that shift (one more launch for all such packets) and the walks
// run again with both sets of results to choose from - a +1 / -1 pair of moves, the usual case on a clean signal, costs one extra round however many
// pairs there are; a clock that drifts steadily needs a round per sample of drift and is handed back after kPayloadHyp - 1 of them (kPayloadUnresolved:
// the complete kernels decode that packet).  Offsets go up in one copy per round; descriptors are read from and results written to page-locked
// host memory directly, as the walkers' jobs and records are.
// (In two halves: payload_pass_begin puts the first round on the handle's payload stream and returns - the scheduler launches the pass's explicit
// probes on the caller's stream meanwhile, the device has room for both - payload_pass_end waits, reads the walks' results and runs the further
// rounds.  The IQ is known to be complete: the pass's main launch has read it and has been waited for.)
static uint32_t payload_fit_end(const PayloadReq &q, int64_t sps, int64_t shift, uint32_t from)
{ // symbols [from, result) of packet q, read `shift` samples behind their zero-drift positions, lie inside the data (:91)
    const int64_t room = (int64_t)q.stream_len - 2 * sps - (q.start + shift); // symbol j fits iff j sps <= room
    uint32_t n = 0;
    if (q.start + shift >= 0 && room >= 0) n = (uint32_t)std::min<int64_t>(room / sps + 1, (int64_t)q.n_walk);
    return std::max(n, from);
}

static lora_hip_status payload_launch_round(lora_hip_decoder *h, const float2 *d_iq, std::vector<PayloadReq> &reqs)
{
    lora_hip_decoder::PayState &ps = h->pay;
    const int64_t sps = (int64_t)h->P.sps;
    const size_t np = reqs.size();
    const hipStream_t st = h->pay_stream;
    PayloadDesc *descs = h->p_pay_desc.p, *launch = h->p_pay_desc.p + np;
    const bool no_alt = getenv("LORA_HIP_NO_SECOND_READS") != nullptr; // diagnostics / tests: every move of the symbol clock costs a round
    int64_t buf_end = 0; // the pass's streams end here at the latest: second reads stay inside the buffer (whether they stay inside their stream is the walk's check)
    for (const PayloadReq &q : reqs) buf_end = std::max<int64_t>(buf_end, (int64_t)(q.stream_off + q.stream_len));
    const bool has_alt = walker3_covers(h->P.sf); // (second reads are the workgroup-per-symbol kernels'; with the wave-per-symbol ones, SF7 / SF8, every move of the symbol clock costs a round)
    const DemodAlt alt{(no_alt || !has_alt) ? nullptr : h->d_alt_shift.p, h->d_alt_bins.p, h->d_alt_fine.p, buf_end - 2 * sps};
    // this round's reads: for every active packet the symbols from `at` on, `shift` behind their zero-drift positions
    size_t n_sym = 0;
    for (uint32_t i : ps.active) {
        PayloadReq &q = reqs[i];
        PayloadDesc &d = descs[i];
        const PayloadOut &o = h->p_pay_out.p[i];
        const uint32_t from = ps.round == 0 ? 0u : o.at;
        const int32_t shift = ps.round == 0 ? 0 : o.shift;
        const uint32_t to = payload_fit_end(q, sps, shift, from);
        const uint32_t hh = d.n_hyp++;
        d.hyp_shift[hh] = shift; d.hyp_from[hh] = from; d.hyp_to[hh] = to;
        d.hyp_base[hh] = (int32_t)(ps.used + n_sym) - (int32_t)from;
        for (uint32_t j = from; j < to; j++) h->p_pay_off.p[n_sym++] = (int64_t)q.stream_off + q.start + (int64_t)j * sps + shift;
    }
    if (ps.used + n_sym > ps.cap_sym) return fail(h, LORA_HIP_ERR_INTERNAL, "payload pass: more reads than symbols x rounds");
    for (size_t a = 0; a < ps.active.size(); a++) launch[a] = descs[ps.active[a]];
    HIP_TRY(h, hipEventRecord(h->ev_pay0, st));
    if (n_sym) {
        HIP_TRY(h, hipMemcpyAsync(h->d_offsets.p, h->p_pay_off.p, n_sym * sizeof(int64_t), hipMemcpyHostToDevice, st));
        DemodAlt ar{alt.shift ? alt.shift + ps.used : nullptr, alt.bins + ps.used, alt.fine + ps.used, alt.max_start};
        // (DemodAlt.shift needs no clearing: every entry is written by the symbol kernel - the wavefront / round of symbol s owns entry s + 1 - whether a second read was taken or not;
        //  until round 6 two hipMemsetAsync fills of a few hundred KB were 11.6 % of a config-4 pass's device time)
        if (launch_demod_symbols(h->P, d_iq, h->d_offsets.p, (uint32_t)n_sym, (int)h->P.demod_mode, h->d_bins.p + ps.used, h->d_fine.p + ps.used, nullptr, st, has_alt ? &ar : nullptr) != 0)
            return fail(h, LORA_HIP_ERR_HIP, "payload pass: symbol launch failed: %s", hipGetErrorString(hipGetLastError()));
    }
    // (launch entry a writes PayloadOut[a]; payload_pass_end spreads them to the packets' own slots)
    if (launch_payload_chain(h->P, h->d_bins.p, h->d_fine.p, alt, launch, h->p_pay_out.p, (uint32_t)ps.active.size(), st) != 0)
        return fail(h, LORA_HIP_ERR_HIP, "payload pass: chain launch failed: %s", hipGetErrorString(hipGetLastError()));
    HIP_TRY(h, hipEventRecord(h->ev_pay1, st));
    HIP_TRY(h, hipEventRecord(h->ev_pay_done, st));
    ps.n_sym = n_sym;
    return LORA_HIP_OK;
}

lora_hip_status payload_pass_begin(lora_hip_decoder *h, const float2 *d_iq, std::vector<PayloadReq> &reqs)
{
    lora_hip_decoder::PayState &ps = h->pay;
    ps = lora_hip_decoder::PayState{};
    const size_t np = reqs.size();
    if (np == 0) return LORA_HIP_OK;
    size_t total = 0;
    for (const PayloadReq &q : reqs) total += q.n_walk;
    ps.cap_sym = total * (size_t)kPayloadHyp;
    HIP_TRY(h, h->p_pay_off.reserve(total));
    HIP_TRY(h, h->p_pay_desc.reserve(2 * np)); // [0, np): one per packet, kept across the rounds; [np, 2 np): the round's launch list
    HIP_TRY(h, h->p_pay_out.reserve(np));
    HIP_TRY(h, h->d_offsets.reserve(total));
    HIP_TRY(h, h->d_bins.reserve(ps.cap_sym));
    HIP_TRY(h, h->d_fine.reserve(ps.cap_sym));
    HIP_TRY(h, h->d_alt_shift.reserve(ps.cap_sym));
    HIP_TRY(h, h->d_alt_bins.reserve(ps.cap_sym));
    HIP_TRY(h, h->d_alt_fine.reserve(ps.cap_sym));
    const int64_t sps = (int64_t)h->P.sps;
    ps.active.resize(np);
    for (size_t i = 0; i < np; i++) {
        PayloadReq &q = reqs[i];
        q.status = kPayloadUnresolved; q.end_shift = 0; q.frame_len = 0;
        PayloadDesc &d = h->p_pay_desc.p[i];
        d = PayloadDesc{};
        d.n_walk = q.n_walk; d.sk = q.sk;
        d.room = (int64_t)q.stream_len - 2 * sps - q.start;
        ps.active[i] = (uint32_t)i;
    }
    ps.open = true;
    return payload_launch_round(h, d_iq, reqs);
}

// an error between payload_pass_begin and the end of payload_pass_end: nothing of the pass may still be running on pay_stream (its kernels read and
// write the page-locked p_pay_desc / p_pay_out / p_pay_off) when the caller retries or reuses the handle
void payload_pass_abort(lora_hip_decoder *h)
{
    if (h->pay_stream) (void)hipStreamSynchronize(h->pay_stream);
    h->pay = lora_hip_decoder::PayState{};
}

lora_hip_status payload_pass_end(lora_hip_decoder *h, const float2 *d_iq, std::vector<PayloadReq> &reqs)
{
    lora_hip_decoder::PayState &ps = h->pay;
    if (!ps.open) return LORA_HIP_OK;
    ps.open = false;
    for (;;) {
        HIP_TRY(h, hipEventSynchronize(h->ev_pay_done));
        ps.used += ps.n_sym;
        h->last_payload_symbols += (uint32_t)ps.n_sym;
        h->last_payload_rounds++;
        float ms = 0.0f;
        HIP_TRY(h, hipEventElapsedTime(&ms, h->ev_pay0, h->ev_pay1));
        ps.ms += ms;
        // results sit at launch order: spread them to the packets' slots from the back (active[a] >= a)
        for (size_t a = ps.active.size(); a-- > 0;) if (ps.active[a] != a) h->p_pay_out.p[ps.active[a]] = h->p_pay_out.p[a];
        std::vector<uint32_t> next;
        for (uint32_t i : ps.active) {
            PayloadReq &q = reqs[i];
            const PayloadOut &o = h->p_pay_out.p[i];
            if (o.result == kWalkComplete && o.frame_len >= 3u && o.frame_len <= (uint32_t)sizeof q.frame) {
                q.status = kPayloadDecoded; q.end_shift = o.shift; q.frame_len = o.frame_len;
                std::memcpy(q.frame, o.frame, o.frame_len);
            } else if (o.result == kWalkOutOfData) {
                q.status = kPayloadOutOfData;
            } else if (o.result == kWalkNeedShift && h->p_pay_desc.p[i].n_hyp < (uint32_t)kPayloadHyp) {
                next.push_back(i);
            } // else: stays kPayloadUnresolved
        }
        ps.active.swap(next);
        ps.round++;
        if (ps.active.empty() || ps.round >= kPayloadHyp) break;
        const lora_hip_status s = payload_launch_round(h, d_iq, reqs);
        if (s != LORA_HIP_OK) return s;
    }
    h->last_payload_ms += ps.ms;
    h->timing.walker_ms += ps.ms;       // (the pass's device time: the header-only walkers and the payload pass's kernels)
    h->timing.total_device_ms += ps.ms;
    h->timing.walker_launches += 2;
    return LORA_HIP_OK;
}

// The device environment of the scheduler (lora_stitch.hpp): jobs are run by the walker kernels.
struct DeviceEnv {
    lora_hip_decoder *h;
    const float2 *d_iq;
    hipStream_t st;
    uint32_t sps() const { return h->P.sps; }
    uint32_t ctor_cr() const { return h->P.ctor_cr; }
    uint32_t segment_symbols() const { return h->cfg.segment_symbols; }
    uint32_t resident_slots() const
    {
        if (h->resident_slots == 0) h->resident_slots = std::max<uint32_t>(walker_resident_slots(h->P), 64u);
        return h->resident_slots;
    }
    uint32_t resident_slots_alt() const
    {
        if (h->resident_slots_alt == ~0u) h->resident_slots_alt = walker_resident_slots_full(h->P);
        return h->resident_slots_alt;
    }
    bool tracing() const { return (h->cfg.flags & LORA_HIP_FLAG_TRACE) != 0; }
    bool implicit() const { return h->P.implicit != 0; }
    // walker3 (SF9-12 at decimation 8) records the FIND_SFD entry states of every attempt; its tail probes stop behind their first
    // FIND_SFD step (Job.tail_stop_sfd) and are matched against those (LORA_HIP_NO_EARLY_PROBE=1: probes run to the header, as before)
    bool early_probe() const
    {
        static const bool off = getenv("LORA_HIP_NO_EARLY_PROBE") != nullptr;
        return !off && h->P.use_fast && h->P.decim == 8u && walker3_covers(h->P.sf);
    }
    // Decoupled pass: worth it while the jobs fit the device at once (a packet's symbols are a serial chain on one CU: the pass lasts as long as its
    // longest job, however few there are - header-only jobs are short whatever the packet's length, and their payloads fill the CUs the jobs leave
    // idle) and the traffic does not keep moving the symbol clock for good (such packets cost a probe each, or are run again whole).
    bool decoupled(size_t n_jobs)
    {
        h->last_payload_packets = 0; h->last_payload_rerun = 0; h->last_payload_symbols = 0; h->last_payload_ms = 0.0f; h->last_payload_moved = 0; h->last_payload_rounds = 0;
        if (h->decoupled_policy == 0 || !walker_has_skip_variant(h->P) || tracing()) return false;
        if (h->decoupled_policy == 1) return true;
        if (h->dec_backoff) return false; // (counted down once per pass: count_jobs)
        const uint32_t full = resident_slots_alt() ? resident_slots_alt() : resident_slots();
        constexpr uint32_t fill = 100u; // percent of the workgroup slots (config 4, 8 s per pass: 54.5 Gsamples/s ordinary, 76.0 decoupled with ~200 jobs)
        return 100u * (uint32_t)n_jobs <= fill * full;
    }
    void set_skip_payload(bool on) { h->launch_skip = on; }
    void abort_payload() { ::payload_pass_abort(h); }
    int run_payload_begin(std::vector<PayloadReq> &reqs) { return ::payload_pass_begin(h, d_iq, reqs) == LORA_HIP_OK ? 0 : -1; }
    int run_payload_end(std::vector<PayloadReq> &reqs) { return ::payload_pass_end(h, d_iq, reqs) == LORA_HIP_OK ? 0 : -1; }
    void count_payload(uint32_t packets, uint32_t moved, uint32_t rerun)
    {
        h->last_payload_packets += packets; h->last_payload_rerun += rerun; h->last_payload_moved += moved;
        if (rerun) h->timing.slow_path_relaunches++;
        if (h->decoupled_policy < 0 && packets >= 4u && 4u * (rerun + moved) > packets) h->dec_backoff = 32; // more than a quarter left the zero-drift grid for good: the complete kernels for a while
        static const bool dbg = getenv("LORA_HIP_DEBUG") != nullptr;
        if (dbg) fprintf(stderr, "[lora_hip] payload pass: %u packets, %u symbol reads in %u round(s), %.3f ms; %u packet(s) ended off the zero-drift grid (their jobs split there), %u handed to the complete kernels\n",
                         packets, h->last_payload_symbols, h->last_payload_rounds, h->last_payload_ms, moved, rerun);
    }
    RunOut &run_out(int which) { return h->run_out[which & 1]; }
    bool quiet_edges(const std::vector<StreamDesc> &streams, std::vector<std::vector<int64_t>> &edges)
    {
        return ::quiet_edges(h, d_iq, streams, st, edges) == LORA_HIP_OK;
    }
    int run_jobs(const std::vector<Job> &jobs, uint32_t rpj, uint32_t trace_cap, RunOut &out)
    {
        return ::run_jobs(h, d_iq, jobs, rpj, trace_cap, st, out) == LORA_HIP_OK ? 0 : -1;
    }
    int run_jobs_begin(const std::vector<Job> &jobs, uint32_t rpj, uint32_t trace_cap, RunOut &out)
    {
        return ::run_jobs_begin(h, d_iq, jobs, rpj, trace_cap, st, out) == LORA_HIP_OK ? 0 : -1;
    }
    int run_jobs_end(RunOut &out) { return ::run_jobs_end(h, out) == LORA_HIP_OK ? 0 : -1; }
    void publish(const AttemptRec &r, StreamDesc &sd) { ::publish(h, r, sd); }
    void append_trace(const RunOut &out, uint32_t job, uint32_t cap, int64_t base) { ::append_trace(h, out, job, cap, base); }
    void count_jobs(uint32_t n) { h->timing.jobs += n; h->last_kernel_jobs = 0; if (h->dec_backoff) h->dec_backoff--; } // (called once per pass, ahead of its main launch)
    void count_probes(uint32_t n) { h->timing.probes += n; }
    void count_slow_path() { h->timing.slow_path_relaunches++; }
    void count_repair()
    { // (a cut repaired in the probe launch - lora_stitch.hpp; reported under LORA_HIP_DEBUG only: lora_hip_timing_t is ABI)
        static const bool dbg = getenv("LORA_HIP_DEBUG") != nullptr;
        if (dbg) fprintf(stderr, "[lora_hip] cut repaired from the true header (no serial walk)\n");
    }
    void note_plan(bool burst_aware, size_t n_segs)
    {
        h->last_plan_burst = burst_aware ? 1u : 0u; h->last_plan_segments = (uint32_t)n_segs;
        static const bool dbg = getenv("LORA_HIP_DEBUG") != nullptr;
        if (dbg) fprintf(stderr, "[lora_hip] segment plan: %s, %zu segments (envelope kernels %.3f ms)\n", burst_aware ? "burst-aware" : "fixed grid", n_segs, burst_aware ? h->envelope_ms : 0.0f);
    }
    double walker_ms() const { return h->timing.walker_ms; }
};

lora_hip_status decode_streams(lora_hip_decoder *h, const float2 *d_iq, std::vector<StreamDesc> &streams, hipStream_t st)
{
    DeviceEnv env{h, d_iq, st};
    const int rc = lora_hip::decode_streams(env, streams);
    if (rc == 0) return LORA_HIP_OK;
    return h->err.empty() ? fail(h, LORA_HIP_ERR_INTERNAL, "scheduler failed") : LORA_HIP_ERR_HIP;
}

} // namespace

// ============================================================== C ABI ========

extern "C" {

uint32_t lora_hip_abi_version(void) { return LORA_HIP_ABI_VERSION; }

const char *lora_hip_strerror(lora_hip_status s)
{
    switch (s) {
    case LORA_HIP_OK: return "ok";
    case LORA_HIP_ERR_BAD_SF: return "spreading factor should be between 6 and 12 (inclusive)";
    case LORA_HIP_ERR_BAD_CONFIG: return "unsupported configuration";
    case LORA_HIP_ERR_NO_DEVICE: return "no usable HIP device (there is no CPU fallback)";
    case LORA_HIP_ERR_HIP: return "HIP runtime error";
    case LORA_HIP_ERR_NOMEM: return "out of memory";
    case LORA_HIP_ERR_ARG: return "bad argument";
    case LORA_HIP_ERR_OVERFLOW: return "buffer too small";
    default: return "internal error";
    }
}

const char *lora_hip_last_error(const lora_hip_decoder_t *h) { return h ? h->err.c_str() : g_create_err.c_str(); }

static void stream_pipe_release(lora_hip_decoder *h);

lora_hip_status lora_hip_create(const lora_hip_config_t *cfg, lora_hip_decoder_t **out)
{
    if (!cfg || !out) return LORA_HIP_ERR_ARG;
    *out = nullptr;
    if (cfg->struct_size != sizeof(lora_hip_config_t)) return LORA_HIP_ERR_ARG;
    if (cfg->sf < 6 || cfg->sf > 12) return LORA_HIP_ERR_BAD_SF; // :57-61 (message says 6..12)
    if (cfg->cr > 4 || cfg->demod < 0 || cfg->demod > 2 || cfg->bandwidth == 0 || !(cfg->samp_rate >= 1.0f)) return LORA_HIP_ERR_BAD_CONFIG;
    int ndev = 0;
    const hipError_t ec = hipGetDeviceCount(&ndev);
    if (ec != hipSuccess || ndev <= 0 || cfg->device < 0 || cfg->device >= ndev)
        return fail(nullptr, LORA_HIP_ERR_NO_DEVICE, "hipGetDeviceCount: %s, %d device(s), requested %d", hipGetErrorString(ec), ndev, cfg->device);
    lora_hip_decoder *h = new (std::nothrow) lora_hip_decoder();
    if (!h) return LORA_HIP_ERR_NOMEM;
    h->cfg = *cfg;
    h->device = cfg->device;
    lora_hip_status s = LORA_HIP_OK;
    const hipError_t es = hipSetDevice(h->device);
    if (es != hipSuccess) s = fail(nullptr, LORA_HIP_ERR_NO_DEVICE, "hipSetDevice(%d): %s", h->device, hipGetErrorString(es));
    if (s == LORA_HIP_OK) s = build_tables(h);
    if (s == LORA_HIP_OK && (hipEventCreate(&h->ev0) != hipSuccess || hipEventCreate(&h->ev1) != hipSuccess || hipEventCreate(&h->ev_done) != hipSuccess ||
                             hipEventCreate(&h->ev_pre0) != hipSuccess || hipEventCreate(&h->ev_pre1) != hipSuccess ||
                             hipEventCreateWithFlags(&h->ev_dep, hipEventDisableTiming) != hipSuccess ||
                             hipStreamCreateWithFlags(&h->pre_stream, hipStreamNonBlocking) != hipSuccess ||
                             hipStreamCreateWithFlags(&h->pay_stream, hipStreamNonBlocking) != hipSuccess || hipEventCreate(&h->ev_pay0) != hipSuccess ||
                             hipEventCreate(&h->ev_pay1) != hipSuccess || hipEventCreate(&h->ev_pay_done) != hipSuccess))
        s = LORA_HIP_ERR_HIP;
    if (s != LORA_HIP_OK) { lora_hip_destroy(h); return s; }
    h->stream_cr = h->P.ctor_cr;
    h->batch_items = cfg->batch_items ? cfg->batch_items : std::max<size_t>(1u << 20, 64ull * h->P.sps);
    *out = h;
    return LORA_HIP_OK;
}

void lora_hip_destroy(lora_hip_decoder_t *h)
{
    if (!h) return;
    (void)hipSetDevice(h->device);
    if (h->sp.comp_st) { (void)hipStreamSynchronize(h->sp.copy_st); (void)hipStreamSynchronize(h->sp.comp_st); } // a streaming pass may be in flight
    if (h->d_down) (void)hipFree(h->d_down);
    if (h->d_twN) (void)hipFree(h->d_twN);
    if (h->d_tws) (void)hipFree(h->d_tws);
    if (h->d_wave_tabs) (void)hipFree(h->d_wave_tabs);
    if (h->d_w3_tw) (void)hipFree(h->d_w3_tw);
    if (h->d_team_tabs) (void)hipFree(h->d_team_tabs);
    if (h->d_w3_ctab) (void)hipFree(h->d_w3_ctab);
    if (h->d_up_ifreq) (void)hipFree(h->d_up_ifreq);
    if (h->d_down_ifreq) (void)hipFree(h->d_down_ifreq);
    if (h->d_up_ifreq_v) (void)hipFree(h->d_up_ifreq_v);
    h->p_pay_off.release(); h->p_pay_desc.release(); h->p_pay_out.release(); h->d_fine.release(); h->d_alt_shift.release(); h->d_alt_bins.release(); h->d_alt_fine.release();
    h->d_jobs.release(); h->d_results.release(); h->d_recs.release(); h->d_scratch.release();
    h->d_trace.release(); h->d_staging.release(); h->d_offsets.release(); h->d_bins.release();
    stream_pipe_release(h);
    h->p_jobs.release(); h->p_res.release(); h->p_recs.release();
    h->d_balance.release(); h->d_env_E.release(); h->d_env_buf.release(); h->p_env_streams.release(); h->p_env_buf.release();
    for (hipEvent_t e : {h->ev0, h->ev1, h->ev_done, h->ev_pre0, h->ev_pre1, h->ev_dep, h->ev_pay0, h->ev_pay1, h->ev_pay_done})
        if (e) (void)hipEventDestroy(e);
    if (h->pre_stream) (void)hipStreamDestroy(h->pre_stream);
    if (h->pay_stream) (void)hipStreamDestroy(h->pay_stream);
    delete h;
}

lora_hip_status lora_hip_get_table(const lora_hip_decoder_t *h, int which, float *buf, size_t cap_floats, size_t *n_floats)
{
    if (!h) return LORA_HIP_ERR_ARG;
    const size_t sps = h->P.sps;
    const void *src = nullptr;
    size_t n = 0;
    bool on_device = true;
    switch (which) {
    case 0: src = h->d_down; n = 2 * sps; break;
    case 1: src = h->h_up.data(); n = 2 * sps; on_device = false; break;
    case 2: src = h->d_down_ifreq; n = sps; break;
    case 3: src = h->d_up_ifreq; n = sps; break;
    case 4: src = h->d_up_ifreq_v; n = h->n_up_ifreq_v; break;
    default: return LORA_HIP_ERR_ARG;
    }
    if (n_floats) *n_floats = n;
    if (!buf) return LORA_HIP_OK;
    if (cap_floats < n || !src) return LORA_HIP_ERR_ARG;
    if (!on_device) { memcpy(buf, src, n * sizeof(float)); return LORA_HIP_OK; }
    if (hipSetDevice(h->device) != hipSuccess || hipMemcpy(buf, src, n * sizeof(float), hipMemcpyDeviceToHost) != hipSuccess) return LORA_HIP_ERR_HIP;
    return LORA_HIP_OK;
}

lora_hip_status lora_hip_get_geometry(const lora_hip_decoder_t *h, uint32_t *sps, uint32_t *bins, uint32_t *decim)
{
    if (!h) return LORA_HIP_ERR_ARG;
    if (sps) *sps = h->P.sps;
    if (bins) *bins = h->P.nbins;
    if (decim) *decim = h->P.decim;
    return LORA_HIP_OK;
}

lora_hip_status lora_hip_set_sf(lora_hip_decoder_t *h, uint8_t sf)
{ // :905-909 -- warn only
    if (!h) return LORA_HIP_ERR_ARG;
    (void)sf;
    fprintf(stderr, "[LoRa Decoder] WARNING : Setting the spreading factor during execution is currently not supported.\n"
                    "Nothing set, kept SF of %u.\n", (unsigned)h->P.sf);
    return LORA_HIP_OK;
}

lora_hip_status lora_hip_set_samp_rate(lora_hip_decoder_t *h, float samp_rate)
{ // :911-915 -- warn only
    if (!h) return LORA_HIP_ERR_ARG;
    (void)samp_rate;
    fprintf(stderr, "[LoRa Decoder] WARNING : Setting the sample rate during execution is currently not supported.\n"
                    "Nothing set, kept SR of %u.\n", (unsigned)h->cfg.samp_rate);
    return LORA_HIP_OK;
}

lora_hip_status lora_hip_decode_device_begin(lora_hip_decoder_t *h, const void *d_iq, size_t total_items, const uint64_t *stream_off,
                                             const uint64_t *stream_len, uint32_t n_streams, void *hip_stream, uint32_t flags)
{
    if (!h || (!d_iq && total_items) || (n_streams && (!stream_off || !stream_len))) return LORA_HIP_ERR_ARG;
    if (h->pass_open) return fail(h, LORA_HIP_ERR_ARG, "lora_hip_decode_device_begin: the previous pass has not been ended");
    HIP_TRY(h, hipSetDevice(h->device));
    h->timing = lora_hip_timing_t{};
    std::vector<StreamDesc> &sds = h->pass_streams;
    sds.assign(n_streams, StreamDesc{});
    for (uint32_t i = 0; i < n_streams; i++) {
        if (stream_off[i] > total_items || stream_len[i] > total_items - stream_off[i]) return fail(h, LORA_HIP_ERR_ARG, "stream %u exceeds the buffer", i);
        sds[i].off = stream_off[i]; sds[i].len = stream_len[i]; sds[i].id = i;
        sds[i].cr_in = h->P.ctor_cr; sds[i].abs_base = 0;
        h->timing.items += stream_len[i];
    }
    h->pass_iq = (const float2 *)d_iq; h->pass_st = (hipStream_t)hip_stream;
    h->iq_ready = (flags & LORA_HIP_BEGIN_IQ_READY) != 0u;
    h->err.clear();
    DeviceEnv env{h, h->pass_iq, h->pass_st};
    const int rc = lora_hip::decode_begin(env, sds, h->pass);
    h->iq_ready = false;
    if (rc != 0) { h->pending.open = false; return h->err.empty() ? fail(h, LORA_HIP_ERR_INTERNAL, "scheduler failed") : LORA_HIP_ERR_HIP; }
    h->pass_open = true;
    return LORA_HIP_OK;
}

lora_hip_status lora_hip_decode_device_prepass(lora_hip_decoder_t *h, const void *d_iq, size_t total_items, const uint64_t *stream_off,
                                               const uint64_t *stream_len, uint32_t n_streams, void *hip_stream, uint32_t flags)
{
    if (!h || !d_iq || !n_streams || !stream_off || !stream_len) return LORA_HIP_ERR_ARG;
    if (h->pass_open) return fail(h, LORA_HIP_ERR_ARG, "lora_hip_decode_device_prepass: a pass is open on this handle");
    HIP_TRY(h, hipSetDevice(h->device));
    // only where decode_begin would plan around the bursts (auto segment length, explicit header, no tracing); harmless otherwise
    if (h->cfg.segment_symbols != 0 || h->P.implicit || (h->cfg.flags & LORA_HIP_FLAG_TRACE)) return LORA_HIP_OK;
    std::vector<StreamDesc> sds(n_streams);
    for (uint32_t i = 0; i < n_streams; i++) {
        if (stream_off[i] > total_items || stream_len[i] > total_items - stream_off[i]) return fail(h, LORA_HIP_ERR_ARG, "stream %u exceeds the buffer", i);
        sds[i].off = stream_off[i]; sds[i].len = stream_len[i]; sds[i].id = i;
    }
    if (h->pre_issued) { HIP_TRY(h, hipStreamSynchronize(h->pre_stream)); h->pre_issued = false; }
    h->iq_ready = (flags & LORA_HIP_BEGIN_IQ_READY) != 0u;
    h->err.clear();
    const lora_hip_status s = quiet_edges_enqueue(h, (const float2 *)d_iq, sds, (hipStream_t)hip_stream);
    h->iq_ready = false;
    return (s == LORA_HIP_OK || h->err.empty()) ? LORA_HIP_OK : s; // streams without an envelope simply get none
}

lora_hip_status lora_hip_decode_device_end(lora_hip_decoder_t *h)
{
    if (!h) return LORA_HIP_ERR_ARG;
    if (!h->pass_open) return fail(h, LORA_HIP_ERR_ARG, "lora_hip_decode_device_end without a pass begun");
    // (the streaming pipeline keeps its in-flight pass in the same PassCtx: collecting it here would leave lora_hip_work's tail,
    // d_phdr.cr and power queue behind - that pass belongs to lora_hip_work / lora_hip_flush)
    if (h->sp.inflight) return fail(h, LORA_HIP_ERR_ARG, "lora_hip_decode_device_end: the open pass is lora_hip_work's; call lora_hip_flush");
    h->pass_open = false;
    HIP_TRY(h, hipSetDevice(h->device));
    h->err.clear();
    DeviceEnv env{h, h->pass_iq, h->pass_st};
    const int rc = lora_hip::decode_end(env, h->pass_streams, h->pass);
    if (rc != 0) { h->pending.open = false; return h->err.empty() ? fail(h, LORA_HIP_ERR_INTERNAL, "scheduler failed") : LORA_HIP_ERR_HIP; }
    return LORA_HIP_OK;
}

lora_hip_status lora_hip_decode_device(lora_hip_decoder_t *h, const void *d_iq, size_t total_items,
                                       const uint64_t *stream_off, const uint64_t *stream_len, uint32_t n_streams,
                                       void *hip_stream)
{
    const lora_hip_status s = lora_hip_decode_device_begin(h, d_iq, total_items, stream_off, stream_len, n_streams, hip_stream, 0u);
    return s == LORA_HIP_OK ? lora_hip_decode_device_end(h) : s;
}

lora_hip_status lora_hip_gap_starts_device(lora_hip_decoder_t *h, const void *d_iq, size_t total_items, const uint64_t *stream_off,
                                           const uint64_t *stream_len, uint32_t n_streams, int64_t *pos, size_t cap, uint32_t *counts,
                                           void *hip_stream)
{
    if (!h || !d_iq || !n_streams || !stream_off || !stream_len || !counts || (!pos && cap)) return LORA_HIP_ERR_ARG;
    HIP_TRY(h, hipSetDevice(h->device));
    std::vector<StreamDesc> sds(n_streams);
    for (uint32_t i = 0; i < n_streams; i++) {
        if (stream_off[i] > total_items || stream_len[i] > total_items - stream_off[i]) return fail(h, LORA_HIP_ERR_ARG, "stream %u exceeds the buffer", i);
        sds[i].off = stream_off[i]; sds[i].len = stream_len[i]; sds[i].id = i;
    }
    std::vector<std::vector<int64_t>> edges;
    h->err.clear();
    const lora_hip_status s = quiet_edges(h, (const float2 *)d_iq, sds, (hipStream_t)hip_stream, edges, /*allow_reuse=*/false);
    if (s != LORA_HIP_OK) return h->err.empty() ? fail(h, LORA_HIP_ERR_BAD_CONFIG, "no envelope for these streams (shorter than a symbol, or fewer than 128 samples per symbol)") : s;
    size_t used = 0;
    for (uint32_t i = 0; i < n_streams; i++) {
        counts[i] = (uint32_t)edges[i].size();
        if (used + edges[i].size() > cap) return LORA_HIP_ERR_OVERFLOW;
        std::copy(edges[i].begin(), edges[i].end(), pos + used);
        used += edges[i].size();
    }
    return LORA_HIP_OK;
}

// ---- lora_hip_work: the reference block's work() contract (host buffers in, frames out), pipelined ------------------
// The stream is cut into chunks of batch_items.  Samples are uploaded AS THEY ARRIVE (asynchronously, on a copy stream of
// the handle) into the chunk area of one of two device buffers; when a chunk is full, the pass over the previous chunk is
// collected (state, frames, the undecoded tail), the tail is copied in front of the new chunk (device to device), and the
// pass over [tail | chunk] is launched - so the device decodes chunk k while chunk k + 1 is being uploaded and the host
// only ever waits for a kernel that had a whole chunk's arrival time to finish.  Frames surface one chunk later than in
// a synchronous pass; lora_hip_flush() drains everything.  Caller memory that is page-locked (or that hipHostRegister
// accepts: GNU Radio's buffers are long-lived) is DMA'd from directly; anything else goes through two pinned bounce
// buffers.  Output is what one pass over the whole stream would give (tests/test_gpu_parity.py::test_streaming_*).
static void stream_pipe_release(lora_hip_decoder *h)
{
    auto &sp = h->sp;
    for (auto &r : sp.pinned) (void)hipHostUnregister((void *)r.first);
    sp.pinned.clear(); sp.refused.clear();
    for (int i = 0; i < 2; i++) {
        sp.dbuf[i].release(); sp.stage[i].release();
        if (sp.stage_ev[i]) { (void)hipEventDestroy(sp.stage_ev[i]); sp.stage_ev[i] = nullptr; }
    }
    if (sp.up_ev) { (void)hipEventDestroy(sp.up_ev); sp.up_ev = nullptr; }
    if (sp.tail_ev) { (void)hipEventDestroy(sp.tail_ev); sp.tail_ev = nullptr; }
    if (sp.copy_st) { (void)hipStreamDestroy(sp.copy_st); sp.copy_st = nullptr; }
    if (sp.comp_st) { (void)hipStreamDestroy(sp.comp_st); sp.comp_st = nullptr; }
}

static lora_hip_status stream_pipe_init(lora_hip_decoder *h)
{
    auto &sp = h->sp;
    if (sp.copy_st) return LORA_HIP_OK;
    HIP_TRY(h, hipSetDevice(h->device));
    HIP_TRY(h, hipStreamCreateWithFlags(&sp.copy_st, hipStreamNonBlocking));
    HIP_TRY(h, hipStreamCreateWithFlags(&sp.comp_st, hipStreamNonBlocking));
    HIP_TRY(h, hipEventCreateWithFlags(&sp.up_ev, hipEventDisableTiming));
    HIP_TRY(h, hipEventCreateWithFlags(&sp.tail_ev, hipEventDisableTiming));
    for (int i = 0; i < 2; i++) HIP_TRY(h, hipEventCreateWithFlags(&sp.stage_ev[i], hipEventDisableTiming));
    sp.tailcap = std::max<size_t>(h->batch_items, 4u * (size_t)h->P.sps);
    for (int i = 0; i < 2; i++) HIP_TRY(h, sp.dbuf[i].reserve(sp.tailcap + h->batch_items));
    return LORA_HIP_OK;
}

// is [p, p + bytes) inside memory the device can DMA from?  Tries to page-lock unknown ranges once.
static bool stream_host_pinned(lora_hip_decoder *h, const void *p, size_t bytes)
{
    auto &sp = h->sp;
    const uintptr_t a = (uintptr_t)p, b = a + bytes;
    for (auto &r : sp.pinned) if (a >= r.first && b <= r.second) return true;
    for (auto &r : sp.refused) if (a >= r.first && b <= r.second) return false;
    { // hipHostMalloc'ed, or registered by the caller: both ends must be (a range registered only in part is not usable)
        hipPointerAttribute_t a0{}, a1{};
        const bool h0 = hipPointerGetAttributes(&a0, p) == hipSuccess && a0.type == hipMemoryTypeHost;
        const bool h1 = bytes && hipPointerGetAttributes(&a1, (const char *)p + bytes - 1) == hipSuccess && a1.type == hipMemoryTypeHost;
        (void)hipGetLastError();
        if (h0 && h1) return true;
        if (h0 || h1) return false; // straddles somebody's registration: registering the rest would overlap it
    }
    // Page-locking the caller's memory ourselves is opt-in (LORA_HIP_FLAG_PIN_HOST): a partly registered allocation makes
    // OTHER HIP copies out of it fail, so it is only safe where the caller hands over long-lived buffers it uses for nothing
    // else (a GNU Radio block's input buffer).  Without the flag, memory that is not page-locked goes through the bounce buffers.
    if (!(h->cfg.flags & LORA_HIP_FLAG_PIN_HOST) || bytes < (256u << 10)) return false; // (small calls: the bounce copy is cheaper)
    const uintptr_t pg = 4096u, ra = a & ~(pg - 1u), rb = (b + pg - 1u) & ~(pg - 1u);
    if (hipHostRegister((void *)ra, rb - ra, hipHostRegisterDefault) == hipSuccess) { sp.pinned.emplace_back(ra, rb); return true; }
    (void)hipGetLastError();
    sp.refused.emplace_back(a, b);
    return false;
}

// collects the pass in flight: decoder state, frames, and the undecoded tail moved in front of the chunk being filled
static lora_hip_status stream_collect(lora_hip_decoder *h)
{
    auto &sp = h->sp;
    if (!sp.inflight) return LORA_HIP_OK;
    sp.inflight = false;
    h->pass_open = false;
    h->err.clear();
    DeviceEnv env{h, h->pass_iq, h->pass_st};
    const int rc = lora_hip::decode_end(env, h->pass_streams, h->pass);
    if (rc != 0) { h->pending.open = false; return h->err.empty() ? fail(h, LORA_HIP_ERR_INTERNAL, "scheduler failed") : LORA_HIP_ERR_HIP; }
    const StreamDesc &sd = h->pass_streams[0];
    h->stream_cr = sd.cr_out;
    h->stream_pwr = sd.pwr;
    const size_t keep_from = (size_t)std::min<int64_t>(std::max<int64_t>(sd.final_pos, 0), (int64_t)sp.fl_len);
    const size_t tail = sp.fl_len - keep_from; // an attempt that ran out of data is re-run from its start with the next chunk behind it
    h->host_base += (int64_t)keep_from;
    if (tail > sp.tailcap) { // (a packet longer than the tail area: grow both buffers, keeping what the filling one holds)
        const size_t ncap = std::max(2u * sp.tailcap, tail + (size_t)h->P.sps);
        HIP_TRY(h, hipStreamSynchronize(sp.copy_st));
        for (int i = 0; i < 2; i++) {
            DevBuf<float2> nb;
            HIP_TRY(h, nb.reserve(ncap + h->batch_items));
            HIP_TRY(h, hipMemcpyAsync(nb.p + ncap, sp.dbuf[i].p + sp.tailcap, h->batch_items * sizeof(float2), hipMemcpyDeviceToDevice, sp.comp_st));
            if (i == (sp.cur ^ 1)) // the buffer the pass ran on: its stream region moves along (it is the source of the tail below)
                HIP_TRY(h, hipMemcpyAsync(nb.p + ncap - (sp.tailcap - sp.fl_off), sp.dbuf[i].p + sp.fl_off, (sp.tailcap - sp.fl_off) * sizeof(float2), hipMemcpyDeviceToDevice, sp.comp_st));
            HIP_TRY(h, hipStreamSynchronize(sp.comp_st));
            std::swap(sp.dbuf[i], nb);
            nb.release();
        }
        sp.fl_off += ncap - sp.tailcap;
        sp.tailcap = ncap;
    }
    if (tail) {
        HIP_TRY(h, hipMemcpyAsync(sp.dbuf[sp.cur].p + sp.tailcap - tail, sp.dbuf[sp.cur ^ 1].p + sp.fl_off + keep_from, tail * sizeof(float2), hipMemcpyDeviceToDevice, sp.comp_st));
        // the source sits in the chunk area that the NEXT uploads overwrite: they wait for this copy
        HIP_TRY(h, hipEventRecord(sp.tail_ev, sp.comp_st));
        HIP_TRY(h, hipStreamWaitEvent(sp.copy_st, sp.tail_ev, 0));
    }
    sp.tail_len = tail;
    return LORA_HIP_OK;
}

// the chunk being filled is complete (or the stream is being flushed): collect the previous pass, launch this one
static lora_hip_status stream_rotate(lora_hip_decoder *h)
{
    auto &sp = h->sp;
    HIP_TRY(h, hipEventRecord(sp.up_ev, sp.copy_st));
    lora_hip_status s = stream_collect(h);
    if (s != LORA_HIP_OK) return s;
    const size_t len = sp.tail_len + sp.fill;
    if (len >= 2u * (size_t)h->P.sps) {
        HIP_TRY(h, hipStreamWaitEvent(sp.comp_st, sp.up_ev, 0));
        std::vector<StreamDesc> &sds = h->pass_streams;
        sds.assign(1, StreamDesc{});
        sds[0].off = sp.tailcap - sp.tail_len; sds[0].len = len; sds[0].id = 0;
        sds[0].cr_in = h->stream_cr; sds[0].pwr = h->stream_pwr; sds[0].abs_base = h->host_base;
        h->timing = lora_hip_timing_t{};
        h->timing.items = len;
        h->pass_iq = sp.dbuf[sp.cur].p; h->pass_st = sp.comp_st;
        h->err.clear();
        DeviceEnv env{h, h->pass_iq, h->pass_st};
        const int rc = lora_hip::decode_begin(env, sds, h->pass);
        if (rc != 0) { h->pending.open = false; return h->err.empty() ? fail(h, LORA_HIP_ERR_INTERNAL, "scheduler failed") : LORA_HIP_ERR_HIP; }
        h->pass_open = true;
        sp.inflight = true; sp.fl_off = sp.tailcap - sp.tail_len; sp.fl_len = len;
        sp.cur ^= 1; sp.fill = 0; sp.tail_len = 0;
        sp.have_first = false;
        sp.passes++;
    }
    // (shorter than one work() call of the reference, :91: keep filling the same chunk)
    return LORA_HIP_OK;
}

lora_hip_status lora_hip_work(lora_hip_decoder_t *h, const float *iq, size_t n_items, size_t *consumed)
{
    if (!h || (!iq && n_items)) return LORA_HIP_ERR_ARG;
    if (h->pass_open && !h->sp.inflight) return fail(h, LORA_HIP_ERR_ARG, "lora_hip_work: a lora_hip_decode_device_begin pass is open on this handle");
    lora_hip_status s = stream_pipe_init(h);
    if (s != LORA_HIP_OK) return s;
    HIP_TRY(h, hipSetDevice(h->device));
    auto &sp = h->sp;
    const float2 *src = reinterpret_cast<const float2 *>(iq);
    size_t left = n_items;
    bool direct_pending = false;
    // a pass whose kernel has finished is collected now - its frames are published by this call, not a chunk later
    if (sp.inflight && h->pending.open && hipEventQuery(h->ev_done) == hipSuccess) {
        s = stream_collect(h);
        if (s != LORA_HIP_OK) return s;
    }
    (void)hipGetLastError(); // (hipErrorNotReady of the query)
    const bool direct = n_items != 0 && stream_host_pinned(h, iq, n_items * sizeof(float2)); // (the whole call's range, once)
    while (left) {
        if (sp.fill == h->batch_items) { // (a chunk that could not be launched yet because the stream was shorter than 2 sps)
            s = stream_rotate(h);
            if (s != LORA_HIP_OK) return s;
            if (sp.fill == h->batch_items) return fail(h, LORA_HIP_ERR_BAD_CONFIG, "batch_items is smaller than two symbols");
        }
        const size_t m = std::min(left, h->batch_items - sp.fill);
        float2 *dst = sp.dbuf[sp.cur].p + sp.tailcap + sp.fill;
        if (direct) {
            HIP_TRY(h, hipMemcpyAsync(dst, src, m * sizeof(float2), hipMemcpyHostToDevice, sp.copy_st));
            direct_pending = true;
            sp.bytes_direct += m * sizeof(float2);
        } else { // through a pinned bounce buffer, in pieces, two in flight
            size_t done = 0;
            const size_t piece = std::max<size_t>(h->batch_items / 4u, 16384u);
            while (done < m) {
                const size_t q = std::min(piece, m - done);
                const int k = sp.stage_i;
                if (sp.stage_busy[k]) { HIP_TRY(h, hipEventSynchronize(sp.stage_ev[k])); sp.stage_busy[k] = false; }
                HIP_TRY(h, sp.stage[k].reserve(piece));
                std::memcpy(sp.stage[k].p, src + done, q * sizeof(float2));
                HIP_TRY(h, hipMemcpyAsync(dst + done, sp.stage[k].p, q * sizeof(float2), hipMemcpyHostToDevice, sp.copy_st));
                HIP_TRY(h, hipEventRecord(sp.stage_ev[k], sp.copy_st));
                sp.stage_busy[k] = true;
                sp.stage_i ^= 1;
                done += q;
            }
            sp.bytes_staged += m * sizeof(float2);
        }
        if (!sp.have_first) { sp.have_first = true; sp.t_first = std::chrono::steady_clock::now(); }
        sp.fill += m; src += m; left -= m;
        if (sp.fill == h->batch_items) {
            s = stream_rotate(h);
            if (s != LORA_HIP_OK) return s;
        }
    }
    // the caller may reuse its buffer as soon as we return (the scheduler's contract): the DMA out of it must be done
    if (direct_pending) HIP_TRY(h, hipStreamSynchronize(sp.copy_st));
    // latency bound: the oldest sample not yet handed to a pass has waited long enough (and a pass could run at all, :91)
    if (sp.max_latency_ms > 0.0f && sp.have_first && sp.fill != 0 && sp.tail_len + sp.fill >= 2u * (size_t)h->P.sps &&
        std::chrono::duration<float, std::milli>(std::chrono::steady_clock::now() - sp.t_first).count() >= sp.max_latency_ms) {
        const uint64_t before = sp.passes;
        s = stream_rotate(h);
        if (s != LORA_HIP_OK) return s;
        sp.passes_by_latency += sp.passes - before;
    }
    if (consumed) *consumed = n_items;
    return LORA_HIP_OK;
}

lora_hip_status lora_hip_set_stream_latency(lora_hip_decoder_t *h, float max_latency_ms)
{
    if (!h || !(max_latency_ms >= 0.0f)) return LORA_HIP_ERR_ARG;
    h->sp.max_latency_ms = max_latency_ms;
    return LORA_HIP_OK;
}

lora_hip_status lora_hip_stream_info(const lora_hip_decoder_t *h, lora_hip_stream_info_t *out)
{
    if (!h || !out) return LORA_HIP_ERR_ARG;
    out->batch_items = h->batch_items;
    out->buffered_items = h->sp.tail_len + h->sp.fill;
    out->passes = h->sp.passes; out->passes_by_latency = h->sp.passes_by_latency;
    out->consumed_base = h->host_base;
    out->max_latency_ms = h->sp.max_latency_ms;
    out->pass_in_flight = h->sp.inflight ? 1u : 0u;
    return LORA_HIP_OK;
}

lora_hip_status lora_hip_flush(lora_hip_decoder_t *h)
{
    if (!h) return LORA_HIP_ERR_ARG;
    if (!h->sp.copy_st) return LORA_HIP_OK; // nothing was ever fed
    HIP_TRY(h, hipSetDevice(h->device));
    lora_hip_status s = stream_rotate(h); // launches what is buffered (collecting the pass before it)
    if (s != LORA_HIP_OK) return s;
    s = stream_collect(h);                // ... and collects that one as well; its tail waits in front of the next chunk
    if (s != LORA_HIP_OK) return s;
    HIP_TRY(h, hipStreamSynchronize(h->sp.comp_st));
    return LORA_HIP_OK;
}

// ---- lora_hip_mux: many channels, one pass (the gateway flowgraph; see include/lora_hip.h) ------------------------------
// Device layout: two buffers of n_channels regions [ tail area (tailcap items, right-aligned) | chunk (batch items) ]; the
// pass over buffer b decodes n_channels streams (region c: the last tail_len[c] items of the tail area + the chunk's fill[c]
// items), while the channels go on filling buffer b ^ 1.  What a pass did not consume of a channel (an unfinished packet) is
// copied in front of that channel's next chunk when the pass is collected.  Decoder state crossing passes (d_phdr.cr, power
// queue, absolute position) is carried per channel exactly as lora_hip_work carries it for its one stream.
struct lora_hip_mux {
    lora_hip_decoder *h = nullptr;   // tables, kernels, scheduler, frame queue
    uint32_t n = 0;
    size_t batch = 0, tailcap = 0, region = 0;
    DevBuf<float2> dbuf[2];
    int cur = 0;
    struct Chan { size_t fill = 0, tail_len = 0; uint32_t cr = 0; PwrState pwr; int64_t host_base = 0; std::vector<float2> ahead; size_t fl_off = 0, fl_len = 0; bool in_pass = false;
                  size_t ahead_off = 0; // items of `ahead` already uploaded (a read offset: the front is erased only once it is more than half of the vector)
                  size_t ahead_left() const { return ahead.size() - ahead_off; } };
    std::vector<Chan> ch;
    hipStream_t copy_st = nullptr, comp_st = nullptr;
    hipEvent_t up_ev = nullptr, tail_ev = nullptr;
    bool inflight = false;
    float max_latency_ms = 50.0f;
    size_t max_ahead = 0;            // surplus of one channel (items in host memory) at which a pass goes without waiting for the others
    std::chrono::steady_clock::time_point t_first;
    bool have_first = false;
    uint64_t passes = 0, passes_by_latency = 0;
    std::string err;
};

#define MUX_TRY(m, expr)                                                                                   \
    do {                                                                                                   \
        hipError_t e__ = (expr);                                                                           \
        if (e__ != hipSuccess) { (m)->err = std::string(#expr) + ": " + hipGetErrorString(e__); return LORA_HIP_ERR_HIP; } \
    } while (0)

lora_hip_status lora_hip_mux_create(const lora_hip_config_t *cfg, uint32_t n_channels, lora_hip_mux_t **out)
{
    if (!cfg || !out || n_channels == 0 || n_channels > 4096) return LORA_HIP_ERR_ARG;
    *out = nullptr;
    lora_hip_decoder *h = nullptr;
    lora_hip_status s = lora_hip_create(cfg, &h);
    if (s != LORA_HIP_OK) return s;
    lora_hip_mux *m = new (std::nothrow) lora_hip_mux();
    if (!m) { lora_hip_destroy(h); return LORA_HIP_ERR_NOMEM; }
    m->h = h; m->n = n_channels;
    m->batch = cfg->batch_items ? cfg->batch_items : std::max<size_t>(1u << 18, 64ull * h->P.sps); // (n channels share a pass: smaller chunks than one stream's)
    m->tailcap = std::max<size_t>(m->batch, 4u * (size_t)h->P.sps);
    m->region = m->tailcap + m->batch;
    m->max_ahead = std::max<size_t>(8u * m->batch, (size_t)1 << 22);
    m->ch.resize(n_channels);
    for (auto &c : m->ch) c.cr = h->P.ctor_cr;
    bool ok = hipSetDevice(h->device) == hipSuccess && hipStreamCreateWithFlags(&m->copy_st, hipStreamNonBlocking) == hipSuccess &&
              hipStreamCreateWithFlags(&m->comp_st, hipStreamNonBlocking) == hipSuccess && hipEventCreateWithFlags(&m->up_ev, hipEventDisableTiming) == hipSuccess &&
              hipEventCreateWithFlags(&m->tail_ev, hipEventDisableTiming) == hipSuccess;
    for (int i = 0; ok && i < 2; i++) ok = m->dbuf[i].reserve((size_t)n_channels * m->region) == hipSuccess;
    if (!ok) { lora_hip_mux_destroy(m); return LORA_HIP_ERR_HIP; }
    *out = m;
    return LORA_HIP_OK;
}

void lora_hip_mux_destroy(lora_hip_mux_t *m)
{
    if (!m) return;
    if (m->h) (void)hipSetDevice(m->h->device);
    if (m->copy_st) { (void)hipStreamSynchronize(m->copy_st); }
    if (m->comp_st) { (void)hipStreamSynchronize(m->comp_st); }
    for (int i = 0; i < 2; i++) m->dbuf[i].release();
    if (m->up_ev) (void)hipEventDestroy(m->up_ev);
    if (m->tail_ev) (void)hipEventDestroy(m->tail_ev);
    if (m->copy_st) (void)hipStreamDestroy(m->copy_st);
    if (m->comp_st) (void)hipStreamDestroy(m->comp_st);
    if (m->h) { m->h->pass_open = false; lora_hip_destroy(m->h); }
    delete m;
}

const char *lora_hip_mux_last_error(const lora_hip_mux_t *m) { return m ? (m->err.empty() ? m->h->err.c_str() : m->err.c_str()) : g_create_err.c_str(); }

// collects the pass in flight: per channel its decoder state, its frames (published through the handle's queue) and its tail
static lora_hip_status mux_collect(lora_hip_mux *m)
{
    if (!m->inflight) return LORA_HIP_OK;
    lora_hip_decoder *h = m->h;
    m->inflight = false;
    h->pass_open = false;
    h->err.clear();
    DeviceEnv env{h, h->pass_iq, h->pass_st};
    const int rc = lora_hip::decode_end(env, h->pass_streams, h->pass);
    if (rc != 0) { h->pending.open = false; m->err = h->err.empty() ? "scheduler failed" : h->err; return LORA_HIP_ERR_INTERNAL; }
    const int prev = m->cur ^ 1; // the buffer the pass ran on
    bool any_tail = false;
    for (const StreamDesc &sd : h->pass_streams) {
        lora_hip_mux::Chan &c = m->ch[sd.id];
        c.in_pass = false;
        c.cr = sd.cr_out; c.pwr = sd.pwr;
        const size_t keep_from = (size_t)std::min<int64_t>(std::max<int64_t>(sd.final_pos, 0), (int64_t)c.fl_len);
        const size_t tail = c.fl_len - keep_from;
        c.host_base += (int64_t)keep_from;
        if (tail > m->tailcap) { // a packet longer than the tail area: grow both buffers, every region keeps its chunk and the right end of its tail area
            const size_t ncap = std::max(2u * m->tailcap, tail + (size_t)h->P.sps), nregion = ncap + m->batch;
            MUX_TRY(m, hipStreamSynchronize(m->copy_st));
            MUX_TRY(m, hipStreamSynchronize(m->comp_st));
            for (int i = 0; i < 2; i++) {
                DevBuf<float2> nb;
                MUX_TRY(m, nb.reserve((size_t)m->n * nregion));
                for (uint32_t q = 0; q < m->n; q++)
                    MUX_TRY(m, hipMemcpyAsync(nb.p + (size_t)q * nregion + (ncap - m->tailcap), m->dbuf[i].p + (size_t)q * m->region, m->region * sizeof(float2), hipMemcpyDeviceToDevice, m->comp_st));
                MUX_TRY(m, hipStreamSynchronize(m->comp_st));
                std::swap(m->dbuf[i], nb);
                nb.release();
            }
            for (auto &cc : m->ch) cc.fl_off = (cc.fl_off / m->region) * nregion + (cc.fl_off % m->region) + (ncap - m->tailcap);
            m->tailcap = ncap; m->region = nregion;
        }
        if (tail) {
            MUX_TRY(m, hipMemcpyAsync(m->dbuf[m->cur].p + (size_t)sd.id * m->region + m->tailcap - tail, m->dbuf[prev].p + c.fl_off + keep_from, tail * sizeof(float2),
                                      hipMemcpyDeviceToDevice, m->comp_st));
            any_tail = true;
        }
        c.tail_len = tail;
    }
    if (any_tail) { // the sources sit in chunk areas the next uploads overwrite: they wait for these copies
        MUX_TRY(m, hipEventRecord(m->tail_ev, m->comp_st));
        MUX_TRY(m, hipStreamWaitEvent(m->copy_st, m->tail_ev, 0));
    }
    return LORA_HIP_OK;
}

static lora_hip_status mux_upload(lora_hip_mux *m, uint32_t c, const float2 *src, size_t n)
{
    lora_hip_mux::Chan &C = m->ch[c];
    MUX_TRY(m, hipMemcpyAsync(m->dbuf[m->cur].p + (size_t)c * m->region + m->tailcap + C.fill, src, n * sizeof(float2), hipMemcpyHostToDevice, m->copy_st));
    C.fill += n;
    return LORA_HIP_OK;
}

// launches a pass over what every channel holds (collecting the pass before it), then refills the new chunk from the surplus
static lora_hip_status mux_rotate(lora_hip_mux *m, bool by_latency)
{
    lora_hip_decoder *h = m->h;
    MUX_TRY(m, hipEventRecord(m->up_ev, m->copy_st));
    lora_hip_status s = mux_collect(m);
    if (s != LORA_HIP_OK) return s;
    std::vector<StreamDesc> &sds = h->pass_streams;
    sds.clear();
    uint64_t items = 0;
    for (uint32_t c = 0; c < m->n; c++) {
        lora_hip_mux::Chan &C = m->ch[c];
        const size_t len = C.tail_len + C.fill;
        if (len < 2u * (size_t)h->P.sps) continue; // (:91: not a work() call's worth yet; it stays where it is)
        StreamDesc sd{};
        sd.off = (uint64_t)c * m->region + m->tailcap - C.tail_len; sd.len = len; sd.id = c;
        sd.cr_in = C.cr; sd.pwr = C.pwr; sd.abs_base = C.host_base;
        sds.push_back(sd);
        C.fl_off = (size_t)sd.off; C.fl_len = len; C.in_pass = true;
        items += len;
    }
    if (sds.empty()) return LORA_HIP_OK;
    MUX_TRY(m, hipStreamWaitEvent(m->comp_st, m->up_ev, 0));
    h->timing = lora_hip_timing_t{};
    h->timing.items = items;
    h->pass_iq = m->dbuf[m->cur].p; h->pass_st = m->comp_st;
    h->err.clear();
    DeviceEnv env{h, h->pass_iq, h->pass_st};
    if (lora_hip::decode_begin(env, sds, h->pass) != 0) { h->pending.open = false; m->err = h->err.empty() ? "scheduler failed" : h->err; return LORA_HIP_ERR_INTERNAL; }
    h->pass_open = true;
    m->inflight = true;
    m->passes++; m->passes_by_latency += by_latency ? 1u : 0u;
    const int old = m->cur;
    m->cur ^= 1;
    m->have_first = false;
    for (uint32_t c = 0; c < m->n; c++) {
        lora_hip_mux::Chan &C = m->ch[c];
        if (!C.in_pass) { // too short to be decoded yet: its samples move along to the new buffer
            const size_t len = C.tail_len + C.fill;
            if (len) MUX_TRY(m, hipMemcpyAsync(m->dbuf[m->cur].p + (size_t)c * m->region + m->tailcap - len, m->dbuf[old].p + (size_t)c * m->region + m->tailcap - C.tail_len, len * sizeof(float2),
                                               hipMemcpyDeviceToDevice, m->copy_st));
            C.tail_len = len; C.fill = 0;
        } else { C.fill = 0; C.tail_len = 0; }
    }
    // what the channels delivered beyond their chunks: every upload is queued first, the copy stream is waited for ONCE, and only then do the
    // host vectors change (one synchronisation and one front erase per channel and rotation used to serialise n_channels waits and ~32 MB memmoves)
    std::vector<size_t> took(m->n, 0);
    bool any_up = false;
    for (uint32_t c = 0; c < m->n; c++) {
        lora_hip_mux::Chan &C = m->ch[c];
        if (C.ahead_left() == 0) continue;
        took[c] = std::min(C.ahead_left(), m->batch);
        s = mux_upload(m, c, C.ahead.data() + C.ahead_off, took[c]);
        if (s != LORA_HIP_OK) return s;
        any_up = true;
    }
    if (any_up) {
        MUX_TRY(m, hipStreamSynchronize(m->copy_st)); // (the vectors are about to change)
        for (uint32_t c = 0; c < m->n; c++)
            if (took[c]) {
                lora_hip_mux::Chan &C = m->ch[c];
                C.ahead_off += took[c];
                if (C.ahead_off == C.ahead.size()) { C.ahead.clear(); C.ahead_off = 0; }
                else if (C.ahead_off > C.ahead.size() / 2) { C.ahead.erase(C.ahead.begin(), C.ahead.begin() + (ptrdiff_t)C.ahead_off); C.ahead_off = 0; }
            }
        if (!m->have_first) { m->have_first = true; m->t_first = std::chrono::steady_clock::now(); }
    }
    return LORA_HIP_OK;
}

lora_hip_status lora_hip_mux_work(lora_hip_mux_t *m, uint32_t channel, const float *iq, size_t n_items)
{
    if (!m || channel >= m->n || (!iq && n_items)) return LORA_HIP_ERR_ARG;
    lora_hip_decoder *h = m->h;
    MUX_TRY(m, hipSetDevice(h->device));
    m->err.clear();
    lora_hip_status s;
    if (m->inflight && h->pending.open && hipEventQuery(h->ev_done) == hipSuccess) { // finished: publish now
        s = mux_collect(m);
        if (s != LORA_HIP_OK) return s;
    }
    (void)hipGetLastError();
    lora_hip_mux::Chan &C = m->ch[channel];
    const float2 *src = reinterpret_cast<const float2 *>(iq);
    size_t left = n_items;
    if (left && C.ahead_left() == 0) {
        const size_t k = std::min(left, m->batch - C.fill);
        if (k) {
            s = mux_upload(m, channel, src, k);
            if (s != LORA_HIP_OK) return s;
            if (!m->have_first) { m->have_first = true; m->t_first = std::chrono::steady_clock::now(); }
            src += k; left -= k;
        }
    }
    if (left) C.ahead.insert(C.ahead.end(), src, src + left); // this channel is a chunk ahead of the slowest one
    MUX_TRY(m, hipStreamSynchronize(m->copy_st)); // the caller may reuse its buffer
    for (;;) { // a pass when every channel's chunk is full (again, while the surplus refills whole chunks) - or when one channel's surplus
               // has reached max_ahead (a silent or stalled neighbour must not let it grow without bound when the latency bound is off:
               // the others then go into the pass with what they hold)
        bool all_full = true, far_ahead = false;
        for (const auto &c : m->ch) { all_full = all_full && c.fill == m->batch; far_ahead = far_ahead || (c.fill == m->batch && c.ahead_left() >= m->max_ahead); }
        if (!all_full && !far_ahead) break;
        const uint64_t before = m->passes;
        s = mux_rotate(m, false);
        if (s != LORA_HIP_OK) return s;
        if (m->passes == before) break; // (nothing could be launched)
    }
    if (m->max_latency_ms > 0.0f && m->have_first &&
        std::chrono::duration<float, std::milli>(std::chrono::steady_clock::now() - m->t_first).count() >= m->max_latency_ms) {
        bool any = false;
        for (const auto &c : m->ch) any = any || (c.fill != 0 && c.tail_len + c.fill >= 2u * (size_t)h->P.sps);
        if (any) { s = mux_rotate(m, true); if (s != LORA_HIP_OK) return s; }
    }
    return LORA_HIP_OK;
}

lora_hip_status lora_hip_mux_flush(lora_hip_mux_t *m)
{
    if (!m) return LORA_HIP_ERR_ARG;
    MUX_TRY(m, hipSetDevice(m->h->device));
    lora_hip_status s;
    for (int guard = 0; guard < 1 << 20; guard++) { // until nothing waits in host memory either
        const uint64_t before = m->passes;
        s = mux_rotate(m, false);
        if (s != LORA_HIP_OK) return s;
        bool more = false;
        for (const auto &c : m->ch) more = more || c.ahead_left() != 0;
        if (!more && m->passes == before) break; // nothing launched and nothing left to upload: what remains is shorter than a work() call (:91)
        if (!more) { // the last uploads are in: one more pass takes them
            bool any = false;
            for (const auto &c : m->ch) any = any || (c.fill != 0 && c.tail_len + c.fill >= 2u * (size_t)m->h->P.sps);
            if (!any) break;
        }
    }
    s = mux_collect(m);
    if (s != LORA_HIP_OK) return s;
    MUX_TRY(m, hipStreamSynchronize(m->comp_st));
    return LORA_HIP_OK;
}

lora_hip_status lora_hip_mux_set_latency(lora_hip_mux_t *m, float max_latency_ms)
{
    if (!m || !(max_latency_ms >= 0.0f)) return LORA_HIP_ERR_ARG;
    m->max_latency_ms = max_latency_ms;
    return LORA_HIP_OK;
}

lora_hip_status lora_hip_mux_set_max_ahead(lora_hip_mux_t *m, size_t max_ahead_items)
{
    if (!m) return LORA_HIP_ERR_ARG;
    m->max_ahead = max_ahead_items ? max_ahead_items : std::max<size_t>(8u * m->batch, (size_t)1 << 22);
    return LORA_HIP_OK;
}

size_t lora_hip_mux_frames_available(const lora_hip_mux_t *m) { return m ? m->h->frames.size() : 0; }

lora_hip_status lora_hip_mux_poll_frame(lora_hip_mux_t *m, uint8_t *buf, size_t cap, size_t *len, lora_hip_frame_info_t *info)
{
    return m ? lora_hip_poll_frame(m->h, buf, cap, len, info) : LORA_HIP_ERR_ARG;
}

lora_hip_status lora_hip_mux_passes(const lora_hip_mux_t *m, uint64_t *passes, uint64_t *passes_by_latency)
{
    if (!m) return LORA_HIP_ERR_ARG;
    if (passes) *passes = m->passes;
    if (passes_by_latency) *passes_by_latency = m->passes_by_latency;
    return LORA_HIP_OK;
}

size_t lora_hip_frames_available(const lora_hip_decoder_t *h) { return h ? h->frames.size() : 0; }

lora_hip_status lora_hip_poll_frame(lora_hip_decoder_t *h, uint8_t *buf, size_t cap, size_t *len, lora_hip_frame_info_t *info)
{
    if (!h || !len) return LORA_HIP_ERR_ARG;
    if (h->frames.empty()) { *len = 0; return LORA_HIP_OK; }
    const FrameQueue::Ref &f = h->frames.front();
    *len = f.info.length;
    if (!buf || cap < f.info.length) return LORA_HIP_ERR_OVERFLOW;
    std::memcpy(buf, h->frames.front_bytes(), f.info.length);
    if (info) *info = f.info;
    h->frames.pop();
    return LORA_HIP_OK;
}

lora_hip_status lora_hip_drain_frames(lora_hip_decoder_t *h, uint8_t *buf, size_t cap, lora_hip_frame_info_t *infos,
                                      size_t max_frames, size_t *n_frames)
{
    if (!h || !n_frames || (max_frames && (!buf || !infos))) return LORA_HIP_ERR_ARG;
    size_t n = 0, used = 0;
    while (n < max_frames && !h->frames.empty()) {
        const FrameQueue::Ref &f = h->frames.front();
        if (used + f.info.length > cap) break;
        std::memcpy(buf + used, h->frames.front_bytes(), f.info.length);
        infos[n] = f.info;
        used += f.info.length;
        n++;
        h->frames.pop();
    }
    *n_frames = n;
    return LORA_HIP_OK;
}

lora_hip_status lora_hip_drain_slots(lora_hip_decoder_t *h, uint8_t *slots, size_t slot_bytes, size_t max_slots, size_t *n_frames)
{
    if (!h || !n_frames || slot_bytes < 16u || (max_slots && !slots)) return LORA_HIP_ERR_ARG;
    size_t n = 0;
    while (n < max_slots && !h->frames.empty()) {
        const FrameQueue::Ref &f = h->frames.front();
        if (16u + (size_t)f.info.length > slot_bytes) return LORA_HIP_ERR_OVERFLOW;
        uint8_t *s = slots + n * slot_bytes;
        std::memcpy(s, &f.info.stream, 4);
        std::memcpy(s + 4, &f.info.length, 4);
        std::memcpy(s + 8, &f.info.header_pos, 8);
        std::memcpy(s + 16, h->frames.front_bytes(), f.info.length);
        std::memset(s + 16 + f.info.length, 0, slot_bytes - 16u - f.info.length);
        n++;
        h->frames.pop();
    }
    *n_frames = n;
    return LORA_HIP_OK;
}

lora_hip_status lora_hip_demod_symbols_ex_device(lora_hip_decoder_t *h, const void *d_iq, size_t total_items,
                                                 const int64_t *offsets, size_t n, int demod, uint32_t *bins_out,
                                                 int32_t *fine_out, void *hip_stream)
{
    if (!h || !d_iq || (n && (!offsets || !bins_out)) || demod < 0 || demod > 2) return LORA_HIP_ERR_ARG;
    if (n == 0) return LORA_HIP_OK;
    for (size_t i = 0; i < n; i++)
        if (offsets[i] < 0 || (uint64_t)offsets[i] + h->P.sps > total_items) return fail(h, LORA_HIP_ERR_ARG, "symbol %zu out of range", i);
    hipStream_t st = (hipStream_t)hip_stream;
    HIP_TRY(h, hipSetDevice(h->device));
    HIP_TRY(h, h->d_offsets.reserve(n));
    HIP_TRY(h, h->d_bins.reserve(2u * n)); // bins | fine
    const uint32_t grid = (uint32_t)std::min<size_t>(n, 2048);
    if (!h->P.ifreq_in_lds_1) HIP_TRY(h, h->d_scratch.reserve((size_t)grid * 2u * h->P.sps));
    HIP_TRY(h, hipMemcpyAsync(h->d_offsets.p, offsets, n * sizeof(int64_t), hipMemcpyHostToDevice, st));
    int32_t *d_fine = fine_out ? reinterpret_cast<int32_t *>(h->d_bins.p + n) : nullptr;
    DevParams P = h->P;
    if (demod != 0) P.demod_mode = (uint32_t)demod; // FFT vs FFT_COMPAT decides the bin fine_sync is run with
    if (launch_demod_symbols(P, (const float2 *)d_iq, h->d_offsets.p, (uint32_t)n, demod, h->d_bins.p, d_fine,
                             h->P.ifreq_in_lds_1 ? nullptr : h->d_scratch.p, st) != 0)
        return fail(h, fine_out ? LORA_HIP_ERR_BAD_CONFIG : LORA_HIP_ERR_HIP, "demod launch failed (fine_sync output needs one of the fast families: SF6 .. SF12 at decimation 8, SF6 .. SF9 at 4, SF7 .. SF9 at 2): %s",
                    hipGetErrorString(hipGetLastError()));
    HIP_TRY(h, hipMemcpyAsync(bins_out, h->d_bins.p, n * sizeof(uint32_t), hipMemcpyDeviceToHost, st));
    if (fine_out) HIP_TRY(h, hipMemcpyAsync(fine_out, d_fine, n * sizeof(int32_t), hipMemcpyDeviceToHost, st));
    HIP_TRY(h, hipStreamSynchronize(st));
    return LORA_HIP_OK;
}

lora_hip_status lora_hip_demod_symbols_device(lora_hip_decoder_t *h, const void *d_iq, size_t total_items,
                                              const int64_t *offsets, size_t n, int demod, uint32_t *bins_out,
                                              void *hip_stream)
{
    return lora_hip_demod_symbols_ex_device(h, d_iq, total_items, offsets, n, demod, bins_out, nullptr, hip_stream);
}

lora_hip_status lora_hip_estimate_cfo_device(lora_hip_decoder_t *h, const void *d_iq, size_t total_items, const int64_t *offsets, size_t n,
                                             int mode, float *cfo_hz_out, void *hip_stream)
{
    if (!h || !d_iq || (n && (!offsets || !cfo_hz_out)) || mode < 0 || mode > 1) return LORA_HIP_ERR_ARG;
    if (n == 0) return LORA_HIP_OK;
    if (h->P.sps < 258u) return fail(h, LORA_HIP_ERR_BAD_CONFIG, "the estimate reads sample 257 of the window: %u samples per symbol", h->P.sps);
    for (size_t i = 0; i < n; i++)
        if (offsets[i] < 0 || (uint64_t)offsets[i] + h->P.sps > total_items) return fail(h, LORA_HIP_ERR_ARG, "window %zu out of range", i);
    hipStream_t st = (hipStream_t)hip_stream;
    HIP_TRY(h, hipSetDevice(h->device));
    HIP_TRY(h, h->d_offsets.reserve(n));
    HIP_TRY(h, h->d_bins.reserve(n)); // one float per window
    HIP_TRY(h, hipMemcpyAsync(h->d_offsets.p, offsets, n * sizeof(int64_t), hipMemcpyHostToDevice, st));
    if (launch_cfo(h->P, (const float2 *)d_iq, h->d_offsets.p, (uint32_t)n, mode, reinterpret_cast<float *>(h->d_bins.p), st) != 0)
        return fail(h, LORA_HIP_ERR_HIP, "cfo launch failed: %s", hipGetErrorString(hipGetLastError()));
    HIP_TRY(h, hipMemcpyAsync(cfo_hz_out, h->d_bins.p, n * sizeof(float), hipMemcpyDeviceToHost, st));
    HIP_TRY(h, hipStreamSynchronize(st));
    return LORA_HIP_OK;
}

lora_hip_status lora_hip_ref_ifreq_device(lora_hip_decoder_t *h, const void *d_iq, size_t n_items, float *arg_out, float *ifreq_out, void *hip_stream)
{
    if (!h || !d_iq || !arg_out || !ifreq_out || n_items < 2 || n_items > 0x7fffffffu) return LORA_HIP_ERR_ARG;
    hipStream_t st = (hipStream_t)hip_stream;
    HIP_TRY(h, hipSetDevice(h->device));
    HIP_TRY(h, h->d_scratch.reserve(2u * n_items));
    if (launch_ref_ifreq((const float2 *)d_iq, (uint32_t)n_items, h->d_scratch.p, h->d_scratch.p + n_items, st) != 0)
        return fail(h, LORA_HIP_ERR_HIP, "ref_ifreq launch failed: %s", hipGetErrorString(hipGetLastError()));
    HIP_TRY(h, hipMemcpyAsync(arg_out, h->d_scratch.p, n_items * sizeof(float), hipMemcpyDeviceToHost, st));
    HIP_TRY(h, hipMemcpyAsync(ifreq_out, h->d_scratch.p + n_items, (n_items - 1u) * sizeof(float), hipMemcpyDeviceToHost, st));
    HIP_TRY(h, hipStreamSynchronize(st));
    return LORA_HIP_OK;
}

// ---- FFT-domain preamble detection (definition: oracle/preamble_oracle.py) ----------------------------
static lora_hip_status window_stats_abs(lora_hip_decoder_t *h, const float2 *d_iq, const std::vector<int64_t> &offsets, std::vector<lora_hip_window_stats_t> &out, hipStream_t st)
{
    static_assert(sizeof(lora_hip_window_stats_t) == 24, "kernel record layout");
    const size_t n = offsets.size();
    out.resize(n);
    if (n == 0) return LORA_HIP_OK;
    HIP_TRY(h, h->d_offsets.reserve(n));
    HIP_TRY(h, h->d_bins.reserve(6u * n));
    HIP_TRY(h, hipMemcpyAsync(h->d_offsets.p, offsets.data(), n * sizeof(int64_t), hipMemcpyHostToDevice, st));
    if (launch_detect_windows(h->P, d_iq, h->d_offsets.p, (uint32_t)n, h->d_bins.p, st) != 0)
        return fail(h, LORA_HIP_ERR_HIP, "detect launch failed: %s", hipGetErrorString(hipGetLastError()));
    HIP_TRY(h, hipMemcpyAsync(out.data(), h->d_bins.p, n * sizeof(lora_hip_window_stats_t), hipMemcpyDeviceToHost, st));
    HIP_TRY(h, hipStreamSynchronize(st));
    return LORA_HIP_OK;
}

lora_hip_status lora_hip_window_stats_device(lora_hip_decoder_t *h, const void *d_iq, size_t total_items, const int64_t *offsets, size_t n,
                                             lora_hip_window_stats_t *out, void *hip_stream)
{
    if (!h || !d_iq || (n && (!offsets || !out))) return LORA_HIP_ERR_ARG;
    for (size_t i = 0; i < n; i++)
        if (offsets[i] < 0 || (uint64_t)offsets[i] + h->P.sps > total_items) return fail(h, LORA_HIP_ERR_ARG, "window %zu out of range", i);
    HIP_TRY(h, hipSetDevice(h->device));
    std::vector<int64_t> offs(offsets, offsets + n);
    std::vector<lora_hip_window_stats_t> res;
    const lora_hip_status s = window_stats_abs(h, (const float2 *)d_iq, offs, res, (hipStream_t)hip_stream);
    if (s != LORA_HIP_OK) return s;
    std::copy(res.begin(), res.end(), out);
    return LORA_HIP_OK;
}

lora_hip_status lora_hip_detect_preambles_device(lora_hip_decoder_t *h, const void *d_iq, size_t total_items, const uint64_t *stream_off,
                                                 const uint64_t *stream_len, uint32_t n_streams, float threshold, lora_hip_preamble_t *out, size_t cap,
                                                 size_t *n_found, void *hip_stream)
{
    if (!h || !d_iq || !n_found || (n_streams && (!stream_off || !stream_len)) || (cap && !out) || !(threshold >= 0.0f)) return LORA_HIP_ERR_ARG;
    *n_found = 0;
    HIP_TRY(h, hipSetDevice(h->device));
    const int64_t sps = h->P.sps, N = h->P.nbins, D = h->P.decim;
    const float thr = threshold > 0.0f ? threshold : (float)(std::log((double)N) + 3.0);
    constexpr int64_t kMinRun = 4, kSfdReach = 6;
    hipStream_t st = (hipStream_t)hip_stream;
    auto pmr = [&](float peak, float total) { const float rest = total - peak; return rest > 0.0f ? peak * (float)(N - 1) / rest : INFINITY; };
    auto circ = [&](int32_t a, int32_t b) { int64_t d = std::llabs((int64_t)a - b) % N; return std::min(d, N - d); };
    // stage A: one window per symbol of every stream, one launch
    std::vector<int64_t> offs;
    std::vector<size_t> first(n_streams + 1u, 0);
    for (uint32_t s = 0; s < n_streams; s++) {
        if (stream_off[s] > total_items || stream_len[s] > total_items - stream_off[s]) return fail(h, LORA_HIP_ERR_ARG, "stream %u exceeds the buffer", s);
        const int64_t K = (int64_t)(stream_len[s] / (uint64_t)sps) - 1;
        first[s] = offs.size();
        for (int64_t k = 0; k < K; k++) offs.push_back((int64_t)stream_off[s] + k * sps);
    }
    first[n_streams] = offs.size();
    std::vector<lora_hip_window_stats_t> A;
    lora_hip_status rc = window_stats_abs(h, (const float2 *)d_iq, offs, A, st);
    if (rc != LORA_HIP_OK) return rc;
    // runs, per stream
    struct Cand { uint32_t stream; int64_t k, e, a0; int32_t bin; float pmr; size_t b_first, b_n; };
    std::vector<Cand> cands;
    std::vector<int64_t> boffs;
    for (uint32_t s = 0; s < n_streams; s++) {
        const lora_hip_window_stats_t *W = A.data() + first[s];
        const int64_t K = (int64_t)(first[s + 1] - first[s]), len = (int64_t)stream_len[s];
        int64_t k = 0;
        while (k < K) {
            if (!(pmr(W[k].peak_down, W[k].total_down) >= thr)) { k++; continue; }
            int64_t e = k + 1;
            while (e < K && pmr(W[e].peak_down, W[e].total_down) >= thr && circ(W[e].bin_down, W[e - 1].bin_down) <= 1) e++;
            if (e - k >= kMinRun) {
                int32_t best = W[k].bin_down; int64_t best_n = 0; // most frequent bin; ties: the smallest
                for (int64_t i = k; i < e; i++) {
                    int64_t c = 0;
                    for (int64_t j = k; j < e; j++) c += W[j].bin_down == W[i].bin_down;
                    if (c > best_n || (c == best_n && W[i].bin_down < best)) { best_n = c; best = W[i].bin_down; }
                }
                float pm = 0.0f;
                for (int64_t i = k; i < e; i++) pm += pmr(W[i].peak_down, W[i].total_down);
                int64_t a0 = k * sps - (int64_t)best * D;
                if (a0 < 0) a0 += sps;
                Cand c{s, k, e, a0, best, pm / (float)(e - k), boffs.size(), 0};
                for (int64_t j = 0; j <= (e - k) + kSfdReach; j++) {
                    const int64_t p = a0 + j * sps;
                    if (p + sps > len) break;
                    boffs.push_back((int64_t)stream_off[s] + p);
                    c.b_n++;
                }
                cands.push_back(c);
            }
            k = e;
        }
    }
    // stage B: the aligned windows of every candidate, one launch
    std::vector<lora_hip_window_stats_t> B;
    rc = window_stats_abs(h, (const float2 *)d_iq, boffs, B, st);
    if (rc != LORA_HIP_OK) return rc;
    // the SFD of every candidate (stage B), then stage C: sub-bin timing of the ones that have one
    struct Hit { size_t cand; int64_t found; size_t c_first, c_n; int64_t j0, j1; };
    std::vector<Hit> hits;
    std::vector<int64_t> coffs;
    constexpr int64_t kRefineWindows = 6;
    int64_t skip_until = -1; uint32_t skip_stream = 0xffffffffu;
    for (size_t ci = 0; ci < cands.size(); ci++) {
        const Cand &c = cands[ci];
        if (c.stream == skip_stream && c.k < skip_until) continue; // (a run that begins inside the packet just reported)
        const lora_hip_window_stats_t *W = B.data() + c.b_first;
        int64_t found = -1;
        for (size_t j = 0; j + 1 < c.b_n; j++) {
            if (pmr(W[j].peak_up, W[j].total_up) >= thr && W[j].peak_up > W[j].peak_down && pmr(W[j + 1].peak_up, W[j + 1].total_up) >= thr &&
                W[j + 1].peak_up > W[j + 1].peak_down) { found = (int64_t)j; break; }
        }
        if (found < 0) continue;
        Hit hit{ci, found, coffs.size(), 0, std::max<int64_t>(0, found - 2 - kRefineWindows), found - 2};
        while (hit.j0 < hit.j1 && c.a0 + hit.j0 * sps - D / 2 < 0) hit.j0++;
        for (int64_t dl = -(D / 2); dl <= D / 2 && hit.j0 < hit.j1; dl++)
            for (int64_t j = hit.j0; j < hit.j1; j++) { coffs.push_back((int64_t)stream_off[c.stream] + c.a0 + j * sps + dl); hit.c_n++; }
        hits.push_back(hit);
        skip_stream = c.stream; skip_until = (c.a0 + (found + 2) * sps) / sps;
    }
    std::vector<lora_hip_window_stats_t> Cst;
    rc = window_stats_abs(h, (const float2 *)d_iq, coffs, Cst, st);
    if (rc != LORA_HIP_OK) return rc;
    size_t n_out = 0;
    bool overflow = false;
    for (const Hit &hit : hits) {
        const Cand &c = cands[hit.cand];
        const lora_hip_window_stats_t *W = B.data() + c.b_first;
        int64_t delta = 0;
        if (hit.c_n) { // P(delta) = power of bin 0 over the preamble windows moved by delta; the delta with the largest 3-point sum
            const int64_t nj = hit.j1 - hit.j0, nd = D + 1;
            std::vector<double> P((size_t)nd, 0.0);
            for (int64_t i = 0; i < nd; i++)
                for (int64_t j = 0; j < nj; j++) {
                    const lora_hip_window_stats_t &w = Cst[hit.c_first + (size_t)(i * nj + j)];
                    if (w.bin_down == 0) P[(size_t)i] += (double)w.peak_down;
                }
            double best = -1.0;
            for (int64_t i = 0; i < nd; i++) {
                const double v = P[(size_t)i] + (i >= 1 ? P[(size_t)i - 1] : 0.0) + (i + 1 < nd ? P[(size_t)i + 1] : 0.0);
                const int64_t d = i - D / 2;
                const bool closer = std::llabs(d) < std::llabs(delta) || (std::llabs(d) == std::llabs(delta) && d < delta);
                if (v > best || (v == best && closer)) { best = v; delta = d; }
            }
            if (!(best > 0.0)) delta = 0;
        }
        const int64_t found = hit.found;
        const int32_t bu = W[found].bin_up, sb = bu < N / 2 ? bu : bu - (int32_t)N;
        if (n_out < cap) {
            lora_hip_preamble_t &o = out[n_out];
            o.header_pos = c.a0 + delta + found * sps + 2 * sps + sps / 4; o.run_pos = c.k * sps; o.stream = c.stream; o.run_len = (uint32_t)(c.e - c.k);
            o.bin = c.bin; o.sfd_index = (int32_t)found; o.pmr = c.pmr; o.cfo_bins = -0.5f * (float)sb;
            o.cfo_hz = o.cfo_bins * (float)h->cfg.bandwidth / (float)N; o.delta = (int32_t)delta;
        } else overflow = true;
        n_out++;
    }
    *n_found = n_out;
    return overflow ? LORA_HIP_ERR_OVERFLOW : LORA_HIP_OK;
}

lora_hip_status lora_hip_decode_at_headers_device(lora_hip_decoder_t *h, const void *d_iq, size_t total_items, const uint64_t *stream_off,
                                                  const uint64_t *stream_len, uint32_t n_streams, const lora_hip_preamble_t *pre, size_t n, void *hip_stream)
{
    if (!h || !d_iq || !n_streams || !stream_off || !stream_len || (n && !pre)) return LORA_HIP_ERR_ARG;
    if (h->pass_open) return fail(h, LORA_HIP_ERR_ARG, "lora_hip_decode_at_headers_device: a pass is open on this handle");
    if (h->P.implicit) return fail(h, LORA_HIP_ERR_BAD_CONFIG, "decode at given headers needs an explicit header (the implicit mode's energy threshold comes from DETECT)");
    if (n == 0) return LORA_HIP_OK;
    HIP_TRY(h, hipSetDevice(h->device));
    std::vector<Job> jobs(n, Job{});
    for (size_t i = 0; i < n; i++) {
        const uint32_t s = pre[i].stream;
        if (s >= n_streams || stream_off[s] > total_items || stream_len[s] > total_items - stream_off[s]) return fail(h, LORA_HIP_ERR_ARG, "entry %zu: bad stream", i);
        if (pre[i].header_pos < 0 || (uint64_t)pre[i].header_pos + 2u * h->P.sps > stream_len[s]) return fail(h, LORA_HIP_ERR_ARG, "entry %zu: header position outside its stream", i);
        Job &j = jobs[i];
        j.stream_off = stream_off[s]; j.stream_len = stream_len[s]; j.stream_id = s;
        j.start = pre[i].header_pos; j.scan_limit = pre[i].header_pos; // (no DETECT step once the packet is done)
        j.cr_prev = h->P.ctor_cr; j.max_attempts = 1; j.start_at_header = 1;
    }
    h->err.clear();
    h->timing = lora_hip_timing_t{};
    RunOut &out = h->run_out[0];
    const lora_hip_status rc = run_jobs(h, (const float2 *)d_iq, jobs, 2, 0, (hipStream_t)hip_stream, out);
    if (rc != LORA_HIP_OK) return rc;
    for (size_t i = 0; i < n; i++) {
        if (out.res[i].n_attempts == 0 || out.rpj == 0) continue;
        const AttemptRec &r = out.rec(i, 0);
        if (r.status != kAttemptFrame) continue; // (ran out of data mid-packet)
        StreamDesc sd{};
        sd.id = jobs[i].stream_id; sd.abs_base = 0;
        publish(h, r, sd);
    }
    return LORA_HIP_OK;
}

lora_hip_status lora_hip_last_plan(const lora_hip_decoder_t *h, uint32_t *burst_aware, uint32_t *segments)
{
    if (!h) return LORA_HIP_ERR_ARG;
    if (burst_aware) *burst_aware = h->last_plan_burst;
    if (segments) *segments = h->last_plan_segments;
    return LORA_HIP_OK;
}

// (the variant the last pass launched where the kernel exists in two workgroup sizes - walker3 SF9 / SF10: *_half with more jobs than CUs)
lora_hip_status lora_hip_last_payload_pass(const lora_hip_decoder_t *h, uint32_t *packets, uint32_t *moved, uint32_t *rerun, uint32_t *rounds, uint32_t *symbols, float *ms)
{
    if (!h) return LORA_HIP_ERR_ARG;
    if (packets) *packets = h->last_payload_packets;
    if (moved) *moved = h->last_payload_moved;
    if (rounds) *rounds = h->last_payload_rounds;
    if (rerun) *rerun = h->last_payload_rerun;
    if (symbols) *symbols = h->last_payload_symbols;
    if (ms) *ms = h->last_payload_ms;
    return LORA_HIP_OK;
}

const char *lora_hip_walker_kernel_name(const lora_hip_decoder_t *h) { return h ? (h->last_kernel ? h->last_kernel : walker_kernel_name(h->P)) : ""; }

lora_hip_status lora_hip_last_timing(const lora_hip_decoder_t *h, lora_hip_timing_t *t)
{
    if (!h || !t) return LORA_HIP_ERR_ARG;
    *t = h->timing;
    return LORA_HIP_OK;
}

size_t lora_hip_trace(const lora_hip_decoder_t *h, const lora_hip_step_t **steps)
{
    if (!h || !steps) return 0;
    *steps = h->trace.data();
    return h->trace.size();
}

void lora_hip_trace_clear(lora_hip_decoder_t *h)
{
    if (h) h->trace.clear();
}

} // extern "C"
