// lora_channelizer.hip -- MI355X channeliser: frequency-translating FIR + decimation in front of the decoder
// (SURVEY 8(f) N1).  C ABI in include/lora_hip_channelizer.h.
//
// Reference: gr::lora::channelizer (lib/channelizer_impl.cc:46-57) = GNU Radio 3.9's
// freq_xlating_fir_filter_ccf(decimation, firdes::low_pass(1, fs, bw/2 + 15 kHz, 10 kHz, WIN_HAMMING), f, fs).
// GNU Radio is not vendored in the reference tree; its published algorithm is: band-pass taps
// bp[k] = h[k] e^{+j 2 pi f k / fs}, decimating FIR over the input, then a rotator e^{-j 2 pi f D m / fs} on the
// output.  That is, exactly,
//     y[m] = sum_k h[k] x'[mD - k],    x'[n] = x[n] e^{-j 2 pi f n / fs},    x[n < 0] = 0
// -- mix first, then a REAL-tap FIR: 2 FMAs per tap and output instead of 4, which is what this kernel computes.
//
// Kernel (fir_mix_kernel).  The work is fp32 vector FMA work (decimation 1, 241 taps: 964 flop per 16 bytes moved,
// far above the HBM ridge), so the design goal is FMA issue rate:
//   * a workgroup stages kTileIn (+ taps - 1) input samples into LDS, already mixed: x[n] * W[i] with W[i] =
//     e^{-j 2 pi f i / fs} for the offset i inside the tile (float table built in double on the host); the tile's
//     base phasor e^{-j 2 pi f n0 / fs} is evaluated in double once per tile and applied to the OUTPUTS
//     (the FIR is linear), so no per-sample sincos and no phase drift however long the stream is;
//   * decimation 1: each thread produces R = 16 consecutive outputs from a register window that slides one sample
//     per tap (one 8-byte LDS read per 32 FMAs); the taps are wave-uniform scalar loads (SGPR operands);
//   * decimation D > 1 (any D <= 64): one output per thread, one LDS read per tap -- correct but LDS-bound (4.5 TFLOP/s);
//     the reference's and gr-lora's apps' operating point is decimation 1.
// Streaming state (filter history, absolute sample index for the oscillator, decimation phase) lives in the handle.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <cstddef>
#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <string>
#include <vector>

#include "../../include/lora_hip_channelizer.h"

namespace {

constexpr int kThreads = 256;
constexpr int kOutD1 = 16;    // outputs per thread at decimation 1 (tile = 4096 input items); any other decimation: 1 (tile = 256 D)
constexpr int kMaxDecim = 64;

// gr::filter::firdes::low_pass(gain, fs, cutoff, transition, WIN_HAMMING) as published (GNU Radio 3.9
// gr-filter/lib/firdes.cc): ntaps = (int)(53 * fs / (22 * transition)) made odd (the Hamming window's 53 dB),
// h[n] = window[n] * (n == M ? fwT0 / pi : sin((n - M) fwT0) / ((n - M) pi)), scaled to `gain` at DC.
std::vector<float> firdes_low_pass(double gain, double fs, double cutoff, double transition)
{
    int ntaps = (int)(53.0 * fs / (22.0 * transition));
    if ((ntaps & 1) == 0) ntaps++;
    const int M = (ntaps - 1) / 2;
    const double fwT0 = 2.0 * M_PI * cutoff / fs;
    std::vector<float> taps(ntaps);
    std::vector<float> w(ntaps);
    for (int n = 0; n < ntaps; n++) w[n] = (float)(0.54 - 0.46 * std::cos(2.0 * M_PI * n / (ntaps - 1))); // fft::window::hamming (float)
    for (int n = -M; n <= M; n++) {
        if (n == 0) taps[n + M] = (float)(fwT0 / M_PI * w[n + M]);
        else taps[n + M] = (float)(std::sin(n * fwT0) / (n * M_PI) * w[n + M]);
    }
    double fmax = taps[M];
    for (int n = 1; n <= M; n++) fmax += 2.0 * taps[n + M];
    const double g = gain / fmax;
    for (int n = 0; n < ntaps; n++) taps[n] = (float)(taps[n] * g);
    return taps;
}

struct ChanParams {
    double turns_per_sample; // f / fs (translation frequency in cycles per input sample)
    double phase0;           // oscillator phase (turns) at absolute sample index n_ref
    long long n_ref;
};

struct FirArgs {
    const float2 *in;      // new input items
    const float2 *hist;    // the ntaps - 1 items before in[0]
    float2 *out;           // n_channels rows of out_stride
    const float *taps;     // h[0 .. ntaps)
    const float2 *wtab;    // per channel: kTileIn + ntaps - 1 entries e^{-j 2 pi f i / fs}
    const ChanParams *chan;
    long long n_abs;       // absolute index of in[0]
    long long n_in;
    long long first;       // local index of the first output's input sample (decimation phase)
    long long n_out;
    long long out_stride;
    int ntaps, decim, wtab_stride; // ntaps: tap count padded with zeros to a multiple of 16
    int tile_in;                   // input items advanced per workgroup = 256 * outputs per thread * decimation
    int nhist;                     // items in hist (= real tap count - 1)
};

// LDS layout of the staged samples: one padding slot after every 16 samples.  At decimation 1 lane l reads around sample
// 16 l, i.e. a lane stride of 128 bytes -- every lane on the same banks, a 32-way conflict; with the padding the stride
// is 136 bytes and the 32 lanes of an LDS access group fall on distinct banks.
__device__ __host__ __forceinline__ int xs_slot(int i) { return i + (i >> 4); }

__device__ __forceinline__ float2 cmulf(float2 a, float2 b) { return make_float2(a.x * b.x - a.y * b.y, a.x * b.y + a.y * b.x); }

// One workgroup: outputs whose input positions fall in [first + tile * tile_in, + tile_in).
template <int R, bool DECIM1>
__global__ __launch_bounds__(kThreads) void fir_mix_kernel(FirArgs A)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    float2 *xs = reinterpret_cast<float2 *>(smem); // mixed samples: xs[i] = x'[t0 - (ntaps - 1) + i]
    const int T = A.ntaps, D = A.decim;
    const int c = blockIdx.y;
    const long long t0 = A.first + (long long)blockIdx.x * A.tile_in; // local index of the tile's first output position
    const int span = A.tile_in + T - 1;
    const float2 *__restrict__ w = A.wtab + (size_t)c * A.wtab_stride;
    for (int i = threadIdx.x; i < span; i += kThreads) {
        const long long n = t0 - (T - 1) + i; // local input index
        float2 v = make_float2(0.f, 0.f);
        if (n >= 0) { if (n < A.n_in) v = A.in[n]; }
        else if (n >= -(long long)A.nhist) v = A.hist[A.nhist + n];
        xs[xs_slot(i)] = cmulf(v, w[i]);
    }
    // base phasor of the tile: e^{-j 2 pi (phase0 + f/fs * (n_abs + t0 - (T-1) - n_ref))}, in double
    __shared__ float2 base;
    if (threadIdx.x == 0) {
        const ChanParams cp = A.chan[c];
        const double turns = cp.phase0 + cp.turns_per_sample * (double)(A.n_abs + t0 - (T - 1) - cp.n_ref);
        const double fr = turns - floor(turns);
        double s, co;
        sincospi(-2.0 * fr, &s, &co);
        base = make_float2((float)co, (float)s);
    }
    __syncthreads();
    const float *__restrict__ h = A.taps;
    const long long m0 = (long long)blockIdx.x * (kThreads * R) + (long long)threadIdx.x * R; // first output of this thread
    float2 acc[R];
#pragma unroll
    for (int r = 0; r < R; r++) acc[r] = make_float2(0.f, 0.f);
    if constexpr (DECIM1) {
        // output j0 + r = sum_k h[k] xs[j0 + r + (T-1) - k].  Taps in blocks of R (T is padded to a multiple of R): inside a
        // block the sample for (r, kk) is win[r - kk + R - 1], a static register index -- no window shifting per tap;
        // between blocks the ring moves down by R samples (R - 1 register moves and R LDS reads per R*R packed FMAs).
        const int j0 = threadIdx.x * R;
        float2 win[2 * R - 1];
        int xb = j0 + (T - 1) - (R - 1); // sample index of win[0]
#pragma unroll
        for (int i = 0; i < 2 * R - 1; i++) win[i] = xs[xs_slot(xb + i)];
        for (int kb = 0; kb < T; kb += R) {
#pragma unroll
            for (int kk = 0; kk < R; kk++) {
                const float hk = h[kb + kk];
#pragma unroll
                for (int r = 0; r < R; r++) {
                    const float2 v = win[r - kk + R - 1];
                    acc[r].x = fmaf(hk, v.x, acc[r].x); acc[r].y = fmaf(hk, v.y, acc[r].y);
                }
            }
            if (kb + R < T) {
                xb -= R;
#pragma unroll
                for (int i = 2 * R - 2; i >= R; i--) win[i] = win[i - R];
#pragma unroll
                for (int i = 0; i < R; i++) win[i] = xs[xs_slot(xb + i)];
            }
        }
    } else {
        const int j0 = threadIdx.x * R * D; // input offset of this thread's first output inside the tile
        for (int k = 0; k < T; k++) {
            const float hk = h[k];
#pragma unroll
            for (int r = 0; r < R; r++) {
                const float2 v = xs[xs_slot(j0 + r * D + (T - 1) - k)];
                acc[r].x = fmaf(hk, v.x, acc[r].x); acc[r].y = fmaf(hk, v.y, acc[r].y);
            }
        }
    }
    float2 *__restrict__ o = A.out + (size_t)c * A.out_stride;
#pragma unroll
    for (int r = 0; r < R; r++) {
        const long long m = m0 + r;
        if (m < A.n_out) o[m] = cmulf(acc[r], base);
    }
}

} // namespace

struct lora_hip_channelizer {
    lora_hip_channelizer_config_t cfg{};
    std::vector<float> channels;
    std::vector<float> taps;       // the reference's d_lpf
    int ntaps_pad = 0;             // taps.size() rounded up to a multiple of 16 (zero taps at the old end)
    int tile_in = 0;               // input items per workgroup
    std::vector<ChanParams> chan;
    float cfo = 0.0f;          // d_cfo is a float upstream (channelizer_impl.h:37)
    bool cfo_applied = false;  // apply_cfo has run: the frequency is d_freq_offset + d_cfo in float from then on (:70)
    int device = 0;
    long long n_abs = 0;       // input items consumed so far
    float *d_taps = nullptr;
    float2 *d_wtab = nullptr, *d_hist = nullptr, *d_stage_in = nullptr, *d_stage_out = nullptr;
    ChanParams *d_chan = nullptr;
    size_t stage_in_cap = 0, stage_out_cap = 0;
    int wtab_stride = 0;
    hipEvent_t ev0 = nullptr, ev1 = nullptr;
    float last_ms = 0.0f;
    std::string err;
};

namespace {

lora_hip_status cfail(lora_hip_channelizer *h, lora_hip_status s, const char *fmt, ...)
{
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    if (h) h->err = buf;
    return s;
}
#define CH_TRY(h, call)                                                                                    \
    do {                                                                                                   \
        hipError_t e_ = (call);                                                                            \
        if (e_ != hipSuccess) return cfail((h), LORA_HIP_ERR_HIP, "%s: %s", #call, hipGetErrorString(e_)); \
    } while (0)

// (re)builds the per-channel oscillator tables for the current frequencies; phase stays continuous at n_abs
lora_hip_status upload_channels(lora_hip_channelizer *h, bool first)
{
    const int T = h->ntaps_pad;
    const int span = h->tile_in + T - 1;
    h->wtab_stride = span;
    std::vector<float2> w((size_t)h->channels.size() * span);
    for (size_t c = 0; c < h->channels.size(); c++) {
        // d_freq_offset = channel_list[0] - center_freq: a float subtraction stored in a uint32_t (:47, channelizer_impl.h:39),
        // i.e. truncated to whole Hz; a negative offset keeps its sign here (upstream it wraps: unsigned field, UB conversion)
        double f = std::trunc((double)(float)(h->channels[c] - h->cfg.center_freq)) + (double)h->cfo; // + d_cfo (:70)
        if (h->cfg.flags & LORA_HIP_CHANNELIZER_FLAG_UINT32_OFFSET) { // upstream's own arithmetic: the unsigned field (a negative offset wraps), the CFO added in float
            const uint32_t u = (uint32_t)(int64_t)(float)(h->channels[c] - h->cfg.center_freq);
            f = h->cfo_applied ? (double)((float)u + h->cfo) : (double)u;
        }
        const double tps = f / (double)h->cfg.samp_rate;
        if (!first) { // keep the oscillator phase continuous across the change
            ChanParams &cp = h->chan[c];
            const double t = cp.phase0 + cp.turns_per_sample * (double)(h->n_abs - cp.n_ref);
            cp.phase0 = t - std::floor(t);
            cp.n_ref = h->n_abs;
            cp.turns_per_sample = tps;
        } else {
            h->chan[c] = ChanParams{tps, 0.0, 0};
        }
        for (int i = 0; i < span; i++) {
            const double t = tps * (double)i;
            const double a = -2.0 * M_PI * (t - std::floor(t));
            w[c * span + i] = make_float2((float)std::cos(a), (float)std::sin(a));
        }
    }
    CH_TRY(h, hipMemcpy(h->d_wtab, w.data(), w.size() * sizeof(float2), hipMemcpyHostToDevice));
    CH_TRY(h, hipMemcpy(h->d_chan, h->chan.data(), h->chan.size() * sizeof(ChanParams), hipMemcpyHostToDevice));
    return LORA_HIP_OK;
}

template <int R, bool D1>
void launch_fir(const FirArgs &a, int n_channels, hipStream_t st)
{
    const long long span_in = (a.n_out - 1) * (long long)a.decim + 1;                // input positions covered
    const unsigned tiles = (unsigned)((span_in + a.tile_in - 1) / a.tile_in);
    const size_t lds = (size_t)(xs_slot(a.tile_in + a.ntaps - 1) + 1) * sizeof(float2);
    if (lds > 64u * 1024u) (void)hipFuncSetAttribute((const void *)fir_mix_kernel<R, D1>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipLaunchKernelGGL((fir_mix_kernel<R, D1>), dim3(tiles, (unsigned)n_channels), dim3(kThreads), lds, st, a);
}

} // namespace

extern "C" {

lora_hip_status lora_hip_channelizer_create(const lora_hip_channelizer_config_t *cfg, lora_hip_channelizer_t **out)
{
    if (!cfg || !out || cfg->struct_size < offsetof(lora_hip_channelizer_config_t, cutoff_hz)) return LORA_HIP_ERR_ARG;
    *out = nullptr;
    if (!cfg->channel_list || cfg->n_channels == 0 || cfg->decimation == 0 || !(cfg->samp_rate > 0.0f)) return LORA_HIP_ERR_BAD_CONFIG;
    if (cfg->decimation > (uint32_t)kMaxDecim) return LORA_HIP_ERR_BAD_CONFIG; // the generic path stages 256 D + taps items in LDS
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || cfg->device < 0 || cfg->device >= ndev) return LORA_HIP_ERR_NO_DEVICE;
    auto *h = new lora_hip_channelizer;
    std::memcpy(&h->cfg, cfg, std::min<size_t>(cfg->struct_size, sizeof h->cfg)); // older callers: no design overrides
    h->device = cfg->device;
    h->channels.assign(cfg->channel_list, cfg->channel_list + cfg->n_channels);
    h->cfg.channel_list = nullptr;
    const double cutoff = h->cfg.cutoff_hz > 0.0f ? (double)h->cfg.cutoff_hz : (double)(cfg->bandwidth / 2u) + 15000.0; // :46 (integer bandwidth/2)
    const double transition = h->cfg.transition_hz > 0.0f ? (double)h->cfg.transition_hz : 10000.0;
    h->taps = firdes_low_pass(1.0, cfg->samp_rate, cutoff, transition);
    h->chan.resize(h->channels.size());
    h->ntaps_pad = ((int)h->taps.size() + 15) & ~15;
    h->tile_in = kThreads * (cfg->decimation == 1 ? kOutD1 : 1) * (int)cfg->decimation;
    const int T = h->ntaps_pad;
    std::vector<float> padded(h->taps);
    padded.resize((size_t)T, 0.0f);
    const size_t nh = h->taps.size() - 1;
    lora_hip_status s = LORA_HIP_OK;
    do {
        if (hipSetDevice(h->device) != hipSuccess) { s = LORA_HIP_ERR_NO_DEVICE; break; }
        if (hipMalloc((void **)&h->d_taps, T * sizeof(float)) != hipSuccess ||
            hipMalloc((void **)&h->d_wtab, h->channels.size() * (size_t)(h->tile_in + T - 1) * sizeof(float2)) != hipSuccess ||
            hipMalloc((void **)&h->d_hist, nh * sizeof(float2)) != hipSuccess ||
            hipMalloc((void **)&h->d_chan, h->channels.size() * sizeof(ChanParams)) != hipSuccess ||
            hipEventCreate(&h->ev0) != hipSuccess || hipEventCreate(&h->ev1) != hipSuccess) { s = LORA_HIP_ERR_NOMEM; break; }
        if (hipMemcpy(h->d_taps, padded.data(), T * sizeof(float), hipMemcpyHostToDevice) != hipSuccess ||
            hipMemset(h->d_hist, 0, nh * sizeof(float2)) != hipSuccess) { s = LORA_HIP_ERR_HIP; break; }
        s = upload_channels(h, true);
    } while (false);
    if (s != LORA_HIP_OK) { lora_hip_channelizer_destroy(h); return s; }
    *out = h;
    return LORA_HIP_OK;
}

void lora_hip_channelizer_destroy(lora_hip_channelizer_t *h)
{
    if (!h) return;
    (void)hipSetDevice(h->device);
    if (h->d_taps) (void)hipFree(h->d_taps);
    if (h->d_wtab) (void)hipFree(h->d_wtab);
    if (h->d_hist) (void)hipFree(h->d_hist);
    if (h->d_chan) (void)hipFree(h->d_chan);
    if (h->d_stage_in) (void)hipFree(h->d_stage_in);
    if (h->d_stage_out) (void)hipFree(h->d_stage_out);
    if (h->ev0) (void)hipEventDestroy(h->ev0);
    if (h->ev1) (void)hipEventDestroy(h->ev1);
    delete h;
}

const char *lora_hip_channelizer_last_error(const lora_hip_channelizer_t *h) { return h ? h->err.c_str() : "null handle"; }

lora_hip_status lora_hip_channelizer_taps(const lora_hip_channelizer_t *h, float *taps, size_t cap, size_t *n)
{
    if (!h || !n) return LORA_HIP_ERR_ARG;
    *n = h->taps.size();
    if (!taps) return LORA_HIP_OK;
    if (cap < h->taps.size()) return LORA_HIP_ERR_OVERFLOW;
    std::memcpy(taps, h->taps.data(), h->taps.size() * sizeof(float));
    return LORA_HIP_OK;
}

size_t lora_hip_channelizer_output_items(const lora_hip_channelizer_t *h, size_t n_in)
{
    if (!h) return 0;
    const long long D = h->cfg.decimation;
    const long long first = (D - (h->n_abs % D)) % D; // outputs sit at absolute input indices that are multiples of D
    return (long long)n_in > first ? (size_t)(((long long)n_in - first + D - 1) / D) : 0;
}

lora_hip_status lora_hip_channelizer_run_device(lora_hip_channelizer_t *h, const void *d_in, size_t n_in, void *d_out,
                                                size_t out_stride, size_t *n_out, void *hip_stream)
{
    if (!h || !n_out || (n_in && (!d_in || !d_out))) return LORA_HIP_ERR_ARG;
    const size_t no = lora_hip_channelizer_output_items(h, n_in);
    *n_out = no;
    if (no > out_stride) return cfail(h, LORA_HIP_ERR_OVERFLOW, "out_stride %zu < %zu output items", out_stride, no);
    hipStream_t st = (hipStream_t)hip_stream;
    CH_TRY(h, hipSetDevice(h->device));
    const int T = (int)h->taps.size(); // real tap count: the history is T - 1 items
    const long long D = h->cfg.decimation;
    h->last_ms = 0.0f;
    if (no) {
        FirArgs a{};
        a.in = (const float2 *)d_in; a.hist = h->d_hist; a.out = (float2 *)d_out; a.taps = h->d_taps; a.wtab = h->d_wtab; a.chan = h->d_chan;
        a.n_abs = h->n_abs; a.n_in = (long long)n_in; a.first = (D - (h->n_abs % D)) % D; a.n_out = (long long)no; a.out_stride = (long long)out_stride;
        a.ntaps = h->ntaps_pad; a.decim = (int)D; a.wtab_stride = h->wtab_stride; a.nhist = T - 1; a.tile_in = h->tile_in;
        CH_TRY(h, hipEventRecord(h->ev0, st));
        if (D == 1) launch_fir<kOutD1, true>(a, (int)h->channels.size(), st);
        else launch_fir<1, false>(a, (int)h->channels.size(), st);
        CH_TRY(h, hipGetLastError());
        CH_TRY(h, hipEventRecord(h->ev1, st));
    }
    // the next call's history: the last ntaps - 1 input items seen so far
    if (n_in >= (size_t)(T - 1)) {
        CH_TRY(h, hipMemcpyAsync(h->d_hist, (const float2 *)d_in + (n_in - (size_t)(T - 1)), (size_t)(T - 1) * sizeof(float2), hipMemcpyDeviceToDevice, st));
    } else if (n_in) {
        const size_t keep = (size_t)(T - 1) - n_in;
        std::vector<float2> tmp((size_t)(T - 1));
        CH_TRY(h, hipStreamSynchronize(st));
        CH_TRY(h, hipMemcpy(tmp.data(), h->d_hist + n_in, keep * sizeof(float2), hipMemcpyDeviceToHost));
        CH_TRY(h, hipMemcpy(tmp.data() + keep, d_in, n_in * sizeof(float2), hipMemcpyDeviceToHost));
        CH_TRY(h, hipMemcpy(h->d_hist, tmp.data(), tmp.size() * sizeof(float2), hipMemcpyHostToDevice));
    }
    CH_TRY(h, hipStreamSynchronize(st));
    if (no) CH_TRY(h, hipEventElapsedTime(&h->last_ms, h->ev0, h->ev1));
    h->n_abs += (long long)n_in;
    return LORA_HIP_OK;
}

lora_hip_status lora_hip_channelizer_work(lora_hip_channelizer_t *h, const float *in, size_t n_in, float *out, size_t out_stride, size_t *n_out)
{
    if (!h || !n_out || (n_in && (!in || !out))) return LORA_HIP_ERR_ARG;
    const size_t no = lora_hip_channelizer_output_items(h, n_in);
    if (no > out_stride) { *n_out = no; return cfail(h, LORA_HIP_ERR_OVERFLOW, "out_stride %zu < %zu output items", out_stride, no); }
    CH_TRY(h, hipSetDevice(h->device));
    const size_t nc = h->channels.size();
    if (n_in > h->stage_in_cap) {
        if (h->d_stage_in) (void)hipFree(h->d_stage_in);
        h->d_stage_in = nullptr; h->stage_in_cap = 0;
        CH_TRY(h, hipMalloc((void **)&h->d_stage_in, (n_in + n_in / 4 + 16) * sizeof(float2)));
        h->stage_in_cap = n_in + n_in / 4 + 16;
    }
    const size_t need_out = nc * std::max<size_t>(no, 1);
    if (need_out > h->stage_out_cap) {
        if (h->d_stage_out) (void)hipFree(h->d_stage_out);
        h->d_stage_out = nullptr; h->stage_out_cap = 0;
        CH_TRY(h, hipMalloc((void **)&h->d_stage_out, (need_out + need_out / 4 + 16) * sizeof(float2)));
        h->stage_out_cap = need_out + need_out / 4 + 16;
    }
    if (n_in) CH_TRY(h, hipMemcpy(h->d_stage_in, in, n_in * sizeof(float2), hipMemcpyHostToDevice));
    lora_hip_status s = lora_hip_channelizer_run_device(h, h->d_stage_in, n_in, h->d_stage_out, std::max<size_t>(no, 1), n_out, nullptr);
    if (s != LORA_HIP_OK) return s;
    for (size_t c = 0; c < nc && no; c++)
        CH_TRY(h, hipMemcpy(out + 2 * c * out_stride, h->d_stage_out + c * std::max<size_t>(no, 1), no * sizeof(float2), hipMemcpyDeviceToHost));
    return LORA_HIP_OK;
}

lora_hip_status lora_hip_channelizer_apply_cfo(lora_hip_channelizer_t *h, float cfo)
{
    if (!h) return LORA_HIP_ERR_ARG;
    CH_TRY(h, hipSetDevice(h->device));
    h->cfo += cfo; // :69
    h->cfo_applied = true;
    return upload_channels(h, false);
}

float lora_hip_channelizer_last_kernel_ms(const lora_hip_channelizer_t *h) { return h ? h->last_ms : 0.0f; }

} // extern "C"
