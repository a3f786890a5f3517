// lora_strict_sync.inc.hip -- SYNC's near-ties decided the way the reference decides them.
//
// detect_upchirp (decoder_impl.cc:392-413) takes the FIRST maximum of C[i] = sum_k ifreq[i+k] * d_upchirp_ifreq[k], each C[i] a
// sequential float sum (volk_32f_x2_dot_prod_32f, :259-263) over ifreq values that come from two atan2f and an unwrap (:224-244).
// On a preamble C[i0+1] - C[i0] = b * sum(u) ~ 6 / sps^2 of the peak: far below the rounding of either sum, so WHICH of the two
// shifts wins is decided by the float arithmetic itself.  The kernels find the maximum in closed form in double (O(sps)); this file
// adds what makes the decision the reference's: every shift whose closed-form value lies within kTolRel of the maximum is a
// candidate (at most kK; a real signal has one or two), and the candidates are re-evaluated with the reference's own arithmetic -
//   * atan2f as glibc 2.35 computes it (the fdlibm float algorithm, restated below; tests/test_atan2f_restatement.py holds the
//     restatement to libm bit for bit on the host, tests/test_gpu_strict_sync.py the device to the host),
//   * the unwrap with its float difference and double correction (:236-237),
//   * one product and one float add per tap, in tap order, no contraction -
// by ONE lane per candidate.  The products are made by the whole workgroup, a chunk of taps at a time, into LDS; wave 0 adds them up.
// Measured (profiles/r04_strict_sync_*): a dependent v_add_f32 chain runs at 8 shader clocks per tap on one wavefront whatever feeds it
// (8 k clocks per SYNC at SF7, 262 k at SF12), the exact arctangents cost about as much again at SF7: 7-8 % of a pass at every
// spreading factor (6-8 %: the figure include/lora_hip.h quotes as well).  Pinned to ONE arithmetic environment - glibc 2.35's atan2f, VOLK's generic
// sequential dot product (oracle/ref_build) - as the flag's description in include/lora_hip.h says.  LORA_HIP_FLAG_FAST_SYNC skips it (the closed-form maximum stands: one sample beside the reference at SF11 / SF12).
// Round 6, in the SF7 walker (two workgroups per CU; a build with the re-evaluation skipped): of a SYNC round's 56 k clocks the exact arctangents are 7 k and the
// re-evaluation 24 k.  Four other adding chains were built, all exact (tests/test_gpu_strict_sync.py, test_gpu_a16.py, test_golden.py, test_gpu_fullsize.py), none
// faster: a ring of eight LDS batches instead of four (spills), the products in registers fed to two independent chains through v_readlane (two instructions per tap
// and candidate), the running sums walking through the lanes with the taps (v_add_f32 with a DPP row_shr:2 operand), and the whole sum by 64 lanes at once - integer
// increments inside a binade, tests/host_sim/par_sum_model.c - with ~7 VALU instructions per tap and nothing outside the registers.  The time was never in the chain:
// it was in the CALL (resolve_lds below).
#pragma once

namespace strict {

constexpr int kK = 4;                 // candidate shifts re-evaluated at most
constexpr float kTolRel = 4.0e-5f;    // closed form vs sequential float sum: table noise ~1e-5 of the peak + rounding ~sqrt(sps) ulp

struct alignas(16) Cands {
    int32_t n;            // candidates pushed (more than kK: the closed-form winner stands)
    int32_t win;
    float   win_v;
    int32_t pad;
    int32_t idx[kK];
    float   acc[kK];
};

// atanf / atan2f of glibc 2.35 = fdlibm's s_atanf.c / e_atan2f.c (Sun Microsystems 1993, float conversion by Ian Lance Taylor):
// argument reduction at 7/16, 11/16, 19/16, 39/16, an odd polynomial of degree 23 in two interleaved halves.  Every operation is a
// single IEEE float operation in the order of the original; no fused multiply-add may replace a product and a sum.
__device__ __forceinline__ float fd_atanf(float x)
{
#pragma clang fp contract(off)
    const float aT0 = 3.3333334327e-01f, aT1 = -2.0000000298e-01f, aT2 = 1.4285714924e-01f, aT3 = -1.1111110449e-01f, aT4 = 9.0908870101e-02f,
                aT5 = -7.6918758452e-02f, aT6 = 6.6610731184e-02f, aT7 = -5.8335702866e-02f, aT8 = 4.9768779427e-02f, aT9 = -3.6531571299e-02f,
                aT10 = 1.6285819933e-02f;
    const int32_t hx = (int32_t)__float_as_uint(x), ix = hx & 0x7fffffff;
    if (ix >= 0x4c000000) { // |x| >= 2^25
        if (ix > 0x7f800000) return x + x;
        return hx > 0 ? 1.5707962513e+00f + 7.5497894159e-08f : -1.5707962513e+00f - 7.5497894159e-08f;
    }
    if (ix < 0x31000000) return x; // |x| < 2^-29
    // the five ranges of the original (|x| < 7/16: none; < 11/16; < 19/16; < 39/16; above) as ONE division with selected operands: the lanes
    // of a wavefront fall into different ranges, and as branches every lane would pay for all four divisions.  x / 1.0f == x exactly.
    const float ax = fabsf(x);
    const bool r0 = ix < 0x3ee00000, r1 = ix < 0x3f300000, r2 = ix < 0x3f980000, r3 = ix < 0x401c0000;
    const float num = r0 ? x : r1 ? 2.0f * ax - 1.0f : r2 ? ax - 1.0f : r3 ? ax - 1.5f : -1.0f;
    const float den = r0 ? 1.0f : r1 ? 2.0f + ax : r2 ? ax + 1.0f : r3 ? 1.0f + 1.5f * ax : ax;
    const float hi = r1 ? 4.6364760399e-01f : r2 ? 7.8539812565e-01f : r3 ? 9.8279368877e-01f : 1.5707962513e+00f;
    const float lo = r1 ? 5.0121582440e-09f : r2 ? 3.7748947079e-08f : r3 ? 3.4473217170e-08f : 7.5497894159e-08f;
    x = num / den;
    const float z = x * x, w = z * z;
    const float s1 = z * (aT0 + w * (aT2 + w * (aT4 + w * (aT6 + w * (aT8 + w * aT10)))));
    const float s2 = w * (aT1 + w * (aT3 + w * (aT5 + w * (aT7 + w * aT9))));
    if (r0) return x - x * (s1 + s2);
    const float r = hi - ((x * (s1 + s2) - lo) - x);
    return hx < 0 ? -r : r;
}

// the exceptional arguments of e_atan2f.c, out of line: NaN, x == 1, zeros, infinities, |y / x| beyond 2^+-60
__device__ __attribute__((noinline)) float fd_atan2f_special(float y, float x)
{
#pragma clang fp contract(off)
    const float tiny = 1.0e-30f, pi_o_4 = 7.8539818525e-01f, pi_o_2 = 1.5707963705e+00f, pi = 3.1415927410e+00f, pi_lo = -8.7422776573e-08f;
    const int32_t hx = (int32_t)__float_as_uint(x), ix = hx & 0x7fffffff;
    const int32_t hy = (int32_t)__float_as_uint(y), iy = hy & 0x7fffffff;
    if (ix > 0x7f800000 || iy > 0x7f800000) return x + y;
    if (hx == 0x3f800000) return fd_atanf(y);
    const int m = ((hy >> 31) & 1) | ((hx >> 30) & 2);
    if (iy == 0) {
        if (m < 2) return y;
        return m == 2 ? pi + tiny : -pi - tiny;
    }
    if (ix == 0) return hy < 0 ? -pi_o_2 - tiny : pi_o_2 + tiny;
    if (ix == 0x7f800000) {
        if (iy == 0x7f800000) return m == 0 ? pi_o_4 + tiny : m == 1 ? -pi_o_4 - tiny : m == 2 ? 3.0f * pi_o_4 + tiny : -3.0f * pi_o_4 - tiny;
        return m == 0 ? 0.0f : m == 1 ? -0.0f : m == 2 ? pi + tiny : -pi - tiny;
    }
    if (iy == 0x7f800000) return hy < 0 ? -pi_o_2 - tiny : pi_o_2 + tiny;
    const int32_t k = (iy - ix) >> 23;
    float z;
    if (k > 60) z = pi_o_2 + 0.5f * pi_lo;
    else if (hx < 0 && k < -60) z = 0.0f;
    else z = fd_atanf(fabsf(y / x));
    if (m == 0) return z;
    if (m == 1) return -z;
    if (m == 2) return pi - (z - pi_lo);
    return (z - pi_lo) - pi;
}

// e_atan2f.c for ordinary arguments without a branch (a wavefront's lanes fall into every range of s_atanf.c at once): the quotient's
// range picks the operands of ONE second division and the constants; |y / x| >= 2^25 and < 2^-29 are selected at the end.  Every
// operation is the original's, in its order.
__device__ __forceinline__ float fd_atan2f(float y, float x)
{
#pragma clang fp contract(off)
    const float pi = 3.1415927410e+00f, pi_lo = -8.7422776573e-08f;
    const int32_t hx = (int32_t)__float_as_uint(x), ix = hx & 0x7fffffff;
    const int32_t hy = (int32_t)__float_as_uint(y), iy = hy & 0x7fffffff;
    const int32_t k = (iy - ix) >> 23;
    if (__builtin_expect(ix >= 0x7f800000 || iy >= 0x7f800000 || ix == 0 || iy == 0 || hx == 0x3f800000 || k > 60 || k < -60, 0))
        return fd_atan2f_special(y, x);
    const float q = fabsf(y / x);                       // > 0, finite
    const int32_t iq = (int32_t)__float_as_uint(q);
    const float aT0 = 3.3333334327e-01f, aT1 = -2.0000000298e-01f, aT2 = 1.4285714924e-01f, aT3 = -1.1111110449e-01f, aT4 = 9.0908870101e-02f,
                aT5 = -7.6918758452e-02f, aT6 = 6.6610731184e-02f, aT7 = -5.8335702866e-02f, aT8 = 4.9768779427e-02f, aT9 = -3.6531571299e-02f,
                aT10 = 1.6285819933e-02f;
    const bool r0 = iq < 0x3ee00000, r1 = iq < 0x3f300000, r2 = iq < 0x3f980000, r3 = iq < 0x401c0000;
    const float num = r0 ? q : r1 ? 2.0f * q - 1.0f : r2 ? q - 1.0f : r3 ? q - 1.5f : -1.0f;
    const float den = r0 ? 1.0f : r1 ? 2.0f + q : r2 ? q + 1.0f : r3 ? 1.0f + 1.5f * q : q;
    const float hi = r1 ? 4.6364760399e-01f : r2 ? 7.8539812565e-01f : r3 ? 9.8279368877e-01f : 1.5707962513e+00f;
    const float lo = r1 ? 5.0121582440e-09f : r2 ? 3.7748947079e-08f : r3 ? 3.4473217170e-08f : 7.5497894159e-08f;
    const float t = num / den;                          // (q / 1.0f == q exactly)
    const float z2 = t * t, w = z2 * z2;
    const float s1 = z2 * (aT0 + w * (aT2 + w * (aT4 + w * (aT6 + w * (aT8 + w * aT10)))));
    const float s2 = w * (aT1 + w * (aT3 + w * (aT5 + w * (aT7 + w * aT9))));
    float z = r0 ? t - t * (s1 + s2) : hi - ((t * (s1 + s2) - lo) - t);
    z = iq >= 0x4c000000 ? 1.5707962513e+00f + 7.5497894159e-08f : iq < 0x31000000 ? q : z;
    const float zl = z - pi_lo;
    return hx >= 0 ? (hy >= 0 ? z : -z) : (hy >= 0 ? pi - zl : zl - pi);
}

// one value of instantaneous_frequency (:231-240): the float difference is compared against the double M_PI, the correction is made in
// double (2.0f * M_PI is a double) and stored back to float.  For a float d: (double)d > M_PI  <=>  d >= (float)M_PI (0x40490fdb lies just
// above the double pi, its predecessor below), so the test itself needs no double.
__device__ __forceinline__ float ref_ifreq(float p1, float p2)
{
#pragma clang fp contract(off)
    const double two_pi = 6.28318530717958647692;
    const float pif = 3.14159274101257324e+00f;
    while (p2 - p1 >= pif) p2 = (float)((double)p2 - two_pi);
    while (p2 - p1 <= -pif) p2 = (float)((double)p2 + two_pi);
    return p2 - p1;
}

__device__ __forceinline__ void cands_reset(Cands &C) { C.n = 0; }

// this thread's best and second-best closed-form shifts against the workgroup's maximum
__device__ __forceinline__ void cands_push(Cands &C, float gmax, float v1, int i1, float v2, int i2)
{
    const float thr = gmax - kTolRel * gmax;
    if (v1 > 0.0f && v1 >= thr) { const int s = atomicAdd(&C.n, 1); if (s < kK) C.idx[s] = i1; }
    if (v2 > 0.0f && v2 >= thr) { const int s = atomicAdd(&C.n, 1); if (s < kK) C.idx[s] = i2; }
}

// The adding lane (one per candidate): acc += p[0], += p[1], ... over GP groups of four taps laid out [group][NC][4] in LDS, in tap order.
// A ring of four register batches of four ds_read_b128 (16 taps each): three are in flight (~200 clocks of cover for the LDS round trip)
// while the fourth is added up - the loop is the serial floor of the whole re-evaluation, ~5.5 shader clocks per tap (with a wait behind
// every read it was 25, with two batches of eight 8).
template <int NC>
__device__ __forceinline__ float chain_add(const float *src, int lane, int GP, float acc)
{
#pragma clang fp contract(off)
    typedef float v4f __attribute__((ext_vector_type(4))); // (a plain vector type: HIP's float4 has no assignment from an LDS-qualified object)
    typedef __attribute__((address_space(3))) const v4f lds_f4;
    lds_f4 *s4 = (lds_f4 *)(__attribute__((address_space(3))) const float *)src + lane;
    v4f R[4][4];
#pragma unroll
    for (int b = 0; b < 3; b++)
#pragma unroll
        for (int q = 0; q < 4; q++) R[b][q] = s4[(4 * b + q) * NC];
    for (int g = 0; g < GP; g += 16) { // (GP is a multiple of 16: chunks of at least 64 taps)
#pragma unroll
        for (int b = 0; b < 4; b++) {
            // request the batch three ahead (wrapping to the chunk's start at its end: read, never added), THEN add this one up; tying the adds to
            // the barrier keeps the scheduler from sinking the reads to their use or hoisting the adds
            const int gn = g + 4 * b + 12;
            lds_f4 *sn = s4 + (gn < GP ? gn : 0) * NC;
#pragma unroll
            for (int q = 0; q < 4; q++) R[(b + 3) & 3][q] = sn[q * NC];
            asm volatile("" : "+v"(acc) :: "memory");
#pragma unroll
            for (int q = 0; q < 4; q++) { acc = acc + R[b][q].x; acc = acc + R[b][q].y; acc = acc + R[b][q].z; acc = acc + R[b][q].w; }
        }
    }
    return acc;
}

// The reference's own sums for the candidates C.idx[0 .. C.n), all T threads of the workgroup together (C.n in 2 .. kK, uniform; the
// pushes are behind a barrier).  x: the SYNC window (2 sps items), u: d_upchirp_ifreq, buf: LDS, (DB ? 2 : 1) * kK * CH floats,
// 16-byte aligned.  Returns the first maximum's shift (0x7fffffff: no sum exceeds 0, `int i = 0` stands, :771) and its sum.
//
// Taps are processed in chunks of W = CH * kK / ST, ST = 2 or 4 candidate slots (two candidates - the normal case - get twice the chunk
// of four).  The products of a chunk lie in LDS as [group of four taps][slot][4], so that lane c of wave 0 reads its next four products
// with one ds_read_b128.  Production: the chunk needs ifreq[n] for n in [idx_min + k0, idx_max + k0 + W): every producer
// takes a run of R consecutive n (R + 1 atan2f), forms ifreq[n] once and multiplies it into every candidate's tap k = n - idx_c.
// DB: wave 0 adds chunk j up while the other wavefronts produce chunk j + 1.
template <int T, int CH, bool DB>
__device__ __attribute__((noinline)) int resolve(const float2 *__restrict__ x, int sps, const float *__restrict__ u, Cands *Cp, float *buf, float *bv_out)
{
#pragma clang fp contract(off)
    static_assert(CH % 4 == 0 && (CH & (CH - 1)) == 0, "chunks are whole groups of four taps");
    Cands &C = *Cp;
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const int nc = C.n;
    if (t == 0) { // ascending shifts: the reference meets them in this order
        int id[kK];
        for (int c = 0; c < kK; c++) id[c] = c < nc ? C.idx[c] : 0x7fffffff;
        for (int a = 1; a < kK; a++) for (int b = a; b > 0 && id[b - 1] > id[b]; b--) { const int s = id[b]; id[b] = id[b - 1]; id[b - 1] = s; }
        for (int c = 0; c < kK; c++) C.idx[c] = id[c];
    }
    __syncthreads();
    int id[kK];
#pragma unroll
    for (int c = 0; c < kK; c++) id[c] = C.idx[c];
    const int idmin = id[0], spread = id[nc - 1] - id[0];
    const int nsteps = sps - 1;
    const int ST = nc <= 2 ? 2 : 4;               // candidate slots per group of four taps
    // taps per chunk: what the buffer holds; when production overlaps the adding, no more than two ifreq values per producer (a chunk's production - 3 atan2f per thread -
    // then takes about as long as adding the previous chunk up: 6 clocks per tap on ONE lane); whole batches of 64 taps
    // (DB only; without the overlap one large chunk is cheapest)
    const int wbuf = CH * (kK / ST), wfit = ((2 * (T - 64) - spread) / 64) * 64;
    const int W = DB && wfit >= 64 && wfit < wbuf ? wfit : wbuf;
    const int nch = (nsteps + W - 1) / W;
    const int last = 2 * sps - 1;                 // last item of the window
    auto taps_of = [&](int j) { const int wv = nsteps - j * W; return ((wv < W ? wv : W) + 63) & ~63; }; // the last chunk: what is left, in whole batches
    auto produce = [&](int j, float *dst, int tp, int np) {
        const int k0 = j * W;
        const int span = taps_of(j) + spread;     // n - idmin - k0 in [0, span)
        const int R = (span + np - 1) / np;
        const int n0 = idmin + k0 + tp * R, n1 = min(n0 + R, idmin + k0 + span);
        float a_prev = 0.0f;
        for (int nb = n0; nb < n1; nb += 4) {     // four values at a time: their five samples are requested together
            float2 xs[5];
#pragma unroll
            for (int q = 0; q < 5; q++) xs[q] = x[min(nb + q, last)];
            if (nb == n0) a_prev = fd_atan2f(xs[0].y, xs[0].x);
#pragma unroll
            for (int q = 0; q < 4; q++) {
                const int n = nb + q;
                if (n < n1) {
                    const float a_cur = fd_atan2f(xs[q + 1].y, xs[q + 1].x);
                    const float f = ref_ifreq(a_prev, a_cur);
                    a_prev = a_cur;
#pragma unroll
                    for (int c = 0; c < kK; c++) {
                        const int k = n - id[c];  // (id[c] = 0x7fffffff for c >= nc: k < 0)
                        if (c < nc && k >= k0 && k < k0 + span - spread) {
                            const int kk = k - k0;
                            ((__attribute__((address_space(3))) float *)dst)[((kk >> 2) * ST + c) * 4 + (kk & 3)] = k < nsteps ? f * u[k] : 0.0f; // taps past sps-2: +0.0f leaves a float sum as it is
                        }
                    }
                }
            }
        }
    };
    float acc = 0.0f;
    auto chain = [&](const float *src, int gp /* groups of four taps, a multiple of 16 */) {
        if (wave == 0) { // everyone else waits for this ONE wavefront: it goes first on its SIMD (a co-resident workgroup's wavefronts took 2/3 of the issue slots)
            __builtin_amdgcn_s_setprio(3);
            if (lane < nc) acc = ST == 2 ? chain_add<2>(src, lane, gp, acc) : chain_add<4>(src, lane, gp, acc);
            __builtin_amdgcn_s_setprio(0);
        }
    };
    if constexpr (DB) {
        produce(0, buf, t, T);
        __syncthreads();
        for (int j = 0; j < nch; j++) {
            if (j + 1 < nch && wave != 0) produce(j + 1, buf + (size_t)((j + 1) & 1) * kK * CH, t - 64, T - 64);
            chain(buf + (size_t)(j & 1) * kK * CH, taps_of(j) / 4);
            __syncthreads();
        }
    } else {
        for (int j = 0; j < nch; j++) {
            produce(j, buf, t, T);
            __syncthreads();
            chain(buf, taps_of(j) / 4);
            __syncthreads();
        }
    }
    if (wave == 0 && lane < nc) C.acc[lane] = acc;
    __syncthreads();
    float best = 0.0f; // max_correlation = 0 (:400)
    int bi = 0x7fffffff;
    for (int c = 0; c < nc; c++) { const float v = C.acc[c]; if (v > best) { best = v; bi = C.idx[c]; } }
    *bv_out = best;
    return bi;
}

// The same decision when the window's EXACT instantaneous frequency is already in LDS (walker2: SYNC computes it with fd_atan2f /
// ref_ifreq instead of the product form when strict SYNC is on, so that closed form and re-evaluation share the arctangents): f = ifreq of
// the 2 sps window, ul = d_upchirp_ifreq[0 .. sps-1) (= the first sps-1 entries of d_upchirp_ifreq_v), buf: LDS, cap floats.  Chunks of
// W = cap / ST taps: the workgroup multiplies, wave 0 adds.
template <int T>
__device__ __forceinline__ int resolve_lds_inl(const float *f, int sps, const float *ul, Cands *Cp, float *buf, int cap, float *bv_out)
{
#include "lora_strict_resolve_lds.inc"
}

// ... as a call.  Where the caller's registers are worth more than the call: the SF8 walker, whose 128-register build spills 120 more with the body inline (-10 %).  In
// the SF7 walker it is the call that costs - the state machine's registers saved and restored around it, 30 k of a job's 159 k SYNC clocks (round 6: the three faster
// adding chains above were hunting time that was never in the chain) - and the body is inline: 249 -> 261 Gsamples/s.
template <int T>
__device__ __attribute__((noinline)) int resolve_lds(const float *f, int sps, const float *ul, Cands *Cp, float *buf, int cap, float *bv_out)
{
#include "lora_strict_resolve_lds.inc"
}

} // namespace strict
