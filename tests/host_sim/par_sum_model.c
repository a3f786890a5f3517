/* par_sum_model.c -- TEST-ONLY model of an exact parallel form of the strict SYNC's sum (NOT in the kernels: built on the device in round 4, exact, and
 * measured slower than the single adding lane of gr_lora_amd/csrc/lora_strict_sync.inc.hip - profiles/r04_ab_wave_parallel_exact_sum.txt): the sequential float sum
 *     s = fl(s + p[0]); s = fl(s + p[1]); ...                       (detect_upchirp's cross_correlate_ifreq_fast, decoder_impl.cc:259-263)
 * computed by 64 lanes at once, EXACTLY.  While s stays inside one binade [2^e, 2^(e+1)) it is an integer multiple M of u = 2^(e-23), and
 * fl(s + p) = (M + q) u with q = p / u rounded to the nearest integer - independent of M unless p / u lies exactly half way between two
 * integers (then the sum's mantissa is rounded to even: q depends on M's parity).  Integer sums are associative: every lane adds up the q of
 * its own run of taps, a scan over the lanes gives every lane its prefix, and the largest run of taps that neither leaves the binade nor
 * meets a half-way case is taken in one step; the tap that ends the run is added the ordinary way.  Built and checked against the plain loop by
 * tests/test_par_sum_model.py (random and real product sequences).  The model walks the lanes one after the other; a kernel would run them side by
 * side with the same arithmetic. */
#include <math.h>
#include <stdint.h>
#include <string.h>

#define LANES 64

static float seq_step(float s, float p)
{
    volatile float r = s + p; /* one IEEE single addition, round to nearest even */
    return r;
}

float par_sum_seq(const float *p, int n, float s)
{
    for (int k = 0; k < n; k++) s = seq_step(s, p[k]);
    return s;
}

/* one chunk of `taps` products (a multiple of 64), entry value s; *steps counts the parallel steps taken (diagnostics) */
float par_sum_chunk(const float *p, int taps, float s, int *steps, int head)
{
    const int L = taps / LANES;
    int pos = 0;
    for (; pos < head && pos < taps; pos++) s = seq_step(s, p[pos]); /* the first taps of a chain, where s changes binade with almost every tap */
    while (pos < taps) {
        uint32_t bits;
        memcpy(&bits, &s, 4);
        const int ex = (int)((bits >> 23) & 0xffu);
        /* the integer form needs a normal positive s whose ulp is a normal float, too */
        if ((bits >> 31) || ex < 32 || ex > 250) { s = seq_step(s, p[pos]); pos++; continue; }
        const int32_t M = (int32_t)((bits & 0x7fffffu) | 0x800000u); /* s = M * 2^(ex - 150) */
        uint32_t ib = (uint32_t)(127 + 150 - ex) << 23, ub = (uint32_t)(ex - 23) << 23; /* 1 / u and u */
        float inv_u, u;
        memcpy(&inv_u, &ib, 4); memcpy(&u, &ub, 4);
        /* pass 1: per lane the sum of q over its taps >= pos, the extremes of its running sum, its first half-way tap */
        int64_t tot[LANES], lo[LANES], hi[LANES];
        int tie[LANES];
        for (int l = 0; l < LANES; l++) {
            int64_t run = 0, mn = INT64_MAX, mx = INT64_MIN; /* extremes of the running sum BEHIND each of its taps (a lane without taps left never stops the step) */
            int t = taps;
            for (int k = l * L; k < (l + 1) * L; k++) {
                if (k < pos) continue;
                const float x = p[k] * inv_u; /* exact: a power of two (overflow to inf only for |p| > 2^127 u: caught below as a range violation) */
                const float ax = fabsf(x);
                float q = rintf(x);
                if (!(ax < 16777216.0f)) q = x > 0 ? 16777216.0f : -16777216.0f; /* (also NaN: leaves the range at once) */
                if (ax < 8388608.0f && ax - floorf(ax) == 0.5f && t == taps) t = k;
                run += (int64_t)q;
                if (run < mn) mn = run;
                if (run > mx) mx = run;
            }
            tot[l] = run; lo[l] = mn; hi[l] = mx; tie[l] = t;
        }
        /* scan over the lanes; which lane is the first whose running sum leaves [2^23, 2^24) or that holds a half-way tap */
        int64_t off[LANES];
        int64_t acc = 0;
        int stop_lane = LANES;
        for (int l = 0; l < LANES; l++) {
            off[l] = acc;
            /* (below 2^23 + 1 the EXACT sum may already lie in the binade underneath, whose ulp is u / 2: M = 2^23 and p = -0.3 u gives 2^e - u / 2, not 2^e) */
            const int bad = lo[l] != INT64_MAX && ((M + acc + lo[l] < 8388609) || (M + acc + hi[l] > 16777215) || tie[l] < taps);
            if (bad && stop_lane == LANES) stop_lane = l;
            acc += tot[l];
        }
        if (steps) (*steps)++;
        if (stop_lane == LANES) { /* the whole rest of the chunk in one step */
            const int32_t Mn = (int32_t)(M + acc);
            s = (float)Mn * u;
            pos = taps;
            continue;
        }
        /* pass 2, the stopping lane only: the first tap that leaves the binade or is a half-way case */
        int64_t run = M + off[stop_lane];
        int k = stop_lane * L;
        if (k < pos) k = pos;
        int stop = taps; /* (a stopping lane always holds the stopping tap) */
        for (; k < (stop_lane + 1) * L; k++) {
            const float x = p[k] * inv_u;
            const float ax = fabsf(x);
            float q = rintf(x);
            if (!(ax < 16777216.0f)) q = x > 0 ? 16777216.0f : -16777216.0f;
            const int half = ax < 8388608.0f && ax - floorf(ax) == 0.5f;
            const int64_t nx = run + (int64_t)q;
            if (half || nx < 8388609 || nx > 16777215) { stop = k; break; }
            run = nx;
        }
        s = (float)(int32_t)run * u; /* the taps before the stopping one */
        if (stop < taps) { s = seq_step(s, p[stop]); pos = stop + 1; } else pos = taps;
    }
    return s;
}

/* a whole chain in chunks of W taps (the last one padded with +0.0f to a multiple of 64, as the kernel's buffers are) */
float par_sum(const float *p, int n, int W, int *steps)
{
    float s = 0.0f;
    float buf[4096];
    for (int k0 = 0; k0 < n; k0 += W) {
        const int wv = n - k0, taps = ((wv < W ? wv : W) + 63) & ~63;
        for (int i = 0; i < taps; i++) buf[i] = (k0 + i < n) ? p[k0 + i] : 0.0f;
        s = par_sum_chunk(buf, taps, s, steps, k0 == 0 ? 32 : 0);
    }
    return s;
}
