/* Stand-in for <liquid/liquid.h> (liquid-dsp is not vendored by the reference and not installed here;
 * the reference clones its master branch, .travis.yml:39-45).  The two pieces lib/decoder_impl.cc uses:
 *
 *  - fft_create_plan / fft_execute / fft_destroy_plan (:112-113,:443,:459,:136-137): an unnormalised DFT,
 *    forward = exp(-j 2 pi k n / N), backward = exp(+j ...), like liquid's.  Evaluated here in DOUBLE
 *    (recursive radix-2 for powers of two, direct O(N^2) otherwise) and rounded once to float: deliberately NOT the
 *    oracle's float radix-2, so that agreement of the two is a statement about the DFT, not about one
 *    implementation's rounding.  liquid's own float FFT differs from either by float rounding.
 *  - fec_create(LIQUID_FEC_HAMMING84) / fec_decode / fec_destroy (:116-117,:661,:138): liquid's Hamming(8,4)
 *    decoder is a 256-entry table: out[i] = dec[in[2i]] << 4 | dec[in[2i+1]].  Its code is the one the
 *    reference itself pins in hamming_encode_soft (include/lora/utilities.h:257-264: 00 d2 55 87 99 4b cc 1e
 *    e1 33 b4 66 78 aa 2d ff).  Codewords with <= 1 bit error decode uniquely; for >= 2 bit errors
 *    liquid's table entry is NOT knowable here (PARITY UNPINNED for those inputs): nearest codeword, lowest
 *    symbol on ties, is assumed - the same assumption oracle/lora_oracle.c states.
 * TEST INFRASTRUCTURE (oracle/ref_build). */
#ifndef REFSTUB_LIQUID_H
#define REFSTUB_LIQUID_H
#include <complex>
#include <cstdlib>
#include <vector>

#define LIQUID_FFT_FORWARD (+1)
#define LIQUID_FFT_BACKWARD (-1)

struct refstub_fftplan_s {
    unsigned int n;
    std::complex<float>* x;
    std::complex<float>* y;
    int dir;
    std::vector<std::complex<double>> w; /* exp(-+j 2 pi k / n) */
    std::vector<std::complex<double>> a, b;
};
typedef refstub_fftplan_s* fftplan;

static inline fftplan fft_create_plan(unsigned int n, std::complex<float>* x, std::complex<float>* y, int dir, int /*flags*/)
{
    fftplan p = new refstub_fftplan_s;
    p->n = n;
    p->x = x;
    p->y = y;
    p->dir = dir;
    p->w.resize(n);
    const double sgn = (dir == LIQUID_FFT_FORWARD) ? -1.0 : 1.0;
    for (unsigned int k = 0; k < n; k++) {
        const double ang = sgn * 2.0 * M_PI * (double)k / (double)n;
        p->w[k] = std::complex<double>(cos(ang), sin(ang));
    }
    p->a.resize(n);
    p->b.resize(n);
    return p;
}

/* out[k*1] = DFT of in[0], in[stride], ... (n points); twiddle for size n is w[k * (N/n)] */
static inline void refstub_fft_rec(const refstub_fftplan_s* p, const std::complex<double>* in, std::complex<double>* out,
                                   unsigned int n, unsigned int stride)
{
    if (n == 1) {
        out[0] = in[0];
        return;
    }
    const unsigned int h = n / 2;
    refstub_fft_rec(p, in, out, h, stride * 2);
    refstub_fft_rec(p, in + stride, out + h, h, stride * 2);
    const unsigned int tw = p->n / n;
    for (unsigned int k = 0; k < h; k++) {
        const std::complex<double> e = out[k], o = out[k + h] * p->w[k * tw];
        out[k] = e + o;
        out[k + h] = e - o;
    }
}

static inline void fft_execute(fftplan p)
{
    const unsigned int n = p->n;
    for (unsigned int i = 0; i < n; i++)
        p->a[i] = std::complex<double>(p->x[i].real(), p->x[i].imag());
    if ((n & (n - 1)) == 0) {
        refstub_fft_rec(p, p->a.data(), p->b.data(), n, 1);
    } else {
        for (unsigned int k = 0; k < n; k++) {
            std::complex<double> acc(0.0, 0.0);
            for (unsigned int i = 0; i < n; i++)
                acc += p->a[i] * p->w[(unsigned long long)k * i % n];
            p->b[k] = acc;
        }
    }
    for (unsigned int i = 0; i < n; i++)
        p->y[i] = std::complex<float>((float)p->b[i].real(), (float)p->b[i].imag());
}

static inline void fft_destroy_plan(fftplan p) { delete p; }

/* ---- FEC ---- */
typedef enum { LIQUID_FEC_UNKNOWN = 0, LIQUID_FEC_NONE, LIQUID_FEC_REP3, LIQUID_FEC_REP5, LIQUID_FEC_HAMMING74, LIQUID_FEC_HAMMING84,
               LIQUID_FEC_HAMMING128 } fec_scheme;

struct refstub_fec_s {
    fec_scheme scheme;
    unsigned char dec[256];
};
typedef refstub_fec_s* fec;

static inline fec fec_create(fec_scheme scheme, void* /*opts*/)
{
    /* liquid's hamming84_enc_gentab == the reference's hamming_encode_soft table (utilities.h:257-264) */
    static const unsigned char enc[16] = { 0x00, 0xd2, 0x55, 0x87, 0x99, 0x4b, 0xcc, 0x1e,
                                           0xe1, 0x33, 0xb4, 0x66, 0x78, 0xaa, 0x2d, 0xff };
    if (scheme != LIQUID_FEC_HAMMING84)
        abort();
    fec q = new refstub_fec_s;
    q->scheme = scheme;
    for (int r = 0; r < 256; r++) {
        int best = 0, bestd = 9;
        for (int s = 0; s < 16; s++) {
            const int d = __builtin_popcount((unsigned)(r ^ enc[s]));
            if (d < bestd) {
                bestd = d;
                best = s;
            }
        }
        q->dec[r] = (unsigned char)best;
    }
    return q;
}

static inline void fec_decode(fec q, unsigned int dec_msg_len, unsigned char* msg_enc, unsigned char* msg_dec)
{
    for (unsigned int i = 0; i < dec_msg_len; i++)
        msg_dec[i] = (unsigned char)((q->dec[msg_enc[2 * i]] << 4) | q->dec[msg_enc[2 * i + 1]]);
}

static inline void fec_destroy(fec q) { delete q; }
#endif
