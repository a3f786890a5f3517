/* Stand-in for <volk/volk.h>: the VOLK kernels lib/decoder_impl.cc calls, as the plain sequential loops of
 * VOLK's own "generic" implementations (volk/kernels/volk/*.h, *_generic).  A SIMD machine-specific VOLK
 * protokernel sums in a different order; that order is not knowable here - results agree with
 * any VOLK build to float rounding.  TEST INFRASTRUCTURE (oracle/ref_build). */
#ifndef REFSTUB_VOLK_H
#define REFSTUB_VOLK_H
#include <complex>
typedef std::complex<float> lv_32fc_t;

/* REFSTUB_VOLK_LANES = L > 1 builds the SECOND variant of the reference (oracle/_ref/libref_decoder_simd.so): the
 * reductions keep L partial sums (element i goes to lane i mod L, the lanes are added up at the end) the way VOLK's
 * SIMD protokernels (_a_sse / _a_avx, L = 4 / 8 floats per register) do.  Same mathematics, another float summation
 * order: tests/test_ref_pin.py uses the pair to MEASURE how far the reference's own decisions depend on which VOLK
 * protokernel the machine selects. */
#ifndef REFSTUB_VOLK_LANES
#define REFSTUB_VOLK_LANES 1
#endif

static inline void volk_32f_x2_dot_prod_32f(float* result, const float* input, const float* taps, unsigned int num_points)
{
    float acc[REFSTUB_VOLK_LANES] = { 0.0f };
    for (unsigned int i = 0; i < num_points; i++)
        acc[i % REFSTUB_VOLK_LANES] += input[i] * taps[i];
    float r = acc[0];
    for (int l = 1; l < REFSTUB_VOLK_LANES; l++)
        r += acc[l];
    *result = r;
}

/* result = sum input[i] * conj(taps[i]) */
static inline void volk_32fc_x2_conjugate_dot_prod_32fc(lv_32fc_t* result, const lv_32fc_t* input, const lv_32fc_t* taps,
                                                        unsigned int num_points)
{
    float re[REFSTUB_VOLK_LANES] = { 0.0f }, im[REFSTUB_VOLK_LANES] = { 0.0f };
    for (unsigned int i = 0; i < num_points; i++) {
        const float ar = input[i].real(), ai = input[i].imag();
        const float br = taps[i].real(), bi = taps[i].imag();
        re[i % REFSTUB_VOLK_LANES] += ar * br + ai * bi;
        im[i % REFSTUB_VOLK_LANES] += ai * br - ar * bi;
    }
    float sr = re[0], si = im[0];
    for (int l = 1; l < REFSTUB_VOLK_LANES; l++) {
        sr += re[l];
        si += im[l];
    }
    *result = lv_32fc_t(sr, si);
}

static inline void volk_32fc_magnitude_squared_32f(float* magnitudeVector, const lv_32fc_t* complexVector, unsigned int num_points)
{
    for (unsigned int i = 0; i < num_points; i++) {
        const float r = complexVector[i].real(), q = complexVector[i].imag();
        magnitudeVector[i] = r * r + q * q;
    }
}

static inline void volk_32f_accumulator_s32f(float* result, const float* inputBuffer, unsigned int num_points)
{
    float acc[REFSTUB_VOLK_LANES] = { 0.0f };
    for (unsigned int i = 0; i < num_points; i++)
        acc[i % REFSTUB_VOLK_LANES] += inputBuffer[i];
    float r = acc[0];
    for (int l = 1; l < REFSTUB_VOLK_LANES; l++)
        r += acc[l];
    *result = r;
}

static inline void volk_32fc_x2_multiply_32fc(lv_32fc_t* cVector, const lv_32fc_t* aVector, const lv_32fc_t* bVector,
                                              unsigned int num_points)
{
    for (unsigned int i = 0; i < num_points; i++)
        cVector[i] = aVector[i] * bVector[i];
}
#endif
