/* Stand-in for <volk/volk.h>: the VOLK kernels lib/decoder_impl.cc calls, as the plain sequential loops of
 * VOLK's own "generic" implementations (volk/kernels/volk/*.h, *_generic).  A SIMD machine-specific VOLK
 * protokernel sums in a different order; that order is not knowable here - results agree with
 * any VOLK build to float rounding.  TEST INFRASTRUCTURE (oracle/ref_build). */
#ifndef REFSTUB_VOLK_H
#define REFSTUB_VOLK_H
#include <complex>
typedef std::complex<float> lv_32fc_t;

static inline void volk_32f_x2_dot_prod_32f(float* result, const float* input, const float* taps, unsigned int num_points)
{
    float acc = 0.0f;
    for (unsigned int i = 0; i < num_points; i++)
        acc += input[i] * taps[i];
    *result = acc;
}

/* result = sum input[i] * conj(taps[i]) */
static inline void volk_32fc_x2_conjugate_dot_prod_32fc(lv_32fc_t* result, const lv_32fc_t* input, const lv_32fc_t* taps,
                                                        unsigned int num_points)
{
    float re = 0.0f, im = 0.0f;
    for (unsigned int i = 0; i < num_points; i++) {
        const float ar = input[i].real(), ai = input[i].imag();
        const float br = taps[i].real(), bi = taps[i].imag();
        re += ar * br + ai * bi;
        im += ai * br - ar * bi;
    }
    *result = lv_32fc_t(re, im);
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
    float acc = 0.0f;
    for (unsigned int i = 0; i < num_points; i++)
        acc += inputBuffer[i];
    *result = acc;
}

static inline void volk_32fc_x2_multiply_32fc(lv_32fc_t* cVector, const lv_32fc_t* aVector, const lv_32fc_t* bVector,
                                              unsigned int num_points)
{
    for (unsigned int i = 0; i < num_points; i++)
        cVector[i] = aVector[i] * bVector[i];
}
#endif
