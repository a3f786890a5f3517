/*
 * lora_hip_channelizer.h -- C ABI of the MI355X channeliser: the stage in front of the decoder.
 *
 * Replaces gr::lora::channelizer (reference: include/lora/channelizer.h:40-50, lib/channelizer_impl.cc:46-71):
 *   freq_xlating_fir_filter_ccf(decimation, firdes::low_pass(1, fs, bw/2 + 15 kHz, 10 kHz, WIN_HAMMING),
 *                               channel_list[0] - center_freq, fs)          (channelizer_impl.cc:46-48)
 *   apply_cfo(cfo): d_cfo += cfo; set_center_freq(d_freq_offset + d_cfo)   (:68-71)
 * i.e.   y[m] = sum_k h[k] * x[m*D - k] * exp(-j 2 pi f (m*D - k) / fs),   x[n < 0] = 0 (filter history starts at zero).
 * The reference translates channel_list[0] only; this library translates every listed channel in one pass
 * (output c is the reference's output for channel_list = {channel c}).
 * Plain C types only; device pointers and the HIP stream travel as void*.  Same conventions as lora_hip.h.
 */
#ifndef LORA_HIP_CHANNELIZER_H
#define LORA_HIP_CHANNELIZER_H

#include <stddef.h>
#include <stdint.h>

#include "lora_hip.h"

#ifdef __cplusplus
extern "C" {
#endif

/* Arguments of gr::lora::channelizer::make (include/lora/channelizer.h:47; channelizer_impl.cc:31-34). */
typedef struct lora_hip_channelizer_config {
    uint32_t struct_size;
    float    samp_rate;      /* arg 1 */
    float    center_freq;    /* arg 2 */
    const float *channel_list; /* arg 3: n_channels centre frequencies (Hz) */
    uint32_t n_channels;
    uint32_t bandwidth;      /* arg 4 */
    uint32_t decimation;     /* arg 5 (>= 1) */
    int32_t  device;         /* HIP device ordinal */
    /* Filter design overrides, 0 = the channeliser's own (cutoff bandwidth/2 + 15 kHz, transition 10 kHz, :46).  The
     * reference's test harness runs a second freq_xlating_fir_filter with low_pass(1, fs, 200 kHz, 100 kHz) in front of
     * the receiver (python/qa_testsuite.py:233); apps/qa_testsuite.py builds that one through these fields.           */
    float    cutoff_hz;
    float    transition_hz;
    uint32_t flags;          /* LORA_HIP_CHANNELIZER_FLAG_* (callers built against the older, shorter struct: 0) */
} lora_hip_channelizer_config_t;

/* Upstream keeps d_freq_offset = channel_list[0] - center_freq in a uint32_t (lib/channelizer_impl.h:39, channelizer_impl.cc:47): whole Hz,
 * and a NEGATIVE offset wraps - the x86-64 conversion of -100032.0f yields 4294867264, the filter translates by that many Hz (an alias:
 * +867264 Hz at 1 Msps, not -100032) and apply_cfo adds the CFO to it in float (:70).  By default this library keeps the sign, which is
 * what the block is meant to do; with this flag it reproduces upstream's arithmetic bit for bit (for A/B runs against an upstream flowgraph). */
#define LORA_HIP_CHANNELIZER_FLAG_UINT32_OFFSET 1u

typedef struct lora_hip_channelizer lora_hip_channelizer_t;

lora_hip_status lora_hip_channelizer_create(const lora_hip_channelizer_config_t *cfg, lora_hip_channelizer_t **out);
void            lora_hip_channelizer_destroy(lora_hip_channelizer_t *h);
const char     *lora_hip_channelizer_last_error(const lora_hip_channelizer_t *h);

/* The low-pass prototype d_lpf (firdes::low_pass, channelizer_impl.cc:46): *n receives the tap count; taps may be
 * NULL to query it.                                                                                             */
lora_hip_status lora_hip_channelizer_taps(const lora_hip_channelizer_t *h, float *taps, size_t cap, size_t *n);

/* Output items the next call will produce for n_in input items (depends on the decimation phase carried over). */
size_t          lora_hip_channelizer_output_items(const lora_hip_channelizer_t *h, size_t n_in);

/* Streaming, device-resident: d_in = n_in cf32 items continuing the input stream; d_out receives n_channels rows
 * of out_stride cf32 items, row c = channel c, *n_out items valid per row.  Filter history, oscillator phase and
 * decimation phase carry over between calls (arbitrary chunking gives the same output stream).  Synchronous on
 * return.                                                                                                       */
lora_hip_status lora_hip_channelizer_run_device(lora_hip_channelizer_t *h, const void *d_in, size_t n_in, void *d_out,
                                                size_t out_stride, size_t *n_out, void *hip_stream);

/* Same with host buffers (the block's work() as a GNU Radio shim calls it): in = n_in cf32, out = n_channels rows. */
lora_hip_status lora_hip_channelizer_work(lora_hip_channelizer_t *h, const float *in, size_t n_in, float *out,
                                          size_t out_stride, size_t *n_out);

/* channelizer_impl::apply_cfo (:68-71): shifts every channel's translation frequency by cfo Hz from now on. */
lora_hip_status lora_hip_channelizer_apply_cfo(lora_hip_channelizer_t *h, float cfo);

/* Kernel time of the last run (HIP events on the launch stream), for the measurements in DESIGN.md. */
float           lora_hip_channelizer_last_kernel_ms(const lora_hip_channelizer_t *h);

#ifdef __cplusplus
}
#endif
#endif /* LORA_HIP_CHANNELIZER_H */
