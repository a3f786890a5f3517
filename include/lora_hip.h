/*
 * lora_hip.h -- C ABI of the MI355X-native LoRa PHY decoder (liblora_hip.so).
 *
 * This is the drop-in boundary for the hot path of rpp0/gr-lora's
 * gr::lora::decoder (reference: include/lora/decoder.h:693-709,
 * lib/decoder_impl.cc).  Plain C types only: no GNU Radio, PMT, torch or HIP
 * types cross it (device pointers and the HIP stream travel as void*).
 * Every entry point names the reference interface it replaces.
 *
 * Error convention: the reference never returns errors -- fatal conditions
 * exit(1) (decoder_impl.cc:57-61, 541-545, 602-605).  Here every call returns
 * a lora_hip_status; a GNU Radio shim maps the fatal ones to the reference's
 * exit(1).  One handle == one decoder instance == one
 * caller thread at a time, exactly like one GNU Radio block; distinct handles
 * are independent.
 */
#ifndef LORA_HIP_H
#define LORA_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define LORA_HIP_ABI_VERSION 4   /* 4: lora_hip_get_table; 3: LORA_HIP_FLAG_FAST_SYNC (strict SYNC is the default), lora_hip_ref_ifreq_device; 2: lora_hip_set_stream_latency, lora_hip_stream_info, lora_hip_walker_kernel_name, lora_hip_window_stats_device, lora_hip_detect_preambles_device, lora_hip_decode_at_headers_device, lora_hip_mux_* */

typedef enum lora_hip_status {
    LORA_HIP_OK = 0,
    LORA_HIP_ERR_BAD_SF = -1,      /* sf < 6 || sf > 12: reference prints and exit(1)s (decoder_impl.cc:57-61) */
    LORA_HIP_ERR_BAD_CONFIG = -2,  /* samples/symbol not a power-of-two multiple of 2^sf, cr > 4, ... */
    LORA_HIP_ERR_NO_DEVICE = -3,   /* no HIP device / device id out of range -- there is NO CPU fallback */
    LORA_HIP_ERR_HIP = -4,         /* a HIP runtime call failed; see lora_hip_last_error() */
    LORA_HIP_ERR_NOMEM = -5,
    LORA_HIP_ERR_ARG = -6,         /* NULL / out-of-range argument */
    LORA_HIP_ERR_OVERFLOW = -7,    /* caller buffer too small */
    LORA_HIP_ERR_INTERNAL = -8
} lora_hip_status;

/* Demodulator used inside demodulate() (decoder_impl.cc:499-500). */
typedef enum lora_hip_demod {
    LORA_HIP_DEMOD_GRAD = 0,        /* max_frequency_gradient_idx (:466-491), the reference's shipped default */
    LORA_HIP_DEMOD_FFT = 1,         /* get_shift_fft (:430-464): dechirp x FFT x argmax; bin_idx = (s-1) mod N   */
    LORA_HIP_DEMOD_FFT_COMPAT = 2   /* FFT, but s == 0 -> bin_idx 0: byte-identical to the default path's quirk */
} lora_hip_demod;

#define LORA_HIP_FLAG_TRACE 0x1u    /* record one lora_hip_step_t per state-machine step (tests / debugging) */
#define LORA_HIP_FLAG_FAST_SYNC 4u /* SYNC (detect_upchirp, decoder_impl.cc:392-413): keep the closed-form maximum as it is.  Without this flag every shift
                                      within rounding of that maximum is re-evaluated with the reference's own arithmetic (glibc's atan2f, the unwrap of
                                      :231-240, one sequential float sum per shift as volk_32f_x2_dot_prod_32f_generic adds) and the FIRST maximum of those
                                      sums wins, as in :399-407 - on a clean preamble two adjacent shifts tie to ~6 / sps^2 of the peak and the float
                                      arithmetic alone decides.  Costs 6-8 % of a pass at every spreading factor (measured: profiles/r05_default_fast_sync_bench_line.json,
                                      profiles/r05_ab_acquisition_experiments.txt); FFT demodulators publish the same bytes either way.
                                      WHAT "the reference's own arithmetic" IS PINNED TO: atan2f as glibc 2.35 computes it (fdlibm's float algorithm) and VOLK's
                                      GENERIC dot product (one sequential float sum) - the build oracle/ref_build compiles and tests/test_ref_pin.py holds.  A
                                      GNU Radio install whose VOLK dispatches a SIMD kernel (several partial sums) or whose libm rounds atan2f differently
                                      decides these ties its own way; no reference build agrees with another one there (tests/test_ref_pin.py::
                                      test_sync_shift_depends_on_volk_summation_order). */
#define LORA_HIP_FLAG_NO_DECOUPLED 8u /* never run a pass decoupled.  A decoupled pass (chosen per pass when its jobs would leave most CUs idle - a gateway's short
                                       * pass, a few packets per channel; SF7-12 at decimation 8, explicit header): the state-machine jobs stop behind every header
                                       * and skip the payload, all payload symbols are demodulated at once at their zero-drift positions, and a packet whose symbols
                                       * moved the symbol clock is decoded again by the complete kernels.  Same frames either way (decoder_impl.cc:838-886).
                                       * LORA_HIP_DECOUPLED=0|1 in the environment overrides the per-pass choice (1: every pass the kernels allow). */
#define LORA_HIP_FLAG_PIN_HOST 2u  /* lora_hip_work may page-lock (hipHostRegister) the caller's buffers to DMA straight from them;
                                      only for long-lived buffers the caller uses for nothing else.  Memory that is already
                                      page-locked (hipHostMalloc / registered by the caller) is always used directly. */

/* Constructor arguments of gr::lora::decoder::make (include/lora/decoder.h:705;
 * python/bindings/decoder_python.cc:36-66), plus the device-side knobs.        */
typedef struct lora_hip_config {
    uint32_t struct_size;            /* sizeof(lora_hip_config_t), for ABI growth                           */
    float    samp_rate;              /* decoder::make arg 1                                                  */
    uint32_t bandwidth;              /* arg 2                                                                */
    uint8_t  sf;                     /* arg 3                                                                */
    uint8_t  implicit;               /* arg 4                                                                */
    uint8_t  cr;                     /* arg 5: initial d_phdr.cr (also decides the first header's FEC, :655) */
    uint8_t  crc;                    /* arg 6                                                                */
    uint8_t  reduced_rate;           /* arg 7                                                                */
    uint8_t  disable_drift_correction; /* arg 8                                                              */
    uint8_t  reserved0[2];
    int32_t  device;                 /* HIP device ordinal                                                   */
    int32_t  demod;                  /* lora_hip_demod                                                       */
    uint32_t flags;                  /* LORA_HIP_FLAG_*                                                      */
    uint32_t segment_symbols;        /* speculation segment length in symbols for long streams; 0 = auto     */
    uint32_t batch_items;            /* streaming: items buffered before a device pass; 0 = auto             */
} lora_hip_config_t;

/* Where a published frame came from. */
typedef struct lora_hip_frame_info {
    uint32_t stream;                 /* index into the stream list of the producing call (0 for lora_hip_work) */
    uint32_t length;                 /* blob length in bytes                                                   */
    int64_t  header_pos;             /* sample index (within its stream) of the first header symbol           */
    int64_t  end_pos;                /* sample index just after the last consumed payload symbol              */
} lora_hip_frame_info_t;

/* One state-machine step (one reference work() call), for position-exact tests. */
typedef struct lora_hip_step {
    int32_t state;                   /* DecoderState on entry (lib/decoder_impl.h:40-48 order)               */
    int32_t consumed;                /* consume_each() amount                                                 */
    int64_t pos;                     /* stream sample index of input[0]                                       */
    int32_t bin;                     /* bin_idx before rate reduction, -1 if none                             */
    int32_t fine;                    /* d_fine_sync after the step                                            */
    float   value;                   /* autocorr / sync corr / SFD corr                                       */
    uint32_t stream;
    uint32_t cycles;                 /* shader clocks the step took on the device                            */
    uint32_t reserved;
} lora_hip_step_t;

/* Device timing of the last lora_hip_decode_device()/lora_hip_flush() pass,
 * measured with HIP events on the launch stream.                                                               */
typedef struct lora_hip_timing {
    float    walker_ms;              /* sum over the walker (decoder state-machine) kernel launches          */
    float    total_device_ms;        /* = walker_ms (the walker launches, HIP events around each; the envelope pre-pass runs beside the previous pass and is not counted) */
    uint32_t walker_launches;
    uint32_t jobs;                   /* workgroups launched in the main pass                                  */
    uint32_t probes;                 /* stitch probes launched                                                */
    uint32_t slow_path_relaunches;   /* serial fix-ups after a failed speculation                             */
    uint64_t items;                  /* IQ items covered                                                      */
} lora_hip_timing_t;

typedef struct lora_hip_decoder lora_hip_decoder_t;

/* ---- lifetime: decoder::make / ~decoder_impl (decoder_impl.cc:41-44, 49-139) ---------------------------- */
lora_hip_status lora_hip_create(const lora_hip_config_t *cfg, lora_hip_decoder_t **out);
void            lora_hip_destroy(lora_hip_decoder_t *h);
const char     *lora_hip_strerror(lora_hip_status s);
const char     *lora_hip_last_error(const lora_hip_decoder_t *h);
uint32_t        lora_hip_abi_version(void);

/* derived rates the constructor prints (decoder_impl.cc:83-87, 93-96) */
lora_hip_status lora_hip_get_geometry(const lora_hip_decoder_t *h, uint32_t *samples_per_symbol,
                                      uint32_t *bins, uint32_t *decimation);

/* warn-only no-ops in the reference (decoder_impl.cc:905-915); return OK and change nothing */
lora_hip_status lora_hip_set_sf(lora_hip_decoder_t *h, uint8_t sf);
lora_hip_status lora_hip_set_samp_rate(lora_hip_decoder_t *h, float samp_rate);

/* ---- streaming: replaces decoder_impl::work() (decoder_impl.cc:740-903) --------------------------------- */
/* Accepts n_items host cf32 items (interleaved re,im) in arbitrary chunking; all of them are consumed
 * (buffered internally).  Frames become available through lora_hip_poll_frame in stream order and are
 * identical to what symbol-at-a-time consumption publishes on the "frames" port (:607-608).               */
lora_hip_status lora_hip_work(lora_hip_decoder_t *h, const float *iq, size_t n_items, size_t *consumed);
/* Runs the device pass over everything buffered, honouring the scheduler rule that work() is only called
 * while 2*samples_per_symbol items remain (set_output_multiple, :91).                                      */
lora_hip_status lora_hip_flush(lora_hip_decoder_t *h);

/* Latency of the streaming path.  The reference publishes a frame inside the work() call that completes the packet
 * (decoder_impl.cc:870-881).  Here samples are decoded in device passes: a pass is launched when batch_items have arrived
 * OR when the oldest sample not yet handed to a pass has waited max_latency_ms of wall-clock time (and at least two symbols
 * are buffered, :91), and a finished pass is collected - its frames published - by the next lora_hip_work call that finds its
 * kernel done (hipEventQuery: no waiting).  A frame therefore surfaces within max_latency_ms + one device pass (< 1 ms for a
 * latency-bounded pass) + the caller's own call period after its last sample arrived; a caller that delivers samples faster
 * than batch_items per max_latency_ms never sees the bound (full chunks, full throughput).  Default 50 ms; 0 = off (passes
 * only on full chunks and on lora_hip_flush).  What is decoded does not depend on where the passes fall.                   */
lora_hip_status lora_hip_set_stream_latency(lora_hip_decoder_t *h, float max_latency_ms);

typedef struct lora_hip_stream_info {
    uint64_t batch_items;            /* effective chunk size (lora_hip_config_t.batch_items, or the automatic value)          */
    uint64_t buffered_items;         /* delivered to lora_hip_work and not yet handed to a pass (tail + chunk being filled)     */
    uint64_t passes;                 /* device passes launched so far                                                          */
    uint64_t passes_by_latency;      /* ... of which because of the latency bound, not a full chunk or a flush                 */
    int64_t  consumed_base;          /* absolute item index up to which the stream is decoded (frames before it are published) */
    float    max_latency_ms;
    uint32_t pass_in_flight;
} lora_hip_stream_info_t;
lora_hip_status lora_hip_stream_info(const lora_hip_decoder_t *h, lora_hip_stream_info_t *out);

/* ---- streaming, many channels through ONE decoder: the gateway flowgraph ---------------------------------------------
 * The reference decodes one channel per block (README.md:13, channelizer_impl.cc:47): a 64-channel gateway is 64 decoder
 * blocks.  As 64 lora_hip_work handles that is 64 near-empty device passes per chunk; a mux feeds all channels' chunks to ONE
 * pass (lora_hip_decode_device over n_channels streams).  Every channel is an independent gr::lora::decoder instance
 * (constructor arguments from cfg; own d_phdr.cr, power queue, position), fed by lora_hip_mux_work(channel, ...) in any order
 * and chunking; a pass over every channel's buffered samples is launched when ALL channels hold batch_items (cfg) or when
 * the oldest unlaunched sample has waited the latency bound (lora_hip_mux_set_latency, default 50 ms) and is collected by
 * the next call that finds it finished.  A channel may run ahead of the slowest one (the surplus waits in host memory) by up
 * to lora_hip_mux_set_max_ahead items (default: 8 chunks, at least 4 Mi items); beyond that the pass goes with what the
 * others hold - a silent or stalled channel cannot let the surplus grow without bound when the latency bound is off.
 * Frames come out of one queue, info.stream = channel; per channel they are what that channel's own
 * lora_hip_work handle would publish.  One mux = one caller thread at a time.                                            */
typedef struct lora_hip_mux lora_hip_mux_t;
lora_hip_status lora_hip_mux_create(const lora_hip_config_t *cfg, uint32_t n_channels, lora_hip_mux_t **out);
void            lora_hip_mux_destroy(lora_hip_mux_t *m);
lora_hip_status lora_hip_mux_work(lora_hip_mux_t *m, uint32_t channel, const float *iq, size_t n_items);
lora_hip_status lora_hip_mux_flush(lora_hip_mux_t *m);
lora_hip_status lora_hip_mux_set_latency(lora_hip_mux_t *m, float max_latency_ms);
lora_hip_status lora_hip_mux_set_max_ahead(lora_hip_mux_t *m, size_t max_ahead_items);   /* 0: the default */
size_t          lora_hip_mux_frames_available(const lora_hip_mux_t *m);
lora_hip_status lora_hip_mux_poll_frame(lora_hip_mux_t *m, uint8_t *buf, size_t cap, size_t *len, lora_hip_frame_info_t *info);
lora_hip_status lora_hip_mux_passes(const lora_hip_mux_t *m, uint64_t *passes, uint64_t *passes_by_latency);
const char     *lora_hip_mux_last_error(const lora_hip_mux_t *m);

/* ---- batched, device-resident: many independent streams in one pass ------------------------------------- */
/* d_iq: device pointer to cf32 items; stream i occupies items [stream_off[i], stream_off[i]+stream_len[i]).
 * Each stream is decoded as by a fresh reference decoder instance constructed with this handle's config.
 * hip_stream: hipStream_t to launch on (NULL = default stream).  Synchronous on return.                    */
lora_hip_status lora_hip_decode_device(lora_hip_decoder_t *h, const void *d_iq, size_t total_items,
                                       const uint64_t *stream_off, const uint64_t *stream_len,
                                       uint32_t n_streams, void *hip_stream);

/* The same pass in two halves, so that a caller can overlap passes: _begin plans the pass and launches its main kernel on
 * hip_stream, then returns; _end waits for that kernel (an event, not the stream), runs the rare follow-up launches, stitches
 * and queues the frames.  One pass per handle at a time; with two handles alternating on ONE stream the device runs
 * pass k+1's kernel right behind pass k's while the host stitches pass k and plans pass k+2 (bench.py does this).
 * lora_hip_decode_device(...) == _begin(..., 0) + _end().  flags: LORA_HIP_BEGIN_IQ_READY = the IQ is complete in memory
 * now, whatever hip_stream still has queued (e.g. the previous pass's kernel): the planner's small envelope pre-pass, which
 * runs on a stream of the handle's own, then does not wait for hip_stream.  Without it the pre-pass is ordered after
 * everything queued on hip_stream so far.                                                                          */
#define LORA_HIP_BEGIN_IQ_READY 1u
lora_hip_status lora_hip_decode_device_begin(lora_hip_decoder_t *h, const void *d_iq, size_t total_items,
                                             const uint64_t *stream_off, const uint64_t *stream_len,
                                             uint32_t n_streams, void *hip_stream, uint32_t flags);
lora_hip_status lora_hip_decode_device_end(lora_hip_decoder_t *h);
/* Optional third stage, one pass further ahead: issues the envelope pre-pass of the pass that will be begun NEXT on this
 * handle (same d_iq and streams; anything else is simply issued again by _begin) and returns without waiting.  Issued
 * while an earlier pass's kernel occupies the device, it runs in that kernel's tail, and _begin finds the gap list ready
 * (bench.py --depth 3: prepass(k+2), begin(k+1), end(k) with three handles).  Same flags as _begin.                     */
lora_hip_status lora_hip_decode_device_prepass(lora_hip_decoder_t *h, const void *d_iq, size_t total_items,
                                               const uint64_t *stream_off, const uint64_t *stream_len,
                                               uint32_t n_streams, void *hip_stream, uint32_t flags);

/* Diagnostics: where the segment planner's energy-envelope pre-pass sees the streams go quiet (the start of every gap
 * between bursts; lora_hip_decode_device cuts its speculation segments there when the traffic is dense enough).
 * counts[i] = gap starts found in stream i; their item positions, relative to the stream and ascending, are packed stream
 * after stream into pos[0 .. sum(counts)) (LORA_HIP_ERR_OVERFLOW when cap is too small).  Streams shorter than one
 * symbol or a samples/symbol below 128 give LORA_HIP_ERR_BAD_CONFIG.  Not part of the reference: it has no scheduler. */
lora_hip_status lora_hip_gap_starts_device(lora_hip_decoder_t *h, const void *d_iq, size_t total_items,
                                           const uint64_t *stream_off, const uint64_t *stream_len, uint32_t n_streams,
                                           int64_t *pos, size_t cap, uint32_t *counts, void *hip_stream);

/* ---- "frames" message port (decoder_impl.cc:120, 588-609) ----------------------------------------------- */
size_t          lora_hip_frames_available(const lora_hip_decoder_t *h);
/* Pops the oldest frame: blob = loratap_header_t (15 B, include/lora/loratap.h:35-55) | loraphy_header_t
 * (3 B, include/lora/loraphy.h:25-32) | payload (+2 CRC bytes).  info may be NULL.                          */
lora_hip_status lora_hip_poll_frame(lora_hip_decoder_t *h, uint8_t *buf, size_t cap, size_t *len,
                                    lora_hip_frame_info_t *info);

/* Pops up to max_frames frames in one call: blobs are packed back to back into buf (infos[i].length bytes
 * each, in order).  Stops early when the next blob would not fit.  *n_frames receives the count.            */
lora_hip_status lora_hip_drain_frames(lora_hip_decoder_t *h, uint8_t *buf, size_t cap, lora_hip_frame_info_t *infos,
                                      size_t max_frames, size_t *n_frames);

/* Pops up to max_slots frames into fixed-size slots of slot_bytes each (the layout the multi-GPU frame gather
 * exchanges): u32 stream | u32 length | i64 header_pos | blob | zero padding.  OVERFLOW if a blob does not fit. */
lora_hip_status lora_hip_drain_slots(lora_hip_decoder_t *h, uint8_t *slots, size_t slot_bytes, size_t max_slots,
                                     size_t *n_frames);

/* ---- symbol-level access for the +-1-bin tests (get_shift_fft :430-464, gradient :466-491) -------------- */
/* offsets (host array, n entries): symbol start item indices into d_iq; bins_out (host, n entries):
 * the raw return value of the selected demodulator (FFT: shift s; GRAD: s-1).                              */
lora_hip_status lora_hip_demod_symbols_device(lora_hip_decoder_t *h, const void *d_iq, size_t total_items,
                                              const int64_t *offsets, size_t n, int demod,
                                              uint32_t *bins_out, void *hip_stream);

/* Same, and additionally fine_out[i] = d_fine_sync after fine_sync(bin_idx, max(D/4, 2)) on that window
 * (decoder_impl.cc:300-338, called from demodulate() :514-518).  fine_out may be NULL; non-NULL needs one of the
 * fast demodulator families (SF7-SF12 at decimation 8: wave-per-symbol at SF7 / SF8, workgroup-per-symbol at SF9-SF12;
 * SF7-SF9 at decimation 2 / 4 and SF6 at 4 / 8: wave-per-symbol; every demod mode), else BAD_CONFIG.          */
lora_hip_status lora_hip_demod_symbols_ex_device(lora_hip_decoder_t *h, const void *d_iq, size_t total_items,
                                                 const int64_t *offsets, size_t n, int demod,
                                                 uint32_t *bins_out, int32_t *fine_out, void *hip_stream);

/* ---- introspection --------------------------------------------------------------------------------------- */
lora_hip_status lora_hip_last_timing(const lora_hip_decoder_t *h, lora_hip_timing_t *t);
/* Name of the state-machine kernel this handle's passes launch (diagnostics; what a rocprofv3 kernel trace will show): before the first pass
 * the default of the configuration, afterwards the variant the last pass's main launch ran.
 * walker2_kernel_sf7/8[_grad] (wavefront per symbol; walker2_kernel_sf7/8[_grad]_wide: the 256-register builds of the same body, for launches with no more jobs than
 * CUs), walker3_kernel_sf9..12[_grad] (SF9 and every _grad: wavefront per symbol; SF10-SF12 FFT: workgroup per symbol; _half: SF10 and SF9_grad as two
 * workgroups per CU), *_skip (the header-only variants of a decoupled pass), walker2_kernel_sf7..9_d2 / _d4[_grad] (decimation 2 / 4: samp_rate = 2 or 4
 * x bandwidth), walker2_kernel_sf6_d8 / _d4[_grad] (SF6), walker_kernel* (generic: other decimations, SF10-SF12 at decimation 2 / 4, LORA_HIP_NO_FAST).                                      */
const char     *lora_hip_walker_kernel_name(const lora_hip_decoder_t *h);
/* The payload pass of the last pass when it ran decoupled (LORA_HIP_FLAG_NO_DECOUPLED above): packets whose payload it took; how many of them
 * ended off the zero-drift grid (their symbols moved the symbol clock by a net amount: the job is split there and probed like a segment boundary);
 * how many it handed back to the complete kernels; rounds of symbol reads (1 + one per distinct clock offset met); symbol reads in all; device
 * time of its kernels (part of lora_hip_timing_t.walker_ms).  All zero when the pass was not decoupled.                                  */
lora_hip_status lora_hip_last_payload_pass(const lora_hip_decoder_t *h, uint32_t *packets, uint32_t *moved, uint32_t *rerun, uint32_t *rounds,
                                           uint32_t *symbols, float *ms);

/* How the last pass cut its streams into speculation segments (diagnostics): *burst_aware = 1 when the cuts were placed
 * in the gaps between bursts found by the energy-envelope pre-pass, 0 for the fixed grid (configured segment length,
 * sparse or weak traffic, tracing); *segments = segments = workgroups of the main launch.                      */
lora_hip_status lora_hip_last_plan(const lora_hip_decoder_t *h, uint32_t *burst_aware, uint32_t *segments);
/* The handle's ideal-chirp tables (diagnostics; ABI 4): what build_ideal_chirps (decoder_impl.cc:141-175) leaves in d_downchirp (which = 0, 2 sps floats:
 * re, im), d_upchirp (1, 2 sps floats), d_downchirp_ifreq (2, sps floats), d_upchirp_ifreq (3, sps floats) and d_upchirp_ifreq_v (4, 3 sps floats, followed
 * by this library's guard tail of 4 D + 8 copies of the last value: the reference indexes past the vector for bin_idx = N - 1, :301,:310).  Tables 0, 2, 3, 4
 * are read back from the device memory the kernels use; the upchirp itself is needed by no kernel and is the host copy its ifreq tables were made from.
 * *n_floats = the table's length; buf (cap_floats floats) may be NULL to ask for the length only.  LORA_HIP_ERR_ARG: unknown table, or buf too small. */
lora_hip_status lora_hip_get_table(const lora_hip_decoder_t *h, int which, float *buf, size_t cap_floats, size_t *n_floats);
size_t          lora_hip_trace(const lora_hip_decoder_t *h, const lora_hip_step_t **steps);
void            lora_hip_trace_clear(lora_hip_decoder_t *h);

/* ---- explicit CFO estimate -----------------------------------------------------------------
 * decoder_impl::experimental_determine_cfo (decoder_impl.cc:730-738) on caller-given windows of the device-resident
 * IQ: each window (samples_per_symbol items from offsets[i], meant to be an aligned preamble upchirp such as
 * detect_upchirp's result, :771-776) is multiplied by the ideal downchirp and the instantaneous frequency of the
 * product is taken.  mode 0 = the reference's estimate, the single value at index 256 (bit-compatible with the
 * compiled reference up to the float tolerance of two atan2f; useless under noise - which may be why its only call
 * is commented out upstream); mode 1 = the mean over the window.  Hz, positive = the signal sits above the tuned
 * frequency.  What upstream meant to do with it: publish ("cfo", value) on the block's "control" port for
 * channelizer::apply_cfo (channelizer_impl.cc:68-71) - gr_lora_amd.lora.lora_receiver(cfo_correction=True) does.   */
lora_hip_status lora_hip_estimate_cfo_device(lora_hip_decoder_t *h, const void *d_iq, size_t total_items,
                                             const int64_t *offsets, size_t n, int mode, float *cfo_hz_out,
                                             void *hip_stream);

/* Diagnostics of the strict SYNC path (LORA_HIP_FLAG_FAST_SYNC above): the arithmetic it re-evaluates near-tied shifts with, on
 * caller-given device samples.  arg_out[i] = atan2f(im, re) of item i as glibc's libm computes it (std::arg, decoder_impl.cc:232-233),
 * ifreq_out[i] = instantaneous_frequency's value for the pair (i, i+1) (:231-240), i < n_items - 1.  Host buffers of n_items and
 * n_items - 1 floats.  tests/test_gpu_strict_sync.py holds both to the host's libm bit for bit.                                     */
lora_hip_status lora_hip_ref_ifreq_device(lora_hip_decoder_t *h, const void *d_iq, size_t n_items, float *arg_out, float *ifreq_out,
                                          void *hip_stream);

/* ---- FFT-domain preamble detection (beyond the reference) --------------------------------------
 * The reference acquires a packet with a time-domain autocorrelation of adjacent symbols (detect_preamble_autocorr,
 * decoder_impl.cc:340-366, gate 0.90 at :755) and an instantaneous-frequency correlation against the ideal downchirp (gate
 * 0.96 at :792); both need the signal above the noise of the whole sample-rate band (nothing acquired at <= 20 dB).
 * These two calls look where LoRa's processing gain is - the dechirped spectrum of get_shift_fft (:430-464) - and acquire
 * down to the sensitivity of the spreading factor (SF12: about -20 dB in-band).  No reference behaviour exists to be
 * identical to: the definition is oracle/preamble_oracle.py (float64), to which the device results are held.
 *
 * lora_hip_window_stats_device: for n windows of samples_per_symbol items at offsets[i] (host array, item indices into
 * d_iq) the peak bin, peak power and total power over the N bins k in [-N/2, N/2) (index k mod N) of the sps-point DFT of
 * x * d_downchirp ("down": sees upchirps) and of conj(x) * d_downchirp ("up": sees downchirps, bins mirrored).
 *
 * lora_hip_detect_preambles_device: scans every stream at one window per symbol; a preamble is a run of >= 4 consecutive
 * windows whose peak-to-mean ratio peak (N-1) / (total - peak) reaches `threshold` (0: ln N + 3) with peak bins agreeing
 * within +-1; the symbol clock is aligned to the run (a carrier offset is absorbed into the alignment exactly as the
 * reference's SYNC step absorbs it, :392-413), the SFD is the first pair of aligned windows dominated by downchirps, and
 * the clock is then moved by up to half a bin so that the preamble sits at the CENTRE of bin 0 (what is left of the timing would
 * otherwise split every data symbol's peak between two bins), and header_pos = SFD + 2.25 symbols (:820-824): feed it to
 * lora_hip_demod_symbols_device at header_pos + k * sps, or to lora_hip_decode_at_headers_device.
 * cfo_hz: from the opposite displacement of up- and downchirps (resolution bw / 2N).  LORA_HIP_ERR_OVERFLOW when cap is
 * too small (*n_found = what was found).                                                                            */
typedef struct lora_hip_window_stats {
    int32_t bin_down; float peak_down, total_down;
    int32_t bin_up;   float peak_up, total_up;
} lora_hip_window_stats_t;
lora_hip_status lora_hip_window_stats_device(lora_hip_decoder_t *h, const void *d_iq, size_t total_items, const int64_t *offsets,
                                             size_t n, lora_hip_window_stats_t *out, void *hip_stream);
typedef struct lora_hip_preamble {
    int64_t  header_pos;          /* item index within the stream of the first header symbol, on the aligned symbol clock */
    int64_t  run_pos;             /* item index of the first window of the run                                            */
    uint32_t stream;
    uint32_t run_len;             /* windows of the run                                                                   */
    int32_t  bin;                 /* the run's peak bin before alignment (timing + carrier offset, in bins)               */
    int32_t  sfd_index;           /* aligned window, counted from the run's first, where the SFD begins                   */
    float    pmr;                 /* mean peak-to-mean ratio over the run                                                  */
    float    cfo_bins, cfo_hz;    /* carrier offset estimate                                                               */
    int32_t  delta;               /* sub-bin timing refinement applied to header_pos, samples (-D/2 .. D/2)               */
} lora_hip_preamble_t;
lora_hip_status lora_hip_detect_preambles_device(lora_hip_decoder_t *h, const void *d_iq, size_t total_items,
                                                 const uint64_t *stream_off, const uint64_t *stream_len, uint32_t n_streams,
                                                 float threshold, lora_hip_preamble_t *out, size_t cap, size_t *n_found,
                                                 void *hip_stream);

/* Decodes the packets whose first header symbol the caller already knows - the detector's output - without the reference's own
 * acquisition: one job per entry starts in DECODE_HEADER at pre[i].header_pos of stream pre[i].stream (decoder_impl.cc:826 onwards:
 * header, payload, fine_sync as configured, the integer chain) and publishes its frame to the handle's queue (info.stream,
 * info.header_pos; the loratap SNR byte is 0: no DETECT step fed the power queue).  Explicit header only.  With
 * lora_hip_detect_preambles_device in front this is a receive path that works where the reference's acquires nothing: BASELINE
 * config 5 (SF12, 255 bytes, carrier offset, -10 dB in-band) decodes end to end (tests/test_gpu_detect.py).  Use an FFT demodulator;
 * the gradient estimator and fine_sync's ifreq correlation need tens of dB (disable_drift_correction = 1 below ~10 dB).          */
lora_hip_status lora_hip_decode_at_headers_device(lora_hip_decoder_t *h, const void *d_iq, size_t total_items,
                                                  const uint64_t *stream_off, const uint64_t *stream_len, uint32_t n_streams,
                                                  const lora_hip_preamble_t *pre, size_t n, void *hip_stream);

/* ---- frame validity (beyond the reference) -------------------------------------------------
 * The reference publishes every frame it demodulates and checks nothing: "CRC checks of the payload and header"
 * are the first item of its list of unsupported features (README.md:12), include/lora/utilities.h:396-404 is a
 * header_checksum() stub that returns true (its comment holds the parity sets implemented here), and the PHY CRC
 * is read out but unused (decoder_impl.cc:839).  lora_hip_check_frame evaluates both checks on a published blob
 * (15 B loratap | 3 B PHY header | payload | 2 B CRC when the header says so), on the host, without a handle:
 *   header: the 5 checksum bits (low nibble of PHY byte 1, high nibble of PHY byte 2) against the parity sets
 *           over the 12 header bits length[7..0], cr[2..0], has_crc;
 *   CRC:    CRC-16/CCITT (0x1021, init 0) over all payload bytes but the last two, XORed with those two
 *           (low byte last); the transmitter does not whiten the CRC field, the reference de-whitens it like
 *           data (decoder_impl.cc:643), so the received value is first XORed with the whitening bytes of its
 *           two positions (x^8 + x^6 + x^5 + x^4 + 1, seed 0xff: the sequence lib/tables.h:30-44 decodes to).
 * Known answer: the README frame 04 90 40 de ad be ef 70 0d passes both.  Implicit-header frames carry no
 * header on air (the 3 PHY bytes of their blob are synthesised from the constructor's cr / crc, decoder_impl.cc:588-600):
 * has_header is inferred from the blob length agreeing with those bytes, which an implicit frame can match by
 * coincidence - it is then reported with has_header = 1, header_checksum_ok = 0 and a crc_ok computed from a length that
 * was never on air.  The result is only meaningful for explicit-header decoders; a caller that runs implicit mode knows
 * so and should not ask.                                                                                             */
typedef struct lora_hip_frame_check {
    uint8_t  has_header;          /* blob length agrees with the PHY header's length / has_crc fields        */
    uint8_t  header_checksum_ok;
    uint8_t  has_crc;             /* PHY header bit 4 of byte 1                                              */
    uint8_t  crc_ok;              /* 0 when has_crc is 0                                                     */
    uint8_t  header_checksum_rx, header_checksum_calc; /* 5 bits each                                        */
    uint16_t crc_rx, crc_calc;    /* as transmitted (whitening undone) / computed                            */
    uint16_t reserved;
} lora_hip_frame_check_t;
lora_hip_status lora_hip_check_frame(const uint8_t *blob, size_t len, lora_hip_frame_check_t *out);

#ifdef __cplusplus
}
#endif
#endif /* LORA_HIP_H */
