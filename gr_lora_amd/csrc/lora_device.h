// lora_device.h -- structures shared between the host runtime and the HIP kernels.
//
// HBM data layout (all device-resident, allocated once per decoder handle):
//   IQ stream      cf32[items]            caller-owned (or the handle's staging buffer), read-only
//   chirp tables   down cf32[sps], up_ifreq f32[sps], down_ifreq f32[sps],
//                  up_ifreq_v f32[3*sps+pad], twN cf32[N/2], tws cf32[sps]   (L2-resident, <= 1.2 MB)
//   jobs           Job[n_jobs]            one workgroup each
//   attempts       AttemptRec[n_jobs * recs_per_job]
//   job results    JobResult[n_jobs]
//   scratch        f32[n_jobs * 2*sps]    only when the ifreq window does not fit in LDS (SF >= 11)
#pragma once
#include <stdint.h>

#if !defined(__HIPCC__) && !defined(HIP_INCLUDE_HIP_HIP_RUNTIME_H)
struct float2 { float x, y; }; // host-only builds (tests/host_sim) do not pull in HIP
#endif

namespace lora_hip {

constexpr int kWG = 256;              // threads per workgroup (4 wavefronts of 64)
constexpr int kMaxBinsPerThread = 16; // N / kWG at SF12
constexpr int kMaxFrame = 3 + 255 + 2;
constexpr int kMaxCodewords = 640;    // (sf-7) + ceil(2*257/ppm)*ppm  <= 5 + 44*12

enum DecState : int32_t { kDetect = 0, kSync, kFindSfd, kPause, kDecodeHeader, kDecodePayload, kStop };

enum AttemptStatus : uint32_t {
    kAttemptNone = 0,
    kAttemptFrame = 1,       // packet decoded, frame[] valid
    kAttemptLostSync = 2,    // d_corr_fails > 4 -> back to DETECT (decoder_impl.cc:808-813)
    kAttemptOutOfData = 3,   // fewer than 2*sps items left mid-attempt (scheduler stops calling work())
    kAttemptAtHeader = 4,    // probe mode: stopped on entering DECODE_HEADER
    kAttemptAtSfd = 5,       // tail probe with Job.tail_stop_sfd: stopped behind its first FIND_SFD step; sfd_pos / sfd_fails[n_sfd - 1] = the state it stopped in
    kAttemptHeaderOnly = 6   // LaunchCfg.skip_payload: header decoded, the payload symbols left to the symbol-parallel pass (SkippedPayload in frame[]); end_pos assumes
                             // that none of them moves the symbol clock
};
// what a walker launched with LaunchCfg.skip_payload leaves in AttemptRec.frame (frame_len = sizeof) instead of a frame
struct SkippedPayload {
    uint8_t  phdr[3];          // d_phdr as parsed (:831-835)
    uint8_t  n_left;           // codewords of the header block behind the five header ones (:632): sf - 7
    uint8_t  left[8];
    int32_t  payload_symbols;  // :842-847
};
constexpr int kMaxSfdRec = 12; // FIND_SFD steps of an attempt whose entry state is recorded (a preamble of 8 + sync word + SFD: at most 11)

struct DevParams {
    uint32_t sf, nbins, nbins_hdr, sps, decim, log_nbins, delay_after_sync;
    uint32_t implicit, reduced_rate, enable_fine_sync, demod_mode;
    uint32_t ctor_cr, ctor_crc;
    uint32_t use_fast;          // allow the wave-per-symbol decode rounds (SF7/SF8, D = 8)
    uint32_t fft_groups;        // passes over the symbol in get_shift_fft (1 for SF <= 10 at D = 8)
    uint32_t fft_stride;        // LDS row stride (in cf32) of one polyphase array
    uint32_t lds_work_bytes;    // size of the shared work area
    uint32_t ifreq_in_lds_1;    // sps floats fit in LDS work area
    uint32_t ifreq_in_lds_2;    // 2*sps floats fit in LDS work area
    float    down_ifreq_avg;    // chirp_avg over sps-1 points (decoder_impl.cc:287)
    float    down_ifreq_sd;     // stddev of ideal downchirp ifreq (decoder_impl.cc:289)
    float    down_ifreq_dsum;   // sum over sps-1 points of (down_ifreq - down_ifreq_avg): float residue
    double   sync_a, sync_b;    // least-squares line a + b*k through d_upchirp_ifreq[0 .. sps-2] (closed-form SYNC)
    uint32_t sync_closed_form;  // use the O(sps) SYNC (sps >= 4096)
    uint32_t strict_sync;       // SYNC: shifts within rounding of the closed-form maximum are re-evaluated with the reference's own float sums (lora_strict_sync.inc.hip)
    uint32_t samples_per_second; // d_samples_per_second (:74), for the CFO estimate's Hz scale
    // fine_sync (:300-338) decided in closed form by the wave demodulator (SF7 / SF8; lora_wave_demod.inc.hip, FMODE 2): d_upchirp_ifreq_v
    // is a ramp of slope ffs_alpha with ONE step of ffs_jump - ffs_alpha inside any window fine_sync looks at (bin_idx < N-1), so
    // c(lag+1) - c(lag) = ffs_alpha * sum(ifreq) + ffs_jump * ifreq[at the step]; what the table's float noise adds is at most ffs_tol
    uint32_t ffs_on;
    float    ffs_alpha, ffs_jump, ffs_tol;
    const float2 *down;         // d_downchirp
    const float  *up_ifreq;     // d_upchirp_ifreq
    const float  *down_ifreq;   // d_downchirp_ifreq
    const float  *up_ifreq_v;   // d_upchirp_ifreq_v (+ guard tail)
    const float2 *twN;          // e^{-2 pi i t / N},   t < N/2
    const float2 *tws;          // e^{-2 pi i m / sps}, m < sps
    const float  *wave_tabs;    // packed table block of the wave demodulator (lora_wave_demod.inc.hip: SF7-SF9 at D = 8; lora_wave_decim.inc.hip: D = 2 / 4)
    const float2 *w3_tw;        // walker3 (SF9-12 at D = 8): W_N^t, and the combine coefficients in pass-3 thread order
    const float2 *w3_ctab;
    const float  *team_tabs;    // SF10-12 at D = 8: the table block of the team demodulator (lora_team_demod.inc.hip)
};

struct Job {
    uint64_t stream_off;   // item index of the stream's first sample inside the IQ buffer
    uint64_t stream_len;   // items in the stream
    int64_t  start;        // DETECT position where this job starts (relative to the stream)
    int64_t  scan_limit;   // no new DETECT step is started at pos >= scan_limit
    uint32_t stream_id;
    uint32_t cr_prev;      // d_phdr.cr carried in (constructor value or previous packet's, :655)
    uint32_t max_attempts; // stop after this many attempts (0 = unlimited up to capacity)
    uint32_t stop_at_header; // probe mode
    int64_t  probe_limit;  // > scan_limit: having reached scan_limit, the job goes on as its successor's probe (stop at the
                           // next header, no new DETECT step at pos >= probe_limit) and reports that part as the "tail"; 0: off
    uint32_t start_at_header; // 1: `start` is the first header symbol of a packet acquired elsewhere (lora_hip_decode_at_headers_device, after
                           // the FFT-domain preamble detector): the job begins in DECODE_HEADER with an attempt open instead of in DETECT
    uint32_t tail_stop_sfd; // 1: the tail probe (probe_limit) stops behind its FIRST FIND_SFD step instead of at the header (kAttemptAtSfd): two
                           // trajectories that start a FIND_SFD step at the same sample with the same d_corr_fails are one from there on
                           // (:785-818 read nothing else), and the successor's attempt records hold every such state it went through
};

struct AttemptRec {
    int64_t  start_pos;    // DETECT position where this attempt's scan began
    int64_t  trig_pos;     // DETECT position that triggered (autocorr >= 0.90)
    int64_t  hdr_pos;      // position entering DECODE_HEADER (-1 if never reached)
    int64_t  end_pos;      // position after the attempt (next DETECT position)
    uint32_t status;       // AttemptStatus
    uint32_t npush;        // d_pwr_queue pushes during this attempt's DETECT scan (incl. trigger step)
    float    push_tail[4]; // the last min(4, npush) pushed values, oldest first
    uint32_t cr_prev;      // d_phdr.cr used for the header FEC branch
    uint32_t hdr_ambig;    // header bytes differ between the {3,4} and {1,2} FEC branches
    uint32_t frame_len;    // 3 + payload_length
    uint32_t n_symbols;    // header + payload symbols demodulated
    // the state at the START of each FIND_SFD step of this attempt (walker3; other kernels report n_sfd = 0): what a tail probe that
    // stopped early (kAttemptAtSfd) is matched against
    int64_t  sfd_pos[kMaxSfdRec];
    uint32_t n_sfd;
    uint8_t  sfd_fails[kMaxSfdRec];
    uint8_t  frame[kMaxFrame + 4];
};

struct JobResult {
    int64_t  final_pos;    // DETECT position where the job stopped
    uint32_t n_attempts;
    uint32_t final_cr;     // d_phdr.cr after the last attempt
    uint32_t npush;        // pushes of the trailing DETECT scan (after the last attempt)
    float    push_tail[4];
    uint32_t stop_reason;  // 0 scan_limit reached, 1 out of data, 2 attempt capacity, 3 max_attempts / probe stop
    uint32_t n_steps;      // trace entries written
    uint32_t pad;
    uint32_t cyc[6];       // shader clocks / 64 spent per state (DETECT, SYNC, FIND_SFD, PAUSE, HEADER, PAYLOAD); walker2 only
    uint32_t rounds[6];    // rounds per state
    // tail probe (Job.probe_limit): what a separate probe job started from this job's end state would have reported
    int64_t  tail_final_pos;
    uint32_t tail_valid, tail_first_rec, tail_n_attempts, tail_final_cr, tail_npush, tail_stop_reason, tail_pad, tail_rsv;
    float    tail_push_tail[4];
    uint32_t ctl[4];       // control wavefront inside decode rounds, shader clocks / 64: state copy-in, symbol loop, plan, copy-out
    uint32_t dbg[6];       // diagnostics (walker2): HW_ID, XCC_ID, s_memrealtime (100 MHz) at the start and the end of the job, clock64 / 64 likewise
};

struct StepRec {           // mirrors lora_hip_step_t
    int32_t  state, consumed;
    int64_t  pos;
    int32_t  bin, fine;
    float    value;
    uint32_t stream;
    uint32_t cycles;     // shader clocks spent in this step (s_memtime), tracing only
    uint32_t pad;
};

// host-side launchers implemented in lora_kernels.hip
struct LaunchCfg {
    const float2 *iq;
    const Job *jobs;
    JobResult *results;
    AttemptRec *recs;
    uint32_t recs_per_job;
    float *scratch;          // n_jobs * 2*sps floats or nullptr
    StepRec *trace;          // n_jobs * trace_cap or nullptr
    uint32_t trace_cap;
    uint32_t n_jobs;
    uint32_t *balance;       // kBalanceWords words, zero-initialised once (walker2: progress exchange between the workgroups of a CU), or nullptr
    uint32_t skip_payload;   // 1: the launch runs the kernels' header-only variant (walker3): a packet's attempt ends behind its header (kAttemptHeaderOnly) and the
                             // job goes on in DETECT where the payload would end if no symbol moved the clock; the payload pass (launch_payload_pass) does the rest
};
constexpr uint32_t kBalanceCus = 2048;                 // (XCC_ID, SE, SH, CU) flattened
constexpr uint32_t kBalanceWords = 3u * kBalanceCus;   // per CU: remaining work of its two workgroups, claim counter

// burst envelope pre-pass (segment planning): one entry per stream, blocks of one symbol
struct EnvStream {
    uint64_t off;          // first item of the stream
    uint32_t first_block;  // index of its first block in E (a multiple of 64: one bitmap word per wavefront)
    uint32_t n_blocks;
};
int launch_envelope(const float2 *iq, const EnvStream *streams /* host table; first_block a multiple of 64 */, uint32_t n_streams, uint32_t sps,
                    float *d_E, unsigned long long *d_bitmap /* one bit per block of E */, void *stream);

int launch_walker(const DevParams &p, const LaunchCfg &c, void *stream);
const char *walker_kernel_name(const DevParams &p);                        // the kernel launch_walker picks for this configuration
// (walker3 symbol kernels) second reads: when symbol s moved the symbol clock (fine[s] != 0) and symbol s + 1 is its successor in the same packet
// (offsets[s + 1] == offsets[s] + sps), the workgroup that demodulated s reads s + 1 again fine[s] samples further on and leaves shift[s + 1] = fine[s]
// and the result in bins / fine [s + 1] of THIS set: on a clean signal the successor moves the clock straight back, and the payload pass's walk
// gets past the pair without another round.  shift[] must be zero on entry; windows outside [0, max_start] are not read.
struct DemodAlt { int32_t *shift; uint32_t *bins; int32_t *fine; int64_t max_start; };
int launch_demod_symbols(const DevParams &p, const float2 *iq, const int64_t *d_offsets, uint32_t n,
                         int demod, uint32_t *d_bins, int32_t *d_fine, float *scratch, void *stream, const DemodAlt *alt = nullptr);
int launch_detect_windows(const DevParams &p, const float2 *iq, const int64_t *d_offsets, uint32_t n, void *d_out /* 24 B per window */, void *stream); // N4: lora_detect.inc.hip
// The payload pass of a launch with LaunchCfg.skip_payload: every payload symbol of every packet demodulated on its own (launch_demod_symbols over
// the zero-drift positions), then one workgroup per packet takes the symbols through demodulate()'s integer chain (:506-529, :866-881).
constexpr int kPayloadHyp = 4;
struct PayloadDesc {         // one packet (host-written)
    uint32_t n_walk;         // symbols DECODE_PAYLOAD demodulates: payload_symbols, or 1 when that is <= 0 (:866-870 run behind the first symbol)
    uint32_t n_hyp;          // demodulated so far: symbols [hyp_from[h], hyp_to[h]) read hyp_shift[h] samples behind their zero-drift positions,
    int32_t  hyp_shift[kPayloadHyp]; // results at bins / fine [hyp_base[h] + symbol]
    int32_t  hyp_base[kPayloadHyp];
    uint32_t hyp_from[kPayloadHyp];
    uint32_t hyp_to[kPayloadHyp];    // (the symbols from hyp_to on would start less than two symbols before the end of the data, :91)
    int64_t  room;           // symbol j read c samples behind its zero-drift position lies inside the data (:91) iff j sps + c <= room
    SkippedPayload sk;
};
enum PayloadWalk : uint32_t { kWalkComplete = 0, kWalkOutOfData = 1, kWalkNeedShift = 2 };
struct PayloadOut {          // (device-written)
    uint32_t result;         // PayloadWalk: the walk along the symbols ended with the frame / at a symbol that does not fit into the data / at a symbol
    uint32_t at;             // (`at`) that has not been demodulated `shift` samples behind its zero-drift position yet
    int32_t  shift;          // kWalkComplete: samples the symbols moved the symbol clock by in all; kWalkNeedShift: the shift wanted
    uint32_t frame_len;
    uint8_t  frame[kMaxFrame + 4];
};
int launch_payload_chain(const DevParams &p, const uint32_t *d_bins, const int32_t *d_fine, const DemodAlt &alt, const PayloadDesc *descs, PayloadOut *outs, uint32_t n_packets, void *stream);
bool walker_has_skip_variant(const DevParams &p);                           // LaunchCfg.skip_payload is honoured (walker3, explicit header)
int launch_ref_ifreq(const float2 *x, uint32_t n, float *d_arg, float *d_ifreq, void *stream); // diagnostics: the strict SYNC path's atan2f / ifreq
int launch_cfo(const DevParams &p, const float2 *iq, const int64_t *d_offsets, uint32_t n, int mode, float *d_out, void *stream); // N4: explicit CFO estimate
bool walker3_covers(uint32_t sf);                                          // SF9-12: lora_walker3.inc.hip
uint32_t w3_tw_entries(uint32_t sf);
void build_w3_tables(uint32_t sf, float2 *tw, float2 *ctab /* sps entries */);
uint32_t wave_tables_floats(uint32_t sf);                                  // 0 when the wave demodulator does not cover sf
void build_wave_tables(uint32_t sf, const float2 *down, float *out);
uint32_t wave_tables_floats_d(uint32_t sf, uint32_t decim);                // decimation 2 / 4 (lora_wave_decim.inc.hip); 0 when not covered
void build_wave_tables_d(uint32_t sf, uint32_t decim, const float2 *down, float *out);
uint32_t team_tables_entries(uint32_t sf);                                 // 8-byte entries; 0 when the team demodulator does not cover sf
void build_team_tables(uint32_t sf, const float2 *down, double dt, double bandwidth, float2 *out);
uint32_t walker_lds_bytes(const DevParams &p);
uint32_t walker_resident_slots(const DevParams &p);
uint32_t walker_resident_slots_full(const DevParams &p);                  // second, smaller slot count where the kernel exists as half- and full-size workgroups (0: none)
const char *walker_kernel_name_for(const DevParams &p, uint32_t n_jobs, bool skip = false); // ... and the variant a launch of n_jobs jobs runs (skip: LaunchCfg.skip_payload)

} // namespace lora_hip
