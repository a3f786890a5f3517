// lora_walker2.inc.hip -- the walker of the wave-per-symbol family: SF7 / SF8 at decimation 8 and, since round 6, SF7 / SF8 / SF9 at decimation 2 / 4
// (walker2_body's LD; lora_wave_decim.inc.hip); every demodulator, explicit or implicit header.  Included by lora_kernels.hip.
//
// Same state machine as walker_body (decoder_impl::work, lib/decoder_impl.cc:740-903), organised in
// ROUNDS: WAVES wavefronts = WAVES-1 workers + 1 control wavefront (512 threads: 7 workers; two workgroups per CU
// at SF7 and, since round 5, at SF8.  A 16-wavefront workgroup with 15 workers was built and measured 30 % slower).  In DETECT, FIND_SFD and DECODE_*
// every worker evaluates upcoming symbol windows at pos + w*sps (zero drift assumed: one per round in DECODE_*, two in FIND_SFD,
// one or four in DETECT); the control
// thread then replays the reference's per-call logic over the results in order and stops at the first
// one whose outcome invalidates the later windows (a trigger, a state change, d_fine_sync != 0, end of
// data).  The accepted sequence is therefore exactly the serial one.  DECODE rounds are PIPELINED: while
// the control thread resolves round r, the workers already demodulate round r+1 at the predicted position
// (same state, one round further; once the header is decoded the control thread knows how many payload
// symbols remain and plans exactly that many windows); a misprediction only discards that round.  SYNC (one
// correlation per packet) is computed by all wavefronts together.  The decoder state lives in LDS and is touched by the
// control thread only; the per-round PLAN is double-buffered so that no wavefront can read a plan that is
// being rewritten.

constexpr int kW2DetectK = 4;   // DETECT windows per worker and round
#ifndef LORA_W2_WAVES_SF7
#define LORA_W2_WAVES_SF7 8
#endif
#ifndef LORA_W2_WAVES_SF8
#define LORA_W2_WAVES_SF8 8
#endif
#ifndef LORA_W2_SFD_K
#define LORA_W2_SFD_K 2 // FIND_SFD windows per worker and round
#endif
#ifndef LORA_W2_EU_SF8
#define LORA_W2_EU_SF8 4 // wavefronts per SIMD the SF8 kernel's register budget is set for (round 5: 128 registers and 73 KB of LDS, two workgroups per CU as at SF7: +19 % - until then 2: 256 registers, one)
#endif
// wavefronts per workgroup the shared structures are sized for (SF7: 16-wave workgroups, one per CU, were measured slower:
// all 15 workers hit their load and VALU phases together)
constexpr int kW2MaxWaves = LORA_W2_WAVES_SF7 > LORA_W2_WAVES_SF8 ? LORA_W2_WAVES_SF7 : LORA_W2_WAVES_SF8;

constexpr int kW2SfdK = LORA_W2_SFD_K;

enum W2Mode : int32_t { kPlanExit = 0, kPlanDetect, kPlanSync, kPlanSfd, kPlanPause, kPlanDecode, kPlanFinalize };

struct alignas(16) W2Plan {
    int64_t pos;          // where the workers evaluate their windows
    int32_t mode;         // W2Mode
    int32_t buf;          // spec buffer the workers fill (kPlanDecode)
    int32_t resolve_prev; // kPlanDecode: spec[buf ^ 1] holds the previous round, to be resolved now
    int32_t n_win;        // windows the workers evaluate this round (kPlanDecode: 0 = resolve-only round)
    int32_t prev_n;       // kPlanDecode with resolve_prev: windows of the previous round (it ended n_win * sps before pos)
    int32_t zmode;        // kPlanDecode / kPlanSfd: the round evaluates its windows with the ZM instantiations - every ifreq value formed as the reference forms
                          // it next to a sample of exactly zero (std::arg(0) = 0; lora_kernels.hip, ifreq_prod_z).  Planned when a window of the previous
                          // round came back POISONED (kPoisonBin / W2SfdOut.pz): the fast evaluations notice such a sample but do not handle it
};

struct alignas(16) W2State {
    int64_t  pos, att_start, att_trig, att_hdr;
    int32_t  state, done, stop_reason, in_attempt;
    uint32_t corr_fails, cr, has_crc, payload_length;
    int32_t  payload_symbols;
    uint32_t n_words, n_cw, n_sym;
    uint32_t n_att, npush, n_steps, frame_ok;
    uint32_t att_cr_prev, att_ambig;
    float    energy_threshold;
    float    push_tail[4];
    uint8_t  phdr[4];
    // payload finalisation requested by thread 0, executed by the whole workgroup
    uint32_t fin_pending, fin_n_bytes, fin_plen;
    int32_t  fin_st, fin_consumed, fin_bin, fin_fine;
};

struct alignas(16) W2Stats { // per-state time accounting (reported under LORA_HIP_DEBUG), kept out of the hot state
    long long prev_t;
    int32_t   prev_state;
    uint32_t  cyc[6], rounds[6], ctl[4];
};

struct alignas(16) W2Shared {
    float    red[kW2MaxWaves * 72 + 8];
    float    specf[kW2MaxWaves][4];
    float    detf[kW2MaxWaves * kW2DetectK][4]; // DETECT: d0, d1, e1, e2 per window
    int32_t  detn[kW2MaxWaves];                 // DETECT: valid windows of each worker
    int32_t  speci[2][kW2MaxWaves][4];   // [buffer][worker]: decode rounds are double-buffered
    W2Plan   plan[2];
    Shared   sh;      // words / codewords / decoded bytes (shared with the integer-chain helpers)
    W2State  st;
    W2Stats  stats;
    int64_t  ph_start;    // hand-over from the job proper to its tail probe (Job.probe_limit): start position,
    uint32_t ph_go, ph_cr, ph_natt, ph_pad; // go flag, d_phdr.cr, attempt records used so far
    uint32_t det_streak;  // consecutive DETECT rounds planned so far (control thread): the first of a streak looks at fewer windows
    uint32_t prio;        // priority of the worker wavefronts for the coming rounds (see the balance exchange in the round loop)
    uint32_t words_pk[2]; // d_words of the current block, one byte per word (words < 256 for SF <= 8); control thread only
    strict::Cands sc;     // SYNC: near-tied shifts for the exact re-evaluation
    uint32_t words_pk16[4]; // SF9 (decimation 2 / 4 builds): d_words of the current block as 16-bit fields; control thread only
};

struct W2Tabs {
    const float  *v;     // d_upchirp_ifreq_v (+ guard)
    const float  *dd;    // d_downchirp_ifreq[k] - chirp_avg
    float        *scratch; // 72 floats per wavefront
};

template <int WAVES>
__device__ __forceinline__ void w2_block_argmax_first(float &v, int &idx, float *red)
{
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        const float ov = __shfl_xor(v, o, 64);
        const int oi = __shfl_xor(idx, o, 64);
        if (ov > v || (ov == v && oi < idx)) { v = ov; idx = oi; }
    }
    __syncthreads();
    if (lane == 0) { red[wave] = v; ((int *)red)[16 + wave] = idx; }
    __syncthreads();
    v = red[0]; idx = ((int *)red)[16];
#pragma unroll
    for (int w = 1; w < WAVES; w++) {
        const float ov = red[w];
        const int oi = ((int *)red)[16 + w];
        if (ov > v || (ov == v && oi < idx)) { v = ov; idx = oi; }
    }
}

// ---- thread-0 bookkeeping (mirrors end_step / the loop-top checks of walker_body) ---------------------
__device__ __forceinline__ bool w2_pre_step(W2State &S, const Job &job, uint32_t rec_cap, uint32_t sps)
{
    if (S.state == kDetect && !S.in_attempt) {
        if (S.pos >= job.scan_limit) { S.stop_reason = 0; S.done = 1; return false; }
        if (S.n_att >= rec_cap) { S.stop_reason = 2; S.done = 1; return false; }
        if (job.max_attempts && S.n_att >= job.max_attempts) { S.stop_reason = 3; S.done = 1; return false; }
    }
    if (S.pos + 2 * (int64_t)sps > (int64_t)job.stream_len) { S.stop_reason = 1; S.done = 1; return false; } // :91
    return true;
}

// `lead`: this lane performs the global stores (the decode rounds run this on the whole control wavefront, uniformly).
// KEEP_SFD: the caller (walker3) maintains AttemptRec.n_sfd itself; otherwise a finished attempt reports none.
template <bool KEEP_SFD = false>
__device__ __forceinline__ void w2_end_step(W2State &S, const Job &job, const LaunchCfg &C, AttemptRec *recs, StepRec *trace, int32_t st_in,
                            int32_t consumed, int32_t step_bin, int32_t fine, float step_val, long long t_start, bool lead = true)
{
    if (trace && lead && S.n_steps < C.trace_cap) {
        StepRec &s = trace[S.n_steps];
        s.state = st_in; s.consumed = consumed; s.pos = S.pos; s.bin = step_bin; s.fine = fine; s.value = step_val;
        s.stream = job.stream_id; s.cycles = (uint32_t)(clock64() - t_start); s.pad = 0;
    }
    S.n_steps++;
    S.pos += consumed;
    if (S.in_attempt && S.state == kDetect) { // attempt finished: frame published, or sync lost
        if (lead) {
            AttemptRec &r = recs[S.n_att];
            r.status = S.frame_ok ? kAttemptFrame : kAttemptLostSync;
            if (!S.frame_ok) r.frame_len = 0;
            r.start_pos = S.att_start; r.trig_pos = S.att_trig; r.hdr_pos = S.att_hdr; r.end_pos = S.pos;
            r.npush = S.npush;
            for (int i = 0; i < 4; i++) r.push_tail[i] = S.push_tail[i];
            r.cr_prev = S.att_cr_prev; r.hdr_ambig = S.att_ambig; r.n_symbols = S.n_sym;
            if constexpr (!KEEP_SFD) r.n_sfd = 0;
        }
        S.n_att++;
        S.in_attempt = 0;
        S.frame_ok = 0;
        S.npush = 0;
        S.att_start = S.pos;
    } else if (S.in_attempt && job.stop_at_header && S.state == kDecodeHeader) {
        S.stop_reason = 3;
        S.done = 1;
    }
}

// everything demodulate()/work() do once the bin is known (:506-529, :826-886); thread 0 only.
// Returns true when the payload is complete: the caller must run the workgroup-wide finalisation.
// WAVE: called by the whole (converged) control wavefront with identical arguments; the deinterleaver then uses the lanes.
// do_demod = false: an implicit-header payload step whose energy fell under the threshold (:861-864) - no word, the packet ends.
// W16: the words as 16-bit fields of wpk[0..3] (SF9) instead of bytes of wpk[0..1]
template <bool WAVE = false, bool W16 = false>
__device__ __forceinline__ bool w2_post_symbol(const DevParams &P, W2State &S, Shared &sh, uint32_t bin_idx, bool is_first, uint32_t *wpk = nullptr, bool do_demod = true)
{
    const bool reduced = is_first || P.reduced_rate; // :495
    bool block_done = false;
    if (do_demod) {
    if (reduced) bin_idx = (uint32_t)lroundf((float)bin_idx / 4.0f) & (P.nbins_hdr - 1u); // :507-509 (% N/4, a power of two, of a value >= 0)
    const uint32_t word = bin_idx ^ (bin_idx >> 1u); // :512
    const uint32_t need = 4u + (is_first ? 4u : S.cr); // :521
    if constexpr (WAVE && W16) {
        const uint32_t sh16 = 16u * (S.n_words & 1u), ins = (word & 0xffffu) << sh16, keep = ~(0xffffu << sh16), q = S.n_words >> 1;
        if (q == 0u) wpk[0] = (wpk[0] & keep) | ins; // (no dynamic index: the array stays in registers)
        else if (q == 1u) wpk[1] = (wpk[1] & keep) | ins;
        else if (q == 2u) wpk[2] = (wpk[2] & keep) | ins;
        else if (q == 3u) wpk[3] = (wpk[3] & keep) | ins;
    } else
    if constexpr (WAVE) { // the block's words stay in a register: no LDS round trip per symbol
        const uint32_t sh8 = 8u * (S.n_words & 3u), ins = (word & 0xffu) << sh8, keep = ~(0xffu << sh8);
        if (S.n_words < 4u) wpk[0] = (wpk[0] & keep) | ins;
        else if (S.n_words < 8u) wpk[1] = (wpk[1] & keep) | ins;
    } else {
        if (S.n_words < 16u) sh.words[S.n_words] = word;
    }
    S.n_words++;
    S.n_sym++;
    if (S.n_words == need) {
        const uint32_t ppm = reduced ? P.sf - 2u : P.sf;
        uint32_t tmp = S.n_cw;
        if constexpr (WAVE && W16) { const uint32_t w4[4] = {wpk[0], wpk[1], wpk[2], wpk[3]}; deinterleave_block_wave16(sh, w4, need, ppm, tmp); }
        else if constexpr (WAVE) deinterleave_block_wave(sh, ((uint64_t)wpk[1] << 32) | wpk[0], need, ppm, tmp); // ppm <= 8 on this kernel's SFs
        else deinterleave_block(sh, need, ppm, tmp);
        S.n_cw = (S.n_cw + ppm <= (uint32_t)kMaxCodewords) ? S.n_cw + ppm : (uint32_t)kMaxCodewords;
        S.n_words = 0;
        block_done = true;
    }
    }
    if (is_first) {
        if (block_done && P.implicit) { // :828-829: no header on air
            S.payload_symbols = 1;
            S.state = kDecodePayload;
        } else
        if (block_done) { // decode(true) and header parse (:831-847)
            uint8_t hA[3], hB[3], h0[3] = {0, 0, 0};
            decode_header_bytes(sh, S.n_cw, 2, hA);
            decode_header_bytes(sh, S.n_cw, 1, hB);
            const uint8_t *use = (S.cr >= 3u) ? hA : (S.cr >= 1u ? hB : h0);
            S.att_ambig = (uint32_t)((hA[0] != hB[0]) || (hA[1] != hB[1]) || (hA[2] != hB[2]));
            S.phdr[0] = use[0]; S.phdr[1] = use[1]; S.phdr[2] = use[2];
            const uint32_t rem = S.n_cw > 5u ? S.n_cw - 5u : 0u; // erase the 5 header codewords (:632)
            for (uint32_t i = 0; i < rem; i++) sh.cw[i] = sh.cw[i + 5u];
            S.n_cw = rem;
            if ((S.phdr[1] >> 5) > 4) S.phdr[1] = (uint8_t)((S.phdr[1] & 0x1f) | (4u << 5)); // :834-835
            S.cr = S.phdr[1] >> 5;
            S.has_crc = (S.phdr[1] >> 4) & 1u;
            S.payload_length = (uint32_t)S.phdr[0] + 2u * S.has_crc; // :838
            const uint32_t redundancy = P.reduced_rate ? 2u : 0u; // :842-847
            const int symbols_per_block = (int)S.cr + 4;
            const float bits_needed = (float)S.payload_length * 8.0f;
            const float symbols_needed = bits_needed * ((float)symbols_per_block / 4.0f) / (float)(P.sf - redundancy);
            const int blocks_needed = (int)ceilf(symbols_needed / (float)symbols_per_block);
            S.payload_symbols = blocks_needed * symbols_per_block;
            S.state = kDecodePayload;
        }
        return false;
    }
    if (block_done && !P.implicit) S.payload_symbols -= (int32_t)(4u + S.cr); // :866-867
    if (S.payload_symbols <= 0) { // :870-881
        uint32_t n_bytes;
        if (S.cr >= 3u) n_bytes = (uint32_t)ceilf((float)S.n_cw * 4.0f / (4.0f + (float)S.cr)); // :658
        else n_bytes = (S.n_cw + 1u) / 2u;
        if (n_bytes > (uint32_t)(kMaxCodewords / 2 + 8)) n_bytes = kMaxCodewords / 2 + 8;
        S.fin_n_bytes = n_bytes;
        S.fin_plen = S.payload_length > 257u ? 257u : S.payload_length;
        return true;
    }
    return false;
}

// ---- wave-level window evaluations ---------------------------------------------------------------------
// DETECT (:340-366): sums of c1*conj(c2), |c1|^2, |c2|^2 over one symbol pair
template <int SF>
__device__ __forceinline__ float4 w2_detect_window(const float2 *p_)
{
    const auto p = (const __attribute__((address_space(1))) v2f *)p_; // global, not flat, loads
    constexpr int SPS = 8 << SF, J = SPS / 64;
    const int lane = threadIdx.x & 63;
    float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
    constexpr int B = J < 16 ? J : 16; // loads are issued B + B at a time (left to itself the compiler waits for every pair)
#pragma unroll
    for (int jb = 0; jb < J; jb += B) {
        v2f u[B], w[B];
#pragma unroll
        for (int j = 0; j < B; j++) u[j] = p[(jb + j) * 64 + lane];
#pragma unroll
        for (int j = 0; j < B; j++) w[j] = p[SPS + (jb + j) * 64 + lane];
        __builtin_amdgcn_sched_group_barrier(0x020, 2 * B, 0); // all VMEM reads of the batch first (left alone the scheduler sinks them into the sums, one wait each)
        __builtin_amdgcn_sched_group_barrier(0x002, 16 * B, 0);
#pragma unroll
        for (int j = 0; j < B; j++) {
            const v2f c1 = u[j], c2 = w[j];
            a0 += c1.x * c2.x + c1.y * c2.y;
            a1 += c1.y * c2.x - c1.x * c2.y;
            a2 += c1.x * c1.x + c1.y * c1.y;
            a3 += c2.x * c2.x + c2.y * c2.y;
        }
    }
    return make_float4(wave_sum_rows(a0), wave_sum_rows(a1), wave_sum_rows(a2), wave_sum_rows(a3));
}

// K consecutive DETECT windows (pos, pos + sps, ...) by one wavefront: the K + 1 symbols are read once,
// each energy sum serves two windows.  Per window the sums are formed exactly as in w2_detect_window.
template <int SF, int K>
__device__ __forceinline__ void w2_detect_windows(const float2 *__restrict__ p, int nvalid, float (&out)[K][4])
{
    constexpr int SPS = 8 << SF, J = SPS / 64;
    const int lane = threadIdx.x & 63;
    float d0[K], d1[K], e[K + 1];
#pragma unroll
    for (int i = 0; i < K; i++) { d0[i] = 0.f; d1[i] = 0.f; }
#pragma unroll
    for (int s = 0; s <= K; s++) e[s] = 0.f;
#pragma unroll 4 // 4 (K + 1) loads in flight; a full unroll hoists all 16 (K + 1) and spills
    for (int j = 0; j < J; j++) {
        float2 c[K + 1];
#pragma unroll
        for (int s = 0; s <= K; s++) c[s] = (s <= nvalid) ? p[s * SPS + j * 64 + lane] : make_float2(0.f, 0.f); // nvalid is wave-uniform
#pragma unroll
        for (int i = 0; i < K; i++) {
            d0[i] += c[i].x * c[i + 1].x + c[i].y * c[i + 1].y;
            d1[i] += c[i].y * c[i + 1].x - c[i].x * c[i + 1].y;
        }
#pragma unroll
        for (int s = 0; s <= K; s++) e[s] += c[s].x * c[s].x + c[s].y * c[s].y;
    }
#pragma unroll
    for (int s = 0; s <= K; s++) e[s] = wave_sum_rows(e[s]);
#pragma unroll
    for (int i = 0; i < K; i++) { out[i][0] = wave_sum_rows(d0[i]); out[i][1] = wave_sum_rows(d1[i]); out[i][2] = e[i]; out[i][3] = e[i + 1]; }
}

// FIND_SFD (:385-390, :283-298, :801-803): Pearson correlation of the window's ifreq with the ideal
// downchirp ifreq, and -- for an upchirp (c < -0.97) -- fine_sync(-1, 4*D) over the 63 lags.
struct W2SfdOut { float c; int32_t fine; int32_t pz; }; // pz: the window holds a sample of exactly zero (its sums are NaN): to be evaluated again with ZM = true
template <int SF, bool ZM = false, int SEARCH = 32 /* fine_sync(-1, 4 D), :801-803: the lags |i| < 4 D (decimation 2 / 4: 8 / 16) */>
__device__ __forceinline__ W2SfdOut w2_sfd_window(const float2 *p, const float *Tv, const float *Tdd, float *scr /* this wavefront's 72 floats */,
                                                          float down_ifreq_sd, float down_ifreq_dsum, double sync_a, double sync_b)
{
    constexpr int SPS = 8 << SF, J = SPS / 64;
    int lane = threadIdx.x & 63;
    asm volatile("" : "+v"(lane));
    float f[J];
    {
        const auto pv = (const __attribute__((address_space(1))) v2f *)p; // global, not flat, loads
        v2f a[J]; // the whole window in one round of loads (lane owns n = 64 j + lane)
#pragma unroll
        for (int j = 0; j < J; j++) a[j] = pv[j * 64 + lane];
        __builtin_amdgcn_sched_barrier(0); // (all loads issued before the first use)
        v2f bprev = (v2f){0.0f, 0.0f};
#pragma unroll
        for (int j = 0; j < J; j += 2) { // ifreq[n-1] = arg(x[n] conj(x[n-1])), two per packed evaluation; x[n-1] by wave rotate
            const v2f b0 = dpp2<kDppWaveRor1>(a[j]), b1 = dpp2<kDppWaveRor1>(a[j + 1]);
            const v2f p0 = (lane == 0) ? bprev : b0, p1 = (lane == 0) ? b0 : b1;
            bprev = b1;
            const v2f fp = ZM ? ifreq_prod_pk_z(p0, a[j], p1, a[j + 1]) : ifreq_prod_pk(p0, a[j], p1, a[j + 1]);
            f[j] = (j == 0 && lane == 0) ? 0.0f : fp.x;
            f[j + 1] = fp.y;
        }
    }
    // one-pass Pearson over the n = sps-1 points k = 0 .. sps-2
    float a0 = 0.f, a1 = 0.f, a2 = 0.f;
#pragma unroll
    for (int j = 0; j < J; j++) {
        const int k = j * 64 + lane - 1;
        const float fk = f[j];
        const float d = Tdd[k < 0 ? 0 : k];
        a0 += fk; a1 += fk * fk; a2 += fk * d; // f[j] is 0 for the non-existent k = -1
    }
    a0 = wave_sum_rows(a0); a1 = wave_sum_rows(a1); a2 = wave_sum_rows(a2);
    if (!ZM && poisoned(a0)) return W2SfdOut{0.0f, 0, 1}; // (uniform)
    const float n = (float)(SPS - 1);
    const float average = a0 / n;
    const float var = fmaxf(a1 / n - average * average, 0.0f);
    const float sd = sqrtf(var) * down_ifreq_sd;
    const float c = (a2 - average * down_ifreq_dsum) / sd / n;
    if (!(c < -0.97f) || c > 0.96f) return W2SfdOut{c, 0, 0};
    // fine_sync(-1, 32) (:300-321): c_i = sum_{k<sps} fe[k] * v[sps + i + k], i = -31 .. 31, with fe[sps-1] = fe[sps-2].
    // v is the ifreq of concatenated upchirps: v[m] = a + b*(m mod sps) except the one wrap sample per period
    // (m mod sps == sps-1), up to float noise ~1e-5 of the sums.  With t = (i+k) mod sps that gives, exactly in
    // the line model,
    //   i >= 0: c_i = a G0 + b (i G0 + G1) - b sps T_i    + Wd fe[sps-1-i],   T_i = sum of the last i samples
    //   i <  0: c_i = a G0 + b (i G0 + G1) + b sps H_{-i} + Wd fe[-i-1],      H_j = sum of the first j samples
    // G0 = sum fe[k], G1 = sum k fe[k], Wd = v[2 sps - 1] - (a + b (sps-1)).  O(sps) instead of 63 x sps.
    double g0 = 0.0, g1 = 0.0;
#pragma unroll
    for (int j = 0; j < J; j++) {
        const int k = j * 64 + lane - 1; // lane 0, j = 0 holds f = 0 (k = -1)
        g0 += (double)f[j];
        g1 += (double)k * (double)f[j];
    }
    if (lane == 63) { g0 += (double)f[J - 1]; g1 += (double)(SPS - 1) * (double)f[J - 1]; } // duplicated last tap (:243)
    {
        // wave-wide sums of the two doubles (as hi/lo float pairs would lose bits; use shuffles on 64-bit values)
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) { g0 += __shfl_xor(g0, o, 64); g1 += __shfl_xor(g1, o, 64); }
    }
    // the first 32 and last 33 samples go to this wavefront's scratch: head[k] = fe[k], tail[q] = fe[sps-1-q]
    if (lane >= 1 && lane <= 32) scr[lane - 1] = f[0];                 // fe[0..31]
    if (lane >= 31) scr[32 + 1 + (63 - lane)] = f[J - 1];              // fe[sps-2-(63-lane)] -> tail[1 + (63-lane)]
    if (lane == 63) scr[32] = f[J - 1];                                // tail[0] = fe[sps-1] = fe[sps-2]
    __builtin_amdgcn_wave_barrier();
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    const int i = lane - 31; // this lane's lag
    // running sums of the head (lanes 0..31) and of the tail (lanes 32..63): lane 32 + q ends up with T_{q+1}, lane q with
    // H_{q+1}.  At most 32 terms of magnitude < pi: float keeps them to ~1e-6, far below the spacing of the c_i.
    float ps = scr[lane];
#pragma unroll
    for (int o = 1; o < 32; o <<= 1) {
        const float up = __shfl_up(ps, o, 32);
        if ((lane & 31) >= o) ps += up;
    }
    const float hsum = __shfl(ps, lane < 31 ? 30 - lane : 0, 64); // H_{-i} for the negative lags
    float c_i = -3.0e38f;
    bool in_range = lane <= 62;
    if constexpr (SEARCH < 32) in_range = in_range && i > -SEARCH && i < SEARCH;
    if (in_range) {
        const double a = sync_a, b = sync_b;
        const double wd = (double)Tv[2 * SPS - 1] - (a + b * (double)(SPS - 1));
        double edge;
        float wrap_f;
        if (i >= 0) {
            edge = i > 0 ? -(double)ps : 0.0;                           // -T_i
            wrap_f = scr[32 + i];                                       // fe[sps-1-i]
        } else {
            edge = (double)hsum;                                        // H_{-i}
            wrap_f = scr[-i - 1];                                       // fe[-i-1]
        }
        c_i = (float)(a * g0 + b * ((double)i * g0 + g1) + b * (double)SPS * edge + wd * (double)wrap_f);
    }
    int li = lane;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) { // first maximum in lag order (strict '>' scan from 0, :311)
        const float ov = __shfl_xor(c_i, o, 64);
        const int oi = __shfl_xor(li, o, 64);
        if (ov > c_i || (ov == c_i && oi < li)) { c_i = ov; li = oi; }
    }
    const int32_t lag = (c_i > 0.0f) ? li - 31 : 0;
    return W2SfdOut{c, -lag, 0};
}

// SYNC (:770-783, detect_upchirp :392-413): this thread's best shift of the sliding correlation of f[0 .. 2 sps) with the
// ideal upchirp ifreq.  Kept out of line: it runs once per packet and its double-precision temporaries would otherwise
// raise the register pressure of the whole state machine.
template <int SF, int WAVES>
__device__ __attribute__((noinline)) void w2_sync_closed_form(const float *f2, double *pre, double sync_a, double sync_b, float &bv, int &bi, float &b2, int &i2)
{
    constexpr uint32_t sps = 8u << SF, kW2 = 64u * WAVES;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
                // C[i] = sum_{k < n} f[i+k] u[k], n = sps-1, i < sps.  d_upchirp_ifreq is a line a + b k (up to float noise
                // ~1e-5 of the peak, below the rounding of the reference's own float sum), so
                //   C[i] = a S0[i] + b S1[i],   S0[i] = sum_k f[i+k],   S1[i] = sum_k k f[i+k]
                // and both come from prefix sums of f and (t - sps) f in double: O(sps) instead of O(sps^2).
                // Prefixes are kept per chunk of CH samples (256 chunks); a thread adds the few samples up to its shift.
                constexpr uint32_t NCH = 2u * sps / 4u < 256u ? 2u * sps / 4u : 256u, CH = 2u * sps / NCH, n = sps - 1u; // (chunks of at least one float4: sps = 256 has 128)
                double sF = 0.0, sG = 0.0;
                if (threadIdx.x < NCH) {
    #pragma unroll
                    for (uint32_t q = 0; q < CH; q += 4u) {
                        const uint32_t t = threadIdx.x * CH + q;
                        const float4 fv = *reinterpret_cast<const float4 *>(f2 + t);
                        const double tc = (double)((int)t - (int)sps);
                        sF += ((double)fv.x + (double)fv.y) + ((double)fv.z + (double)fv.w);
                        sG += tc * (double)fv.x + (tc + 1.0) * (double)fv.y + (tc + 2.0) * (double)fv.z + (tc + 3.0) * (double)fv.w;
                    }
                }
                double iF = sF, iG = sG; // inclusive scan over the wavefront's chunks
    #pragma unroll
                for (int o = 1; o < 64; o <<= 1) {
                    const double uF = __shfl_up(iF, o, 64), uG = __shfl_up(iG, o, 64);
                    if (lane >= o) { iF += uF; iG += uG; }
                }
                if (threadIdx.x < NCH && lane == 63) { pre[2u * NCH + wave] = iF; pre[2u * NCH + 4u + wave] = iG; }
                __syncthreads();
                if (threadIdx.x < NCH) { // exclusive prefix of every chunk
                    double oF = iF - sF, oG = iG - sG;
                    for (int w = 0; w < wave; w++) { oF += pre[2u * NCH + w]; oG += pre[2u * NCH + 4u + w]; }
                    pre[threadIdx.x] = oF; pre[NCH + threadIdx.x] = oG;
                }
                __syncthreads();
                constexpr uint32_t R = (sps + kW2 - 1u) / kW2; // consecutive shifts per thread (the last threads may have none)
                const uint32_t i0 = threadIdx.x * R;
                bv = 0.0f; // max_correlation = 0 (:400)
                bi = 0x7fffffff;
                b2 = 0.0f; // second best of this thread: the other half of a near-tie (lora_strict_sync.inc.hip)
                i2 = 0x7fffffff;
                if (sps % kW2 != 0u && i0 >= sps) return;
                auto prefix_at = [&](uint32_t j, double &F, double &G) { // sums over t < j
                    const uint32_t c = j / CH;
                    F = pre[c]; G = pre[NCH + c];
                    for (uint32_t t = c * CH; t < j; t++) { const double fv = (double)f2[t]; F += fv; G += (double)((int)t - (int)sps) * fv; }
                };
                double F0, G0, F1, G1;
                prefix_at(i0, F0, G0);
                prefix_at(i0 + n, F1, G1);
                double s0 = F1 - F0, s1 = (G1 - G0) + (double)((int)sps - (int)i0) * s0;
    #pragma unroll
                for (uint32_t r = 0; r < R; r++) {
                    const uint32_t i = i0 + r;
                    if (sps % kW2 != 0u && i >= sps) break;
                    const float c = (float)(sync_a * s0 + sync_b * s1);
                    if (c > bv) { b2 = bv; i2 = bi; bv = c; bi = (int)i; }
                    else if (c > b2) { b2 = c; i2 = (int)i; }
                    const double fin = (double)f2[i + n], fout = (double)f2[i];
                    s0 += fin - fout;
                    s1 += (double)n * fin - s0;
                }
}

// strict SYNC (lora_strict_sync.inc.hip): the window's instantaneous frequency as the REFERENCE computes it - two atan2f and the unwrap
// (:231-240), bit for bit - so that the closed form and the re-evaluation of its near-ties share the arctangents (one per sample: every
// thread's arguments pass through f2 to the neighbour that needs them).  Out of line: it runs once per packet, and its registers must not
// be the state machine's.
template <int SF, int WAVES>
__device__ __attribute__((noinline)) void w2_sync_exact_ifreq(const float2 *__restrict__ x, float *f2)
{
    constexpr int SPS = 8 << SF, kW2 = 64 * WAVES, NI = 2 * SPS / kW2;
    static_assert(NI * kW2 == 2 * SPS, "whole samples per thread");
    typedef __attribute__((address_space(3))) float lds_f;
    lds_f *fl = (lds_f *)f2;
    float2 xv[NI];
    float am[NI], an[NI];
#pragma unroll
    for (int k = 0; k < NI; k++) xv[k] = x[threadIdx.x + (uint32_t)k * kW2];
#pragma unroll
    for (int k = 0; k < NI; k++) { am[k] = strict::fd_atan2f(xv[k].y, xv[k].x); fl[threadIdx.x + (uint32_t)k * kW2] = am[k]; }
    __syncthreads();
#pragma unroll
    for (int k = 0; k < NI; k++) { const uint32_t i = threadIdx.x + (uint32_t)k * kW2; an[k] = fl[i + 1u < 2u * SPS ? i + 1u : i]; }
    __syncthreads();
#pragma unroll
    for (int k = 0; k < NI; k++) fl[threadIdx.x + (uint32_t)k * kW2] = strict::ref_ifreq(am[k], an[k]); // (the last one, 0, is overwritten by the caller: :243)
}

// ---- the ZM evaluations (a window that holds a sample of exactly zero): they run in rounds of their own (W2Plan.zmode), for the rare window only; inside them
// every ifreq value is a call of ifreq_prod_z -----------------------------------------------------------------------------------------------------
#ifndef LORA_W2_ZM_ATTR
#define LORA_W2_ZM_ATTR __forceinline__ // (as calls - noinline - they cost the gradient kernels 7 % in register allocation, and the calls themselves faulted: profiles/r05_ab_zero_samples.txt)
#endif
template <int SF, int SEARCH = 32>
__device__ LORA_W2_ZM_ATTR W2SfdOut w2_sfd_window_zm(const float2 *p, const float *Tv, const float *Tdd, float *scr, float down_ifreq_sd, float down_ifreq_dsum, double sync_a, double sync_b)
{
    return w2_sfd_window<SF, true, SEARCH>(p, Tv, Tdd, scr, down_ifreq_sd, down_ifreq_dsum, sync_a, sync_b);
}
template <int SF> constexpr bool kW2Alias = SF == 8; // SYNC's work areas inside the FFT table block (walker2_body ALIAS)
struct W2DemodZ { uint32_t s; int32_t fine; float en; };
template <int SF, bool GRAD, int LD = 3>
__device__ LORA_W2_ZM_ATTR W2DemodZ w2_demod_zm(uint32_t enable_fine_sync, uint32_t demod_mode, bool want_energy, WaveTabs T, const float2 *x)
{
    DevParams Q{}; // (only the fields the demodulators read)
    Q.enable_fine_sync = enable_fine_sync; Q.demod_mode = demod_mode;
    W2DemodZ r{0u, 0, 0.0f};
    if constexpr (LD != 3 || SF < 7) {
        if constexpr (GRAD) wave_demod_symbol_grad_d<SF, LD, true>(Q, T.v, x, want_energy, r.s, r.fine, r.en);
        else wave_demod_symbol_d<SF, LD, true>(Q, T, x, r.s, r.fine, want_energy ? &r.en : nullptr);
    } else
    if constexpr (GRAD) wave_demod_symbol_grad<SF, true>(Q, T.v, x, want_energy, r.s, r.fine, r.en);
    else wave_demod_symbol<SF, 0, true>(Q, T, x, r.s, r.fine, want_energy ? &r.en : nullptr);
    return r;
}
// 8-byte entries of the demodulator's table block (decimation 8: lora_wave_demod.inc.hip; 2 / 4: lora_wave_decim.inc.hip)
template <int SF, int LD> constexpr uint32_t w2_table_entries() { if constexpr (LD == 3 && SF >= 7) return WaveGeom<SF>::n_ent; else return WaveGeomD<SF, LD>::n_ent; }

// ---- the kernel -----------------------------------------------------------------------------------------
// GRAD: the reference's shipped demodulator (max_frequency_gradient_idx, :466-491, :499) in the decode rounds instead of the
// dechirp + FFT: wave_demod_symbol_grad.  The FFT twiddle block is then not needed in LDS.
// SKIP: the header-only variant of a decoupled pass (LaunchCfg.skip_payload; docs/LAB_NOTEBOOK.md 4.13, as walker3's): behind the header parse (:831-847) the attempt is
// closed as kAttemptHeaderOnly - the record carries d_phdr, the header block's spare codewords and d_payload_symbols - and the job goes on in DETECT where
// DECODE_PAYLOAD would end if no symbol moved the symbol clock; the payload pass (launch_demod_symbols + payload_chain_kernel) does the rest.
// LD: log2 of the decimation.  2 / 1 (decimation 4 / 2, round 6): the same state machine on windows of sps = D N samples - the window-level helpers are
// instantiated for the SF that has this sps at decimation 8 (GSF), the demodulators come from lora_wave_decim.inc.hip, FIND_SFD's fine_sync searches 4 D lags.
template <int SF, int WAVES, bool GRAD, bool SKIP = false, int LD = 3>
__device__ __forceinline__ void walker2_body(const DevParams &P, const LaunchCfg &C)
{
    constexpr int N = 1 << SF, SPS = N << LD;
    constexpr int GSF = SF + LD - 3; // 8 << GSF == SPS
    constexpr uint32_t NENT = w2_table_entries<SF, LD>();
    constexpr int kW2 = 64 * WAVES, kW2Workers = WAVES - 1; // the last wavefront is the control wavefront
    // Decode rounds look at kW2Win windows: one per worker.  (A second window for the worker that shares its SIMD with the mostly waiting control
    // wavefront was built and measured 13 % slower - the SIMD time-slices its wavefronts evenly, docs/LAB_NOTEBOOK.md 5.2 - and is gone from the sources.)
    constexpr int kW2Win = kW2Workers;
    static_assert(kW2Win <= kW2MaxWaves, "speci is sized for kW2MaxWaves windows");
    static_assert(WAVES <= kW2MaxWaves, "W2Shared is sized for kW2MaxWaves wavefronts");
    constexpr uint32_t sps = SPS;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    W2Shared &W = *reinterpret_cast<W2Shared *>(smem);
    W2State &S = W.st;
    Shared &sh = W.sh;
    // LDS carve-up after W2Shared: f2[2 sps] | v[3 sps + 40 (padded to 4)] | dd[sps] | twiddle block of the wave
    // demodulator | chunk prefix sums of the SYNC correlation
    // ALIAS (SF8): SYNC's two work areas lie IN the twiddle block - which only decode rounds read - and every SYNC round ends by copying the
    // block in again (36 KB from L2, once per acquisition): 73 KB instead of 94 KB, two workgroups per CU
    constexpr bool ALIAS = kW2Alias<SF> && !GRAD && LD == 3;
    constexpr uint32_t NV = (3u * SPS + 40u + 3u) & ~3u;
    float *lds0 = reinterpret_cast<float *>(smem + ((sizeof(W2Shared) + 15) & ~(size_t)15));
    float *vl = ALIAS ? lds0 : lds0 + 2 * SPS;
    float *ddl = vl + NV;
    v2f *tab2 = reinterpret_cast<v2f *>(ddl + SPS);
    float *f2 = ALIAS ? reinterpret_cast<float *>(tab2) : lds0;
    double *pre = ALIAS ? reinterpret_cast<double *>(f2 + 2 * SPS)
                        : reinterpret_cast<double *>(tab2 + (GRAD ? 0u : NENT)); // SYNC: chunk prefix sums (2 x 256) + per-wavefront totals
    static_assert(!ALIAS || (2u * SPS * 4u + (2u * 256u + 8u) * 8u <= NENT * 8u), "SYNC's work areas fit the table block");

    const uint32_t jid = blockIdx.x;
    if (jid >= C.n_jobs) return;
    Job job = C.jobs[jid];           // (phase 1 rewrites start / limits: the job carries on as the next segment's probe)
    if constexpr (SF == 7) job = uniform_job(job); // (scalar registers for the stream base and the limits: 54 -> 40 spilled VGPRs, +1.5 %; SF8 does not spill)
    uint32_t rec_cap = C.recs_per_job; // ... with what is left of the attempt-record capacity
    const float2 *__restrict__ X = C.iq + job.stream_off;
    const int64_t n_items = (int64_t)job.stream_len;
    AttemptRec *recs = C.recs + (size_t)jid * C.recs_per_job;
    StepRec *trace = C.trace ? C.trace + (size_t)jid * C.trace_cap : nullptr;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const bool is_ctl = wave == kW2Workers;                 // control wavefront
    const bool t0 = threadIdx.x == kW2Workers * 64;         // the control thread: sole owner of the decoder state S
    if (is_ctl) __builtin_amdgcn_s_setprio(2);              // its serial bookkeeping is on every round's critical path

    const uint32_t dbg_t0 = (uint32_t)__builtin_amdgcn_s_memrealtime(), dbg_c0 = (uint32_t)(clock64() >> 6);
    WaveTabs FT{};
    if constexpr (GRAD) { // only the ifreq template
        for (uint32_t i = threadIdx.x; i < 3u * sps + 40u; i += kW2) vl[i] = P.up_ifreq_v[i];
        FT.v = vl;
    } else {
        if constexpr (LD == 3 && SF >= 7) FT = wave_tabs_to_lds<SF>(P, tab2, vl, kW2);
        else FT = wave_tabs_to_lds_d<SF, LD>(P, tab2, vl, kW2);
    }
    for (uint32_t i = threadIdx.x; i < sps; i += kW2) ddl[i] = P.down_ifreq[i] - P.down_ifreq_avg;

    // plan for the next round from the TRUE state (control thread only)
    auto plan_from = [&](W2State &S, W2Plan &pl, bool zreq = false /* a window of this round came back poisoned: the next round's are evaluated by the ZM instantiations */) {
        pl.buf = 0; pl.resolve_prev = 0; pl.n_win = kW2Workers; pl.pos = S.pos; pl.prev_n = 0; pl.zmode = zreq ? 1 : 0;
        if (!S.done) (void)w2_pre_step(S, job, rec_cap, sps);
        if (S.done) { pl.mode = kPlanExit; return; }
        if (S.fin_pending) { pl.mode = kPlanFinalize; return; }
        if (S.state != kDetect) W.det_streak = 0u;
        switch (S.state) {
        case kDetect:
            // windows per worker: a preamble usually follows within a few symbols of a job's start or a packet's end, so
            // the first round of a streak evaluates one window per worker and only the following ones kW2DetectK
            pl.mode = kPlanDetect; pl.n_win = W.det_streak++ == 0u ? 1 : kW2DetectK;
            return;
        case kSync: pl.mode = kPlanSync; break;
        case kFindSfd: pl.mode = kPlanSfd; break;
        case kPause: pl.mode = kPlanPause; break;
        default:
            pl.mode = kPlanDecode; pl.n_win = kW2Win;
            if (S.state == kDecodePayload && !P.implicit) { // symbols left in the packet (:866-870); implicit: unknown until the energy drops
                const int32_t rem = S.payload_symbols - (int32_t)S.n_words;
                pl.n_win = rem < kW2Win ? (rem > 0 ? rem : 1) : kW2Win;
            }
            break;
        }
    };
    auto plan_from_state = [&](W2Plan &pl) { plan_from(S, pl); };

    W2Tabs T{vl, ddl, W.red};
    // Balance between the two workgroups of a CU.  The SIMDs favour the older wavefronts, so of two equal jobs sharing a
    // CU the one dispatched first runs ahead and finishes early, and the other then runs out its rest alone at well under
    // the CU's throughput (measured: 407 vs 495 us for equal work).  Each control thread therefore publishes its job's
    // remaining work every round and reads its neighbour's; the workers of the job with more left run at raised priority.
    uint32_t *bal_mine = nullptr, *bal_other = nullptr;
    uint32_t bal_rem = 0xffffffffu, bal_seen = 0u, bal_slot = 0u; // control thread only
    if (t0) W.prio = 0u;
    if (t0 && C.balance) {
        const uint32_t hw = __builtin_amdgcn_s_getreg((31 << 11) | 4), xcc = __builtin_amdgcn_s_getreg((31 << 11) | 20);
        const uint32_t cu = ((xcc & 7u) << 8) | ((hw >> 8) & 0xffu);
        bal_slot = atomicAdd(&C.balance[2u * kBalanceCus + cu], 1u) & 1u;
        bal_mine = C.balance + 2u * cu + bal_slot;
        bal_other = C.balance + 2u * cu + (bal_slot ^ 1u);
    }
    // Phase 0 is the job proper.  Phase 1 (Job.probe_limit): having reached its scan limit the workgroup runs what a
    // separate probe job started from its end state would run -- a FRESH job (state re-initialised, tables kept) that
    // stops at the next header -- and reports it as the "tail"; the host then needs no second launch.
    for (int phase = 0; phase < 2; phase++) {
    if (t0) {
        S = W2State{};
        S.state = kDetect; S.pos = job.start; S.cr = job.cr_prev; S.has_crc = P.ctor_crc;
        S.phdr[1] = (uint8_t)((P.ctor_cr << 5) | (P.ctor_crc << 4));
        S.att_start = job.start; S.att_trig = -1; S.att_hdr = -1; S.att_cr_prev = job.cr_prev;
        if (job.start_at_header && phase == 0) { S.state = kDecodeHeader; S.in_attempt = 1; S.att_trig = job.start; S.att_hdr = job.start; } // (acquired elsewhere)
        if (phase == 0) W.stats = W2Stats{};
        W.stats.prev_state = -1;
        W.det_streak = 0u;
        plan_from_state(W.plan[0]);
    }

    for (uint32_t it = 0;; it++) {
        __syncthreads(); // plan[it & 1] and everything the control thread wrote are visible; plan[(it+1) & 1] is free
        // the plan is the same in every lane: its fields go to scalar registers one by one (a struct copy would be
        // parked in scratch and reloaded piecemeal, each reload a full s_waitcnt)
        const W2Plan &pl_in = W.plan[it & 1u];
        const uint64_t pl_pp = (uint64_t)pl_in.pos;
        const int64_t plan_pos = (int64_t)(((uint64_t)__builtin_amdgcn_readfirstlane((uint32_t)(pl_pp >> 32)) << 32) | __builtin_amdgcn_readfirstlane((uint32_t)pl_pp));
        const int32_t plan_mode = __builtin_amdgcn_readfirstlane(pl_in.mode), plan_buf = __builtin_amdgcn_readfirstlane(pl_in.buf);
        const int32_t plan_resolve_prev = __builtin_amdgcn_readfirstlane(pl_in.resolve_prev), plan_n_win = __builtin_amdgcn_readfirstlane(pl_in.n_win);
        const int32_t plan_prev_n = __builtin_amdgcn_readfirstlane(pl_in.prev_n), plan_z = __builtin_amdgcn_readfirstlane(pl_in.zmode);
        W2Plan &next = W.plan[(it + 1u) & 1u];
        if (plan_mode == kPlanExit) break;
        const int64_t pos = plan_pos;
        const long long t_start = clock64();
        if (t0) {
            W2Stats &Q = W.stats;
            const int sidx = plan_mode == kPlanDetect ? 0 : plan_mode == kPlanSync ? 1 : plan_mode == kPlanSfd ? 2 : plan_mode == kPlanPause ? 3 : 5;
            if (Q.prev_state >= 0) { Q.cyc[Q.prev_state] += (uint32_t)((t_start - Q.prev_t) >> 6); Q.rounds[Q.prev_state]++; }
            Q.prev_state = sidx; Q.prev_t = t_start;
        }
        if (t0 && bal_mine) { // decision from the previous round's sample (that load has long returned), then the next sample
            W.prio = (bal_rem > bal_seen || (bal_rem == bal_seen && bal_slot != 0u)) ? 1u : 0u; // a tie goes to the younger workgroup
            const int64_t target = phase == 0 ? job.scan_limit + (job.probe_limit > job.scan_limit ? 12 * (int64_t)sps : 0) : pos + 6 * (int64_t)sps;
            const int64_t left = (target - pos) / (int64_t)sps;
            bal_rem = left < 1 ? 1u : (left > 0x3fffffff ? 0x3fffffffu : (uint32_t)left);
            __hip_atomic_store(bal_mine, bal_rem, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            bal_seen = __hip_atomic_load(bal_other, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        if (!is_ctl) { // (s_setprio takes an immediate)
            if (W.prio) __builtin_amdgcn_s_setprio(1); else __builtin_amdgcn_s_setprio(0);
        }
        const int64_t wpos = pos + (int64_t)wave * sps;

        if (plan_mode == kPlanDetect) { // every worker evaluates kd (1 or kW2DetectK) consecutive windows
            const int kd = plan_n_win;
            if (!is_ctl) {
                const int64_t dpos = pos + (int64_t)wave * kd * sps;
                int nvalid = 0; // windows of this worker that lie inside the data (:91)
                if (dpos + 2 * (int64_t)sps <= n_items) {
                    const int64_t fit = (n_items - dpos) / (int64_t)sps - 1;
                    nvalid = fit < kd ? (int)fit : kd;
                }
                float a[kW2DetectK][4];
                if (nvalid > 0) {
                    if (kd == 1) { const float4 r = w2_detect_window<GSF>(X + dpos); a[0][0] = r.x; a[0][1] = r.y; a[0][2] = r.z; a[0][3] = r.w; }
                    else w2_detect_windows<GSF, kW2DetectK>(X + dpos, nvalid, a);
                }
                if (lane == 0) {
                    W.detn[wave] = nvalid;
                    for (int i = 0; i < kW2DetectK; i++)
                        if (i < nvalid) { float *o = W.detf[wave * kW2DetectK + i]; o[0] = a[i][0]; o[1] = a[i][1]; o[2] = a[i][2]; o[3] = a[i][3]; }
                }
            }
            __syncthreads();
            if (is_ctl) {
                // the control wavefront evaluates all windows at once (lane q = window q); what the serial replay
                // of :740-768 would do with them -- stop at the first trigger, at the scan limit or at the end of
                // the data -- is then applied to the decoder state in one step
                constexpr int NWmax = kW2Workers * kW2DetectK;
                static_assert(NWmax <= 64, "one lane per DETECT window");
                const int NW = kW2Workers * kd;
                const int q = lane;
                const int qw = q / kd, qi = q - qw * kd; // worker and its window
                const bool valid = q < NW && qi < W.detn[qw];
                float autocorr = 0.0f, pushed = 0.0f, e2 = 0.0f;
                if (valid) {
                    const float *o = W.detf[qw * kW2DetectK + qi];
                    const float d0 = o[0], d1 = o[1], e1 = o[2];
                    e2 = o[3];
                    pushed = e1 / (float)sps; // :360
                    const float sq = sqrtf(e1 * e2);
                    autocorr = hypotf(d0 / sq, d1 / sq); // :363
                }
                const unsigned long long vmask = __ballot(valid), tmask = __ballot(valid && autocorr >= 0.90f); // :755
                W.red[q] = pushed; W.red[64 + q] = e2; W.red[128 + q] = autocorr;
                __builtin_amdgcn_wave_barrier();
                if (t0) {
                    W2State L = S; // register copy: every field access in LDS is a ~130-cycle round trip
                    // steps q >= 1 are preceded by the loop-top checks: in DETECT outside an attempt only the scan limit
                    // can change between steps (the data end shows up as an invalid window)
                    int64_t n_lim = (job.scan_limit - L.pos + (int64_t)sps - 1) / (int64_t)sps;
                    if (n_lim < 1) n_lim = 1;
                    const int n_valid = (~vmask == 0ull) ? 64 : __builtin_ctzll(~vmask);
                    int n_max = n_valid < NW ? n_valid : NW;
                    if ((int64_t)n_max > n_lim) n_max = (int)n_lim;
                    const int qt = tmask ? __builtin_ctzll(tmask) : 64;
                    const bool trig = qt < n_max;
                    const int nd = trig ? qt + 1 : n_max; // DETECT steps executed
                    if (nd > 0) {
                        if (trace) { // position-exact trace: one record per step
                            for (int i = 0; i < nd; i++) {
                                if (L.n_steps + (uint32_t)i < C.trace_cap) {
                                    StepRec &r = trace[L.n_steps + (uint32_t)i];
                                    r.state = kDetect; r.consumed = (trig && i == nd - 1) ? 0 : (int32_t)sps; r.pos = L.pos + (int64_t)i * sps; r.bin = -1; r.fine = 0;
                                    r.value = W.red[128 + i]; r.stream = job.stream_id; r.cycles = (uint32_t)(clock64() - t_start); r.pad = 0;
                                }
                            }
                        }
                        for (int i = (nd > 4 ? nd - 4 : 0); i < nd; i++) { // d_pwr_queue keeps the last 4 pushes (:360)
                            const float pv = W.red[i];
                            if (L.npush >= 4u) { L.push_tail[0] = L.push_tail[1]; L.push_tail[1] = L.push_tail[2]; L.push_tail[2] = L.push_tail[3]; L.push_tail[3] = pv; }
                            else { // (no dynamic index: that would put the whole copy in scratch)
                                const uint32_t k = L.npush;
                                if (k == 0u) L.push_tail[0] = pv; else if (k == 1u) L.push_tail[1] = pv; else if (k == 2u) L.push_tail[2] = pv; else L.push_tail[3] = pv;
                            }
                            L.npush++;
                        }
                        if (nd > 4) L.npush += (uint32_t)(nd - 4);
                        L.energy_threshold = W.red[64 + nd - 1] / 2.0f; // :357
                        L.n_steps += (uint32_t)nd;
                        L.pos += (int64_t)(trig ? nd - 1 : nd) * sps;
                        if (trig) {
                            L.corr_fails = 0u;
                            L.state = kSync;
                            L.in_attempt = 1;
                            L.att_trig = L.pos; L.att_hdr = -1; L.att_cr_prev = L.cr; L.att_ambig = 0; L.n_sym = 0;
                        }
                    }
                    W2Plan np;
                    plan_from(L, np);
                    next = np; S = L;
                }
            }
            continue;
        }

        if (plan_mode == kPlanSync) { // :770-783, detect_upchirp :392-413 -- all wavefronts together
            const float2 *__restrict__ x = X + pos;
            if (P.strict_sync) {
                w2_sync_exact_ifreq<GSF, WAVES>(x, f2);
            } else {
                constexpr int NI = (2 * SPS + kW2 - 1) / kW2; // samples per thread: all loads first, then the arithmetic
                float2 xb[NI], xa[NI];
#pragma unroll
                for (int k = 0; k < NI; k++) {
                    const uint32_t i = 1u + threadIdx.x + (uint32_t)k * kW2;
                    if (i < 2u * sps) { xb[k] = x[i - 1]; xa[k] = x[i]; }
                }
                __builtin_amdgcn_sched_group_barrier(0x020, 2 * NI, 0);
                __builtin_amdgcn_sched_group_barrier(0x002, 64 * NI, 0);
#pragma unroll
                for (int k = 0; k < NI; k++) {
                    const uint32_t i = 1u + threadIdx.x + (uint32_t)k * kW2;
                    if (i < 2u * sps) f2[i - 1] = ifreq_prod_z(xb[k], xa[k]); // (closed-form-only SYNC, once per packet: the zero-aware form directly)
                }
            }
            __syncthreads();
            if (t0) { f2[2u * sps - 1u] = f2[2u * sps - 2u]; strict::cands_reset(W.sc); } // :243
            __syncthreads();
            float bv, b2;
            int bi, i2;
            w2_sync_closed_form<GSF, WAVES>(f2, pre, P.sync_a, P.sync_b, bv, bi, b2, i2);
            const float my_bv = bv;
            const int my_bi = bi;
            w2_block_argmax_first<WAVES>(bv, bi, W.red);
            if (P.strict_sync) { // shifts within rounding of the maximum: the reference's own float sums decide (:399-407)
                strict::cands_push(W.sc, bv, my_bv, my_bi, b2, i2);
                __syncthreads();
                const int nc = W.sc.n;
                if (nc >= 2 && nc <= strict::kK) {
                    float ev;
                    if constexpr (SF == 7 && LD == 3) bi = strict::resolve_lds_inl<kW2>(f2, SPS, vl, &W.sc, reinterpret_cast<float *>(pre), 1024, &ev); // (inline: see resolve_lds)
                    else bi = strict::resolve_lds<kW2>(f2, SPS, vl, &W.sc, reinterpret_cast<float *>(pre), 1024, &ev); // (the closed form's prefix sums are done with: 4160 bytes; vl[k] = d_upchirp_ifreq[k], k < sps-1)
                    bv = ev;
                }
            }
            if (t0) {
                const int32_t consumed = (bi == 0x7fffffff) ? 0 : bi; // :771
                W2State L = S;
                L.state = kFindSfd;
                w2_end_step(L, job, C, recs, trace, kSync, consumed, -1, 0, bv, t_start);
                W2Plan np;
                plan_from(L, np);
                next = np; S = L;
            }
            if constexpr (ALIAS) { // the twiddle block back over SYNC's work areas (visible behind the next round's barrier)
                __syncthreads();
                const v2f *__restrict__ src = reinterpret_cast<const v2f *>(P.wave_tabs);
                for (uint32_t i = threadIdx.x; i < NENT; i += kW2) tab2[i] = src[i];
            }
            continue;
        }

        if (plan_mode == kPlanSfd) {
            // kW2SfdK windows per worker: worker w looks at the windows w, w + workers, ... of the round.  Behind SYNC lie ~6 upchirps,
            // the two sync-word symbols and the first downchirp - nine steps: two rounds of one window per worker (17 k clocks each)
            // or one round of two (30 k): FIND_SFD 104 k -> 89 k clocks per job, the SF7 walker +3 %, SF8 +1 %.  (A touch of the
            // second window's lines behind the first one's loads: nothing.)
            constexpr int kSfdWin = kW2SfdK * kW2Workers;
            static_assert(kW2SfdK <= 2 && kSfdWin <= 64, "results go to specf[w][k] / speci[k][w]");
#pragma unroll // (both evaluations inlined: the second window's loads are issued under the first one's arithmetic, +1 %)
            for (int k = 0; k < kW2SfdK; k++) {
                const int64_t kpos = wpos + (int64_t)k * kW2Workers * sps;
                const bool kvalid = !is_ctl && wave < plan_n_win && kpos + 2 * (int64_t)sps <= n_items;
                float c = 0.0f;
                int32_t fine = 0;
                int32_t pz = 0;
                if (kvalid) {
                    const W2SfdOut r = plan_z ? w2_sfd_window_zm<GSF, (4 << LD)>(X + kpos, T.v, T.dd, T.scratch + wave * 72, P.down_ifreq_sd, P.down_ifreq_dsum, P.sync_a, P.sync_b)
                                              : w2_sfd_window<GSF, false, (4 << LD)>(X + kpos, T.v, T.dd, T.scratch + wave * 72, P.down_ifreq_sd, P.down_ifreq_dsum, P.sync_a, P.sync_b);
                    c = r.c; fine = r.fine; pz = r.pz;
                }
                if (lane == 0 && !is_ctl) { W.specf[wave][k] = c; W.speci[k][wave][0] = kvalid ? 1 : 0; W.speci[k][wave][1] = fine; W.speci[k][wave][2] = pz; }
            }
            __syncthreads();
            if (is_ctl) { // the whole control wavefront, uniformly, on a register copy of the state (as the decode rounds do)
                W2State L = S;
                const int qw = lane % kW2Workers, qk = lane / kW2Workers; // lane q fetches window q = qk * workers + qw
                const float my_c = lane < kSfdWin ? W.specf[qw][qk] : 0.0f;
                const int32_t my_v = lane < kSfdWin ? W.speci[qk][qw][0] : 0, my_f = lane < kSfdWin ? W.speci[qk][qw][1] : 0;
                const int32_t my_p = lane < kSfdWin ? W.speci[qk][qw][2] : 0;
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                bool zreq = false;
                for (int w = 0; w < kSfdWin; w++) {
                    if (w > 0 && !w2_pre_step(L, job, rec_cap, sps)) break;
                    if (!__builtin_amdgcn_readlane(my_v, w)) break;
                    if (__builtin_amdgcn_readlane(my_p, w)) { zreq = true; break; } // a sample of exactly zero in this window: it opens a round of ZM evaluations
                    const float cw = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int32_t, my_c), w));
                    int32_t fw = 0;
                    if (cw > 0.96f) { // :792
                        L.state = kPause;
                    } else {
                        if (cw < -0.97f) fw = __builtin_amdgcn_readlane(my_f, w); // :801-803
                        else L.corr_fails++;
                        if (L.corr_fails > 4u) L.state = kDetect; // :808-809
                    }
                    w2_end_step(L, job, C, recs, trace, kFindSfd, (int32_t)sps + fw, -1, fw, cw, t_start, t0);
                    if (L.state != kFindSfd || L.done || fw != 0) break;
                }
                // PAUSE (:820-824) looks at no sample: the step is taken here, behind its own loop-top checks, instead of
                // in a round of its own (a barrier, a plan hand-over and ~4 k clocks per packet)
                if (L.state == kPause && !L.done && w2_pre_step(L, job, rec_cap, sps)) {
                    L.state = kDecodeHeader;
                    const int32_t consumed = (int32_t)(sps + sps / 4u);
                    L.att_hdr = L.pos + consumed;
                    w2_end_step(L, job, C, recs, trace, kPause, consumed, -1, 0, 0.0f, t_start, t0);
                }
                W2Plan np;
                plan_from(L, np, zreq);
                if (t0) { next = np; S = L; }
            }
            continue;
        }

        if (plan_mode == kPlanPause) { // :820-824
            if (t0) {
                S.state = kDecodeHeader;
                const int32_t consumed = (int32_t)(sps + sps / 4u);
                S.att_hdr = S.pos + consumed;
                w2_end_step(S, job, C, recs, trace, kPause, consumed, -1, 0, 0.0f, t_start);
                plan_from_state(next);
            }
            continue;
        }

        if (plan_mode == kPlanFinalize) { // decode(false) + frame bytes (:870-881), all threads
            const uint32_t n_cw = S.n_cw, cr = S.cr, n_bytes = S.fin_n_bytes, plen = S.fin_plen;
            decode_payload_bytes(sh, n_cw, cr, n_bytes);
            AttemptRec &r = recs[S.n_att];
            for (uint32_t i = threadIdx.x; i < plen; i += kW2) r.frame[3u + i] = (i < n_bytes) ? sh.dec[i] : 0;
            __syncthreads();
            if (t0) {
                r.frame[0] = S.phdr[0]; r.frame[1] = S.phdr[1]; r.frame[2] = S.phdr[2]; // d_phdr (:600)
                r.frame_len = 3u + plen;
                S.frame_ok = 1;
                S.state = kDetect;
                S.n_words = 0; S.n_cw = 0;
                S.fin_pending = 0;
                w2_end_step(S, job, C, recs, trace, S.fin_st, S.fin_consumed, S.fin_bin, S.fin_fine, 0.0f, t_start);
                plan_from_state(next);
            }
            continue;
        }

        // ---- kPlanDecode: DECODE_HEADER / DECODE_PAYLOAD rounds (:826-886), pipelined.  Workers demodulate the
        // 7 symbols at plan_pos into spec[plan_buf]; concurrently the control thread resolves the previous round
        // (spec[plan_buf ^ 1]) and decides whether this round's position was predicted correctly.
        // The symbol clock moves when a window's fine sync is non-zero (:506-512, consumed = sps + fine): the windows after
        // it in that round started at the wrong sample, and so would this round, planned while that one was still being
        // computed.  The workers' fine values are in LDS by now, so every wavefront re-aims this round itself: it restarts
        // right after the first such window of the previous round, shifted by its fine value.  (Whether that is the true
        // continuation is still decided by the control wavefront's resolve, below.)
        int64_t dpos = pos;
        int32_t dn = plan_n_win;
        if (plan_resolve_prev && plan_prev_n > 0) {
            const int rb = plan_buf ^ 1;
            const int32_t s_l = lane < plan_prev_n ? W.speci[rb][lane][0] : -1, f_l = lane < plan_prev_n ? W.speci[rb][lane][1] : 0;
            const unsigned long long moved = __ballot(s_l >= 0 && f_l != 0);
            if (moved) {
                const int wq = __builtin_ctzll(moved);
                const int32_t redo = plan_prev_n - (wq + 1); // windows of the previous round to be demodulated again
                dpos = pos - (int64_t)redo * sps + (int64_t)__builtin_amdgcn_readlane(f_l, wq);
                dn = plan_n_win + redo < kW2Win ? plan_n_win + redo : kW2Win;
            }
        }
        if (!is_ctl) {
            { // window `wave`
                const int widx = wave;
                const int64_t dwpos = dpos + (int64_t)widx * sps;
                const bool dvalid = widx < dn && dwpos + 2 * (int64_t)sps <= n_items;
                uint32_t ws = 0;
                int32_t wfine = 0;
                float wen = 0.0f;
                if (dvalid) {
                    if (plan_z) { // (uniform) a round of ZM evaluations: a window of the previous round holds a sample of exactly zero
                        const W2DemodZ z = w2_demod_zm<SF, GRAD, LD>(P.enable_fine_sync, P.demod_mode, P.implicit != 0u, FT, X + dwpos);
                        ws = z.s; wfine = z.fine; wen = z.en;
                    } else
                    if constexpr (LD != 3 || SF < 7) { // (lora_wave_decim.inc.hip: decimation 2 / 4, and SF6)
                        if constexpr (GRAD) wave_demod_symbol_grad_d<SF, LD>(P, FT.v, X + dwpos, P.implicit != 0u, ws, wfine, wen);
                        else wave_demod_symbol_d<SF, LD>(P, FT, X + dwpos, ws, wfine, P.implicit != 0u ? &wen : nullptr);
                    } else
                    if constexpr (GRAD) wave_demod_symbol_grad<SF>(P, FT.v, X + dwpos, P.implicit != 0u, ws, wfine, wen); // ws = bin_idx itself; kPoisonBin: see W2Plan.zmode
                    else wave_demod_symbol<SF, kWaveFmode<SF>>(P, FT, X + dwpos, ws, wfine, P.implicit != 0u ? &wen : nullptr);
                }
                if (lane == 0) { W.speci[plan_buf][widx][0] = dvalid ? (int32_t)ws : -1; W.speci[plan_buf][widx][1] = wfine; W.speci[plan_buf][widx][2] = __builtin_bit_cast(int32_t, wen); }
            }
        } else { // the whole control wavefront, uniformly (identical values in every lane); t0 does the stores
            bool predicted = true, zreq = false;
            const long long tr0 = clock64();
            constexpr bool W16 = SF > 8; // (SF9: decimation 2 / 4 only - at 8 it is walker3's)
            uint32_t wpk[W16 ? 4 : 2];
            if constexpr (W16) { wpk[0] = W.words_pk16[0]; wpk[1] = W.words_pk16[1]; wpk[2] = W.words_pk16[2]; wpk[3] = W.words_pk16[3]; }
            else { wpk[0] = W.words_pk[0]; wpk[1] = W.words_pk[1]; }
            W2State L = S; // the resolve works on a register copy: every field access in LDS is a ~130-cycle round trip
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            const long long tr1 = clock64();
            if (plan_resolve_prev) {
                const int rb = plan_buf ^ 1;
                // lane w fetches worker w's result: one LDS round trip for the round instead of two per symbol
                const int32_t my_s = lane < kW2Win ? W.speci[rb][lane][0] : -1, my_f = lane < kW2Win ? W.speci[rb][lane][1] : 0;
                const int32_t my_e = (P.implicit != 0u && lane < kW2Win) ? W.speci[rb][lane][2] : 0;
                for (int w = 0; w < kW2Win; w++) {
                    if (w > 0 && !w2_pre_step(L, job, rec_cap, sps)) break;
                    const int32_t sw = __builtin_amdgcn_readlane(my_s, w);
                    int32_t fw = __builtin_amdgcn_readlane(my_f, w);
                    if (!(L.state == kDecodeHeader || L.state == kDecodePayload)) break;
                    if (sw == (int32_t)kPoisonBin) { zreq = true; break; } // a sample of exactly zero in this window: it opens a round of ZM evaluations
                    if (sw < 0) break;
                    const bool is_first = L.state == kDecodeHeader;
                    const int32_t st_w = L.state;
                    const uint32_t sres = (uint32_t)sw;
                    uint32_t bin_idx = GRAD ? sres : ((sres == 0u && P.demod_mode == 2u) ? 0u : (sres + (uint32_t)N - 1u) % (uint32_t)N);
                    int32_t step_bin = (int32_t)bin_idx;
                    bool do_demod = true;
                    if (P.implicit != 0u && !is_first) { // determine_energy (:861-864, :368-375)
                        const float ew = __builtin_bit_cast(float, __builtin_amdgcn_readlane(my_e, w));
                        if (ew < L.energy_threshold) { L.payload_symbols = 0; L.payload_length = L.n_cw / 2u; do_demod = false; bin_idx = 0u; step_bin = -1; fw = 0; }
                    }
                    if (w2_post_symbol<true, W16>(P, L, sh, bin_idx, is_first, wpk, do_demod)) { // payload complete: finalise with all threads
                        L.fin_pending = 1; L.fin_st = st_w; L.fin_consumed = (int32_t)sps + fw; L.fin_bin = step_bin; L.fin_fine = fw;
                        break;
                    }
                    w2_end_step(L, job, C, recs, trace, st_w, (int32_t)sps + fw, step_bin, fw, 0.0f, t_start, t0);
                    if constexpr (SKIP) {
                        if (is_first && L.state == kDecodePayload && !L.done) { // header parsed (:831-847): the payload is the payload pass's
                            AttemptRec &r = recs[L.n_att];
                            if (t0) {
                                SkippedPayload sk;
                                sk.phdr[0] = L.phdr[0]; sk.phdr[1] = L.phdr[1]; sk.phdr[2] = L.phdr[2];
                                sk.n_left = (uint8_t)(L.n_cw < 8u ? L.n_cw : 8u);
                                for (uint32_t i = 0; i < 8u; i++) sk.left[i] = i < L.n_cw ? sh.cw[i] : (uint8_t)0;
                                sk.payload_symbols = L.payload_symbols;
                                *reinterpret_cast<SkippedPayload *>(r.frame) = sk;
                            }
                            const int32_t n_walk = L.payload_symbols > 0 ? L.payload_symbols : 1; // (:866-870 are reached behind the first symbol at the earliest)
                            L.state = kDetect; L.frame_ok = 0; L.n_words = 0; L.n_cw = 0;
                            w2_end_step(L, job, C, recs, nullptr, kDecodePayload, n_walk * (int32_t)sps, -1, 0, 0.0f, t_start, t0); // (the attempt is closed: status, positions, pushes)
                            if (t0) { r.status = kAttemptHeaderOnly; r.frame_len = (uint32_t)sizeof(SkippedPayload); }
                            break;
                        }
                    }
                    if (L.done || fw != 0) break; // later windows started at the wrong sample
                }
                // is the round the workers are computing right now the true continuation?
                predicted = !zreq && !L.done && !L.fin_pending && (L.state == kDecodeHeader || L.state == kDecodePayload) && L.pos == dpos &&
                            w2_pre_step(L, job, rec_cap, sps);
            }
            const long long tr2 = clock64();
            W2Plan np; // (built in registers, stored by t0)
            if (predicted) {
                // the round in flight continues the packet; how much of the packet is left after it?
                int32_t n_next = kW2Win;
                if (L.state == kDecodePayload && !P.implicit) {
                    const int32_t rem = L.payload_symbols - (int32_t)L.n_words - dn;
                    n_next = rem < kW2Win ? (rem > 0 ? rem : 0) : kW2Win; // 0: nothing left to demodulate, only resolve
                }
                np.mode = kPlanDecode; np.pos = dpos + (int64_t)dn * sps; np.buf = plan_buf ^ 1; np.resolve_prev = 1; np.n_win = n_next;
                np.prev_n = dn; np.zmode = 0;
            } else {
                plan_from(L, np, zreq); // this round's results are discarded
            }
            const long long tr3 = clock64();
            if (t0) {
                next = np;
                S = L;
                if constexpr (W16) { W.words_pk16[0] = wpk[0]; W.words_pk16[1] = wpk[1]; W.words_pk16[2] = wpk[2]; W.words_pk16[3] = wpk[3]; }
                else { W.words_pk[0] = wpk[0]; W.words_pk[1] = wpk[1]; }
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                W.stats.ctl[0] += (uint32_t)((tr1 - tr0) >> 6); W.stats.ctl[1] += (uint32_t)((tr2 - tr1) >> 6); W.stats.ctl[2] += (uint32_t)((tr3 - tr2) >> 6);
                W.stats.ctl[3] += (uint32_t)((clock64() - tr3) >> 6);
                W.stats.cyc[4] += (uint32_t)((clock64() - tr0) >> 6); W.stats.rounds[4]++; // control wavefront's share of a decode round
            }
        }
    }

    // an attempt cut short (out of data, or probe stop) is reported but not counted as complete
    __syncthreads();
    if (t0) {
        const bool in_attempt = S.in_attempt != 0;
        if (in_attempt && S.n_att < rec_cap) {
            AttemptRec &r = recs[S.n_att];
            r.status = (S.stop_reason == 3) ? kAttemptAtHeader : kAttemptOutOfData;
            r.start_pos = S.att_start; r.trig_pos = S.att_trig; r.hdr_pos = S.att_hdr; r.end_pos = S.pos;
            r.npush = S.npush;
            for (int i = 0; i < 4; i++) r.push_tail[i] = S.push_tail[i];
            r.cr_prev = S.att_cr_prev; r.hdr_ambig = S.att_ambig; r.n_symbols = S.n_sym; r.frame_len = 0; r.n_sfd = 0;
        }
        JobResult &jr = C.results[jid];
        const int64_t e_pos = in_attempt ? S.att_start : S.pos;
        const uint32_t e_natt = S.n_att + (in_attempt ? 1u : 0u), e_npush = in_attempt ? 0u : S.npush, e_pad = in_attempt ? 1u : 0u;
        bool go = false;
        if (phase == 0) {
            jr.final_pos = e_pos;
            jr.n_attempts = e_natt;
            jr.final_cr = S.cr;
            jr.npush = e_npush;
            for (int i = 0; i < 4; i++) jr.push_tail[i] = S.push_tail[i];
            jr.stop_reason = (uint32_t)S.stop_reason;
            jr.n_steps = S.n_steps < C.trace_cap ? S.n_steps : C.trace_cap;
            jr.pad = e_pad;
            jr.tail_valid = 0;
            go = job.probe_limit > job.scan_limit && S.stop_reason == 0 && !in_attempt && !trace && e_natt < C.recs_per_job;
            W.ph_go = go ? 1u : 0u; W.ph_start = e_pos; W.ph_cr = S.cr; W.ph_natt = e_natt; W.ph_pad = 0;
        } else {
            jr.tail_valid = 1; jr.tail_first_rec = W.ph_natt;
            jr.tail_final_pos = e_pos; jr.tail_n_attempts = e_natt; jr.tail_final_cr = S.cr; jr.tail_npush = e_npush;
            for (int i = 0; i < 4; i++) jr.tail_push_tail[i] = S.push_tail[i];
            jr.tail_stop_reason = (uint32_t)S.stop_reason; jr.tail_pad = e_pad; jr.tail_rsv = 0;
        }
        if (!go) { // last phase of this job: the per-state time accounting goes out with it
            W2Stats &Q = W.stats;
            if (Q.prev_state >= 0) { Q.cyc[Q.prev_state] += (uint32_t)((clock64() - Q.prev_t) >> 6); Q.rounds[Q.prev_state]++; }
            for (int i = 0; i < 6; i++) { jr.cyc[i] = Q.cyc[i]; jr.rounds[i] = Q.rounds[i]; }
            for (int i = 0; i < 4; i++) jr.ctl[i] = Q.ctl[i];
            jr.dbg[0] = __builtin_amdgcn_s_getreg((31 << 11) | 4); jr.dbg[1] = __builtin_amdgcn_s_getreg((31 << 11) | 20);
            jr.dbg[2] = dbg_t0; jr.dbg[3] = (uint32_t)__builtin_amdgcn_s_memrealtime();
            jr.dbg[4] = dbg_c0; jr.dbg[5] = (uint32_t)(clock64() >> 6);
        }
    }
    if (phase == 1) break;
    __syncthreads();
    if (!W.ph_go) break;
    bal_rem = 0xffffffffu;
    // the probe: a job from the end state of the job proper to the next header (what decode_streams would launch)
    { // wave-uniform values: keep them in SGPRs
        const int64_t st = W.ph_start;
        const uint32_t lo = __builtin_amdgcn_readfirstlane((uint32_t)st), hi = __builtin_amdgcn_readfirstlane((uint32_t)((uint64_t)st >> 32));
        const uint32_t natt = __builtin_amdgcn_readfirstlane(W.ph_natt);
        job.start = (int64_t)(((uint64_t)hi << 32) | lo);
        job.cr_prev = __builtin_amdgcn_readfirstlane(W.ph_cr);
        job.scan_limit = job.probe_limit; job.stop_at_header = 1; job.max_attempts = 0;
        recs += natt;
        rec_cap = C.recs_per_job - natt;
    }
    __syncthreads(); // everyone has read the hand-over before the control thread re-initialises the state
    } // phase
    if (t0 && bal_mine) __hip_atomic_store(bal_mine, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); // nothing left: the neighbour keeps the CU
}

constexpr int kW2WavesSf7 = LORA_W2_WAVES_SF7, kW2WavesSf8 = LORA_W2_WAVES_SF8;
__global__ __launch_bounds__(64 * kW2WavesSf7, 4) void walker2_kernel_sf7(DevParams P, LaunchCfg C) { walker2_body<7, kW2WavesSf7, false>(P, C); }
__global__ __launch_bounds__(64 * kW2WavesSf8, LORA_W2_EU_SF8) void walker2_kernel_sf8(DevParams P, LaunchCfg C) { walker2_body<8, kW2WavesSf8, false>(P, C); }
// the same body at the 256-register budget (one workgroup per CU by registers): what launch_walker picks for a launch with no more jobs than CUs, where a
// workgroup has its CU to itself anyway - 0.161 against 0.138 of HBM peak at 256 packets; with more jobs the two-per-CU build above wins, 0.251 against 0.211 at 1024
__global__ __launch_bounds__(64 * kW2WavesSf8, 2) void walker2_kernel_sf8_wide(DevParams P, LaunchCfg C) { walker2_body<8, kW2WavesSf8, false>(P, C); }
// ... and SF7's (round 6: 192 registers, nothing spilled; 256 packets 125 -> 138 Gsamples/s), and the gradient kernels'
__global__ __launch_bounds__(64 * kW2WavesSf7, 2) void walker2_kernel_sf7_wide(DevParams P, LaunchCfg C) { walker2_body<7, kW2WavesSf7, false>(P, C); }
__global__ __launch_bounds__(64 * kW2WavesSf7, 2) void walker2_kernel_sf7_grad_wide(DevParams P, LaunchCfg C) { walker2_body<7, kW2WavesSf7, true>(P, C); }
__global__ __launch_bounds__(64 * kW2WavesSf8, 2) void walker2_kernel_sf8_grad_wide(DevParams P, LaunchCfg C) { walker2_body<8, kW2WavesSf8, true>(P, C); }
// the gradient demodulator (demod_mode 0, the reference's default): no FFT tables, fewer live registers
#ifndef LORA_W2_EU_GRAD_SF7
#define LORA_W2_EU_GRAD_SF7 4
#endif
#ifndef LORA_W2_EU_GRAD_SF8
#define LORA_W2_EU_GRAD_SF8 4
#endif
__global__ __launch_bounds__(64 * kW2WavesSf7, LORA_W2_EU_GRAD_SF7) void walker2_kernel_sf7_grad(DevParams P, LaunchCfg C) { walker2_body<7, kW2WavesSf7, true>(P, C); }
__global__ __launch_bounds__(64 * kW2WavesSf8, LORA_W2_EU_GRAD_SF8) void walker2_kernel_sf8_grad(DevParams P, LaunchCfg C) { walker2_body<8, kW2WavesSf8, true>(P, C); }
// the header-only variants of a decoupled pass (SKIP); a decoupled pass has no more jobs than CUs: SF8 at the 256-register budget
__global__ __launch_bounds__(64 * kW2WavesSf7, 4) void walker2_kernel_sf7_skip(DevParams P, LaunchCfg C) { walker2_body<7, kW2WavesSf7, false, true>(P, C); }
__global__ __launch_bounds__(64 * kW2WavesSf8, 2) void walker2_kernel_sf8_skip(DevParams P, LaunchCfg C) { walker2_body<8, kW2WavesSf8, false, true>(P, C); }
__global__ __launch_bounds__(64 * kW2WavesSf7, LORA_W2_EU_GRAD_SF7) void walker2_kernel_sf7_grad_skip(DevParams P, LaunchCfg C) { walker2_body<7, kW2WavesSf7, true, true>(P, C); }
__global__ __launch_bounds__(64 * kW2WavesSf8, LORA_W2_EU_GRAD_SF8) void walker2_kernel_sf8_grad_skip(DevParams P, LaunchCfg C) { walker2_body<8, kW2WavesSf8, true, true>(P, C); }

// decimation 4 / 2 (lora_wave_decim.inc.hip): 128 registers, two workgroups per CU
// (SF9 at decimation 4: 2048-sample windows, 32 samples per lane, 95 KB of LDS - one workgroup per CU at 256 registers)
#define LORA_W2_DECIM_KERNEL(SFV, DV, LDV, EU) \
    __global__ __launch_bounds__(512, EU) void walker2_kernel_sf##SFV##_d##DV(DevParams P, LaunchCfg C) { walker2_body<SFV, 8, false, false, LDV>(P, C); } \
    __global__ __launch_bounds__(512, EU) void walker2_kernel_sf##SFV##_d##DV##_grad(DevParams P, LaunchCfg C) { walker2_body<SFV, 8, true, false, LDV>(P, C); }
LORA_W2_DECIM_KERNEL(7, 2, 1, 4)
LORA_W2_DECIM_KERNEL(7, 4, 2, 4)
LORA_W2_DECIM_KERNEL(8, 2, 1, 4)
LORA_W2_DECIM_KERNEL(8, 4, 2, 4)
LORA_W2_DECIM_KERNEL(9, 2, 1, 4)
LORA_W2_DECIM_KERNEL(9, 4, 2, 2)
LORA_W2_DECIM_KERNEL(6, 4, 2, 4)
LORA_W2_DECIM_KERNEL(6, 8, 3, 4) // SF6 at decimation 8: the same body on lora_wave_decim.inc.hip's demodulators (LD = 3)
#undef LORA_W2_DECIM_KERNEL
static bool walker2_decim_covers(uint32_t sf, uint32_t decim) { return ((decim == 2u || decim == 4u) && sf >= 7u && sf <= 9u) || (sf == 6u && (decim == 4u || decim == 8u)); }

static uint32_t walker2_threads(uint32_t sf) { return 64u * (uint32_t)(sf == 7u ? kW2WavesSf7 : kW2WavesSf8); }

static uint32_t walker2_lds_bytes(uint32_t sf, bool grad = false, uint32_t decim = 8u)
{
    const uint32_t sps = decim << sf;
    const uint32_t nv = (3u * sps + 40u + 3u) & ~3u;
    if (decim != 8u || sf < 7u)
        return (uint32_t)((sizeof(W2Shared) + 15) & ~(size_t)15) + (2u * sps + nv + sps) * (uint32_t)sizeof(float) +
               (grad ? 0u : wave_tables_floats_d(sf, decim) * (uint32_t)sizeof(float)) + (2u * 256u + 8u) * (uint32_t)sizeof(double);
    if (!grad && sf == 8u && kW2Alias<8>) // (SYNC's work areas inside the table block)
        return (uint32_t)((sizeof(W2Shared) + 15) & ~(size_t)15) + (nv + sps) * (uint32_t)sizeof(float) + wave_tables_floats(sf) * (uint32_t)sizeof(float);
    return (uint32_t)((sizeof(W2Shared) + 15) & ~(size_t)15) + (2u * sps + nv + sps) * (uint32_t)sizeof(float) +
           (grad ? 0u : wave_tables_floats(sf) * (uint32_t)sizeof(float)) + (2u * 256u + 8u) * (uint32_t)sizeof(double);
}
