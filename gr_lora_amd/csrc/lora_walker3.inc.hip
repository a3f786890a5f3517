// lora_walker3.inc.hip -- the SF9 .. SF12 walker (decimation 8): one workgroup per job; the FFT demodulators of SF10 .. SF12 demodulate
// a symbol window cooperatively (below), SF9's and every gradient demodulator one window per WAVEFRONT.  Included by lora_kernels.hip.
//
// SF9 (round 5): a 4096-sample window is 64 samples per lane - wave_demod_symbol<9> of lora_wave_demod.inc.hip holds it in registers, and a
// decode round is eight independent wavefronts (W3Geom::WFFT: the LDS array then holds 16 KB of acquisition scratch and the wave
// demodulator's tables).  What follows describes the cooperative form, which SF9 keeps only for its acquisition rounds' geometry.
//
// A symbol is 8 N samples = 32 .. 256 KB: too much for one wavefront's registers (the SF7 / SF8 scheme), and the
// generic kernel's radix-2 stages in LDS (one barrier per stage, twiddles and tables from global memory) leave the
// CU idle most of the time.  Here the pruned sps-point DFT of get_shift_fft (lib/decoder_impl.cc:430-464),
//     X[k] = sum_{r<8} W_sps^{k r} Y_r[k mod N],   Y_r = FFT_N(x[8q+r] down[8q+r]),
// is three register passes with two LDS transposes between them:
//   pass 1  thread (q0, r) loads the 16 samples n = c sps/16 + t (c < 16; t = 8 q0 + r: every load instruction
//           covers 512 contiguous bytes), i.e. q = q0 + (N/16) c of polyphase branch r: dechirp, radix-16 DIF in
//           registers (output register m holds a = bitrev4(m)), twiddle W_N^{q0 a}, store row m of the LDS array;
//   pass 2  N/16 = 16 M2: thread (row, q1, r) reads the 16 entries q0 = q1 + M2 c2 of its row, radix-16 (a2),
//           twiddle W_{N/16}^{q1 a2} (wave-uniform), stores in place;
//   pass 3  radix-M2 (2, 4, 8, 16 for SF9 .. 12) over q1, 16 / M2 butterflies per thread -> Y_r[a + 16 a2 + 256 b2];
//           times the combine coefficient W_sps^{k r} (+ the reference's fold tmp[N/2] += F[N/2], :450; a table in
//           pass-3 thread order, read coalesced), summed over r = lane bits 0-2 with DPP adds; |X|^2 arg-max, first
//           maximum in bin order (:454-463).
// SF12 (256 KB per symbol against 160 KB of LDS) runs passes 2 and 3 twice: the radix-16 outputs of pass 1 are 16
// independent sub-problems (bins k = a mod 16); rows m < 8 go to LDS first, rows m >= 8 wait in registers.
// The LDS array is addressed e = row (8 N/16 + 8) + pos 8 + r (8-byte entries): rows 64 bytes apart modulo the 256
// bytes of the bank space, so that the 32 lanes of a ds_read_b64 (8 r x 4 rows) and the 16 lanes of a ds_write_b64
// (8 r x 2 rows) always cover distinct banks.
//
// fine_sync (:300-338) for the three lags of a payload symbol uses the window's instantaneous frequency, computed from
// the samples while they are in registers for the dechirp (x[n-1] by a second buffer load one item down: the same
// cache lines); at SF12, where the sixteen samples of both pairs do not fit beside the held rows, from a second read of
// the window behind the arg-max (LATE_F).
//
// A workgroup is 1024 threads = NG groups of TG threads (SF9: 4 x 256, SF10: 2 x 512, SF11 / SF12: 1 x 1024), one
// workgroup per CU; every group demodulates ONE symbol window, so a round evaluates NG consecutive windows speculatively
// (zero drift assumed), as walker2's workers do.  The state machine is the reference's work() (:740-903): the decoder
// state lives in LDS (W3Shared::st) and thread 0 replays the round's results over a register copy of it, in order,
// stopping at the first window whose outcome invalidates the later ones (a state change, d_fine_sync != 0 :321,:856,
// :883, the scan limit, the end of the data); every thread reads the next round's plan from LDS (double-buffered) into
// scalar registers.  DETECT (one window per group), SYNC (closed form over block-wide prefix sums in double, all 1024
// threads), FIND_SFD (one-pass Pearson + closed-form 63-lag fine_sync per group) and the finalisation of a frame are
// rounds of the same loop; a job that reaches its scan limit carries on as the next segment's tail probe (second phase,
// as in walker2).

// Geometry decisions that were build-time switches while they were being measured (same-box A/B grids: profiles/r02_*, r03_ab_*, docs/LAB_NOTEBOOK.md) and
// are now fixed: 512-thread workgroups at a 256-register budget (every thread does the work of two: +4 ... 28 % against 1024 x 128); the job record in
// scalar registers (uniform_job); the replay's power-of-two reductions as masks; the replay's short path for a full round of unmoved payload symbols;
// fine_sync's ifreq kept from pass 1 except at SF12 (LATE_F: its eight held rows leave no registers for it).  Dropped after measurement: a register
// prefetch of the next pair in pass 1, pass 1 as a load pipeline, acquisition rounds over several windows at K > 1 (kept only for the header-only variants).
#ifndef LORA_W3_P1_PREFETCH
#define LORA_W3_P1_PREFETCH 0   // pass 1 of w3_demod_round as a load pipeline (see there): built and measured in round 6 as in round 3 - off
#endif
#ifndef LORA_W3_REPLAY_STATS
#define LORA_W3_REPLAY_STATS 1  // LORA_HIP_DEBUG accounting of thread 0's replay inside the decode rounds (2: finer)
#endif

template <int SF, int HV = 0> struct W3Geom {
    static constexpr int N = 1 << SF, SPS = 8 * N;
    static constexpr bool T512 = true;                  // 512 threads x 256 registers (the 1024 x 128 geometry of round 2 is what T512 = false still describes)
    // HV = 1 (SF9 / SF10 only): HALF the workgroup - half the groups, half the threads, the same work per thread and the same geometry per group - so
    // that TWO workgroups share a CU (79 / 78 KB of LDS each) and one's rounds fill the other's waits (docs/LAB_NOTEBOOK.md 5.4: +12 % at SF9, +4.5 % at SF10 when a
    // launch holds at least two jobs per CU; with one job per CU it would walk its packet at half the width - the launcher picks by job count)
    static_assert(HV == 0 || (HV == 1 && SF <= 10), "half-size workgroups exist for SF9 and SF10");
    static constexpr int T = (T512 ? 512 : 1024) >> HV; // threads per workgroup (16 or 8 wavefronts, one workgroup per CU; HV: 4, two per CU)
    static constexpr int NG = (SF == 9 ? 4 : SF == 10 ? 2 : 1) >> HV; // groups: windows evaluated per round
    static constexpr int TG = T / NG;                   // threads of one group = one symbol window
    static constexpr int VT = SF >= 11 ? 1024 : N / 2;  // "virtual threads" of a group: the units of passes 2 and 3 (what TG is at T = 1024)
    static constexpr int U = VT / TG;                   // units per thread in passes 2 and 3 (1; T512: 2)
    static constexpr int GW = TG / 64;                  // wavefronts per group
    static constexpr int PAIRS = SPS / (16 * TG);       // (q0, r) pairs per thread in pass 1: 1; SF12: 2; T512: 2 / 2 / 2 / 4
    static constexpr int ROUNDS = SF == 12 ? 2 : 1;     // passes over the rows of pass 1 (SF12: a symbol is 256 KB against 160 KB of LDS)
    static constexpr int AR = 16 / ROUNDS;              // rows resident in LDS per round
    static constexpr int M = N / 16, M2 = M / 16, LOGM2 = ilog2(M2);
    static constexpr int SA = 8 * M + 8;                // entries per row
    // SF9, full-size workgroup (round 5): the FFT demodulator is wave_demod_symbol<9> - one window per WAVEFRONT, everything in registers (lora_wave_demod.inc.hip) -
    // and the LDS array holds [16 KB of acquisition scratch | its tables]; passes 1-3 below are then SF10-SF12's (and the half-size SF9 workgroup's)
    static constexpr bool WFFT = SF == 9 && HV == 0;
    static constexpr int NTW = WFFT ? 0 : (N > 2048 || (HV && SF == 10)) ? N / 2 : N; // W_N^t entries kept in LDS (SF12, and SF10's half-size workgroup - 82 080 -> 77 984 B -: half, the rest by sign)
    static constexpr int STRICT_CH = WFFT ? 512 : 2048; // strict SYNC's chunk unit (strict::resolve; two buffers of kK x CH floats at the head of the LDS array: 16 KB / 64 KB)
    static constexpr int CH = SPS / 16;                 // samples between a thread's consecutive loads
    static constexpr int NWL = VT / (8 * AR), NB = 16 / NWL; // pass 3: butterflies per unit
    static constexpr int LEN = SPS / T;                 // SYNC: shifts per thread (all groups together)
    static constexpr bool LATE_F = SF == 12;            // fine_sync's ifreq from a second read of the window (SF12: 64 more live registers otherwise)
    static constexpr bool UNIFORM_JOB = true;           // the job record through readfirstlane (uniform_job)
    static constexpr bool FAST_MOD = true;              // power-of-two reductions of the replay as masks
    static constexpr uint32_t kWfftScratch = 16384u;
    static constexpr uint32_t data_entries = WFFT ? (kWfftScratch + kWaveLdsBytes<9>) / 8u / (uint32_t)NG : (uint32_t)AR * SA; // per group
    static_assert(!WFFT || (kWfftScratch + kWaveLdsBytes<9>) % (8u * (uint32_t)NG) == 0u, "the SF9 table block in whole entries");
    static_assert(NB * M2 == 16, "pass 3 covers 16 values per unit");
    static_assert(AR * M2 * 8 == VT, "pass 2 uses every unit of the group once per round");
    static_assert(U * TG == VT && U == PAIRS / (SF == 12 ? 2 : 1) && TG % 64 == 0, "geometry");
};

struct alignas(16) W3Shared {
    float    red[2][4][72];        // group reductions, double-buffered (one barrier each): [slot][group][wave * K + k]
    double   dred[2][72];          // block scans / sums of doubles (SYNC; FIND_SFD at [group * 18 + ..])
    float    sfd[4][72];           // FIND_SFD: head / tail samples of the window's ifreq, per group
    int32_t  ibc[8];               // broadcasts (FIND_SFD lag per group)
    int32_t  wres[16][4];          // gradient decode rounds: (bin, d_fine_sync, energy, valid) of the window each wavefront demodulated
    W2Plan   plan[2];              // the round plan, double-buffered
    W2State  st;                   // decoder state: thread 0 only
    strict::Cands sc;              // SYNC: near-tied shifts for the exact re-evaluation
    W2Stats  stats;                // per-state time accounting (LORA_HIP_DEBUG)
    int64_t  ph_start;             // hand-over from the job proper to its tail probe (Job.probe_limit)
    uint32_t ph_go, ph_cr, ph_natt, ph_pad;
    Shared   sh;                   // integer chain (words / codewords / decoded bytes)
};

template <int SF, int HV = 0> struct W3Lds {
    v2f      *data;  // [NG][AR][SA]
    v2f      *tw;    // [NTW]  W_N^t
    W3Shared *ws;
};

template <int SF, int HV = 0>
__device__ __forceinline__ W3Lds<SF, HV> w3_carve(unsigned char *smem)
{
    using G = W3Geom<SF, HV>;
    W3Lds<SF, HV> L;
    L.data = reinterpret_cast<v2f *>(smem);
    L.tw = L.data + (size_t)G::NG * G::data_entries;
    L.ws = reinterpret_cast<W3Shared *>(L.tw + G::NTW);
    return L;
}
template <int SF, int HV = 0> constexpr uint32_t w3_lds_bytes()
{
    using G = W3Geom<SF, HV>;
    return (uint32_t)(((size_t)G::NG * G::data_entries + G::NTW) * sizeof(v2f) + ((sizeof(W3Shared) + 15) & ~(size_t)15));
}

template <int SF, int HV = 0>
__device__ __forceinline__ v2f w3_tw(const W3Lds<SF, HV> &L, uint32_t idx)
{
    using G = W3Geom<SF, HV>;
    if constexpr (G::NTW == G::N) return L.tw[idx];
    else {
        const v2f w = L.tw[idx & (uint32_t)(G::NTW - 1)];
        return (idx & (uint32_t)G::NTW) ? -w : w;
    }
}

// K complex multiplies a[i] *= w[i], the multiplies first and the multiply-adds after them (a v_pk_fma_f32 directly behind
// the v_pk_mul_f32 it depends on costs a wait state)
template <int K>
__device__ __forceinline__ void cmul_batch(v2f *a, const v2f *w)
{
    v2f t[K];
#pragma unroll
    for (int i = 0; i < K; i++) asm("v_pk_mul_f32 %0, %1, %2 op_sel:[1,1] op_sel_hi:[1,0] neg_lo:[0,1]" : "=v"(t[i]) : "v"(a[i]), "v"(w[i]));
#pragma unroll
    for (int i = 0; i < K; i++) asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel_hi:[0,1,1]" : "=v"(a[i]) : "v"(a[i]), "v"(w[i]), "v"(t[i]));
}
// Sums each of the 32 floats v[0..31] over the 8 lanes that share lane bits 3-5 (r = lane bits 0-2): quad xor 1, quad xor 2,
// then the mirror inside each half row (which pairs the two quads) - one v_add_f32 with a DPP operand per step, written out
// level by level: left to the compiler the adds are re-packed into v_pk_add_f32 (no DPP operand: a v_mov_b32_dpp per
// component in front) and every dependent step then waits out the VALU -> DPP hazard
__device__ __forceinline__ void w3_sum_r32(float (&v)[32])
{
#pragma unroll
    for (int i = 0; i < 32; i++) asm volatile("v_add_f32_dpp %0, %1, %1 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf bound_ctrl:1" : "=v"(v[i]) : "v"(v[i]));
#pragma unroll
    for (int i = 0; i < 32; i++) asm volatile("v_add_f32_dpp %0, %1, %1 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf bound_ctrl:1" : "=v"(v[i]) : "v"(v[i]));
#pragma unroll
    for (int i = 0; i < 32; i++) asm volatile("v_add_f32_dpp %0, %1, %1 row_half_mirror row_mask:0xf bank_mask:0xf bound_ctrl:1" : "=v"(v[i]) : "v"(v[i]));
}
// Raw buffer loads: address = descriptor base (scalar) + lane byte offset (one VGPR shared by all the loads of a thread)
// + a scalar or immediate offset per load - no per-load address arithmetic on the VALU (global_load with 64-bit lane
// addresses costs two VALU adds per load once the chunk offsets exceed the 12-bit immediate).
typedef __amdgpu_buffer_rsrc_t w3_buf_t;
__device__ __forceinline__ w3_buf_t w3_buf(const void *uniform_base)
{
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void *>(uniform_base), 0, 0x7fffffff, 0x00020000); // raw, 32-bit elements (gfx9 word 3)
}
__device__ __forceinline__ v2f w3_ld2(w3_buf_t b, uint32_t voff, uint32_t soff)
{
    return __builtin_bit_cast(v2f, __builtin_amdgcn_raw_buffer_load_b64(b, (int)voff, (int)soff, 0));
}
__device__ __forceinline__ float w3_ld1(w3_buf_t b, uint32_t voff, uint32_t soff)
{
    return __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(b, (int)voff, (int)soff, 0));
}
template <typename T> __device__ __forceinline__ T *w3_uniform_ptr(T *p)
{
    const uint64_t b = (uint64_t)p;
    const uint32_t lo = __builtin_amdgcn_readfirstlane((uint32_t)b), hi = __builtin_amdgcn_readfirstlane((uint32_t)(b >> 32));
    return (T *)(((uint64_t)hi << 32) | lo);
}
__device__ __forceinline__ float w3_uni(float v) { return __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, v))); }
// a decision derived from floats (VALU results: there is no scalar float ALU) as a scalar condition, so that what depends
// on it - the whole replicated state machine - stays in scalar registers
__device__ __forceinline__ bool w3_ub(bool b) { return __builtin_amdgcn_readfirstlane(b ? 1 : 0) != 0; }

// ---- group reductions: one barrier; slots alternate.  Every thread gets the totals of its OWN group (uniform); with `all`
// (wave-uniform) the totals of EVERY group - what thread 0's replay consumes, so only its wavefront pays for them --------
template <int SF, int K, int HV = 0>
__device__ __forceinline__ void w3_group_sums(float (&v)[K], W3Shared &ws, int &slot, int grp, int gwave, float (&out)[W3Geom<SF, HV>::NG][K], bool all = true)
{
    using G = W3Geom<SF, HV>;
    const int lane = threadIdx.x & 63;
    float (*red)[72] = ws.red[slot];
    slot ^= 1;
#pragma unroll
    for (int k = 0; k < K; k++) v[k] = wave_sum_rows(v[k]);
    if (lane == 0) {
#pragma unroll
        for (int k = 0; k < K; k++) red[grp][gwave * K + k] = v[k];
    }
    __syncthreads();
#pragma unroll
    for (int g = 0; g < G::NG; g++) {
        if (!all && g != grp) {
#pragma unroll
            for (int k = 0; k < K; k++) out[g][k] = 0.0f;
            continue;
        }
#pragma unroll
        for (int k = 0; k < K; k++) {
            float s = 0.0f;
#pragma unroll
            for (int w = 0; w < G::GW; w++) s += red[g][w * K + k];
            out[g][k] = w3_uni(s);
        }
    }
}

template <int SF, int HV = 0>
__device__ __forceinline__ void w3_group_argmax_first(float v, int idx, W3Shared &ws, int &slot, int grp, int gwave, float (&bv_out)[W3Geom<SF, HV>::NG],
                                                      int (&bi_out)[W3Geom<SF, HV>::NG], bool all = true)
{
    using G = W3Geom<SF, HV>;
    const int lane = threadIdx.x & 63;
    float (*red)[72] = ws.red[slot];
    slot ^= 1;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        const float ov = __shfl_xor(v, o, 64);
        const int oi = __shfl_xor(idx, o, 64);
        if (ov > v || (ov == v && oi < idx)) { v = ov; idx = oi; }
    }
    if (lane == 0) { red[grp][gwave] = v; ((int *)red[grp])[32 + gwave] = idx; }
    __syncthreads();
#pragma unroll
    for (int g = 0; g < G::NG; g++) {
        if (!all && g != grp) { bv_out[g] = 0.0f; bi_out[g] = 0; continue; }
        float bv = red[g][0];
        int bi = ((int *)red[g])[32];
#pragma unroll
        for (int w = 1; w < G::GW; w++) {
            const float ov = red[g][w];
            const int oi = ((int *)red[g])[32 + w];
            if (ov > bv || (ov == bv && oi < bi)) { bv = ov; bi = oi; }
        }
        bv_out[g] = w3_uni(bv);
        bi_out[g] = __builtin_amdgcn_readfirstlane(bi);
    }
}

__device__ __forceinline__ double w3_wave_sum_d(double v)
{
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}

// atan2 of four points, as two packed evaluations of lean_atan2_pk's polynomial with the Horner steps interleaved (a
// v_pk_fma_f32 directly behind the one it depends on costs a wait state)
__device__ __forceinline__ void w3_atan2_x4(v2f y0, v2f x0, v2f y1, v2f x1, v2f &r0_out, v2f &r1_out)
{
    const float ax00 = fabsf(x0.x), ay00 = fabsf(y0.x), ax01 = fabsf(x0.y), ay01 = fabsf(y0.y);
    const float ax10 = fabsf(x1.x), ay10 = fabsf(y1.x), ax11 = fabsf(x1.y), ay11 = fabsf(y1.y);
    const float m00 = fmaxf(ax00, ay00), m01 = fmaxf(ax01, ay01); // atan2(0, 0) = NaN: a zero product means a SAMPLE of exactly zero - the NaN poisons the sums it feeds and
    const float m10 = fmaxf(ax10, ay10), m11 = fmaxf(ax11, ay11); // the window is evaluated again by the ZM instantiations (lean_atan2_pk, lora_wave_demod.inc.hip)
    v2f a0, a1;
    a0.x = fminf(ax00, ay00) * __builtin_amdgcn_rcpf(m00);
    a1.x = fminf(ax10, ay10) * __builtin_amdgcn_rcpf(m10);
    a0.y = fminf(ax01, ay01) * __builtin_amdgcn_rcpf(m01);
    a1.y = fminf(ax11, ay11) * __builtin_amdgcn_rcpf(m11);
    const v2f s0 = a0 * a0, s1 = a1 * a1;
    v2f p0 = (v2f){-0.0040545230731368065f, -0.0040545230731368065f}, p1 = p0;
#define LORA_W3_HORNER(C) p0 = __builtin_elementwise_fma(p0, s0, (v2f){C, C}); p1 = __builtin_elementwise_fma(p1, s1, (v2f){C, C})
    LORA_W3_HORNER(0.02186279185116291f);
    LORA_W3_HORNER(-0.0559120774269104f);
    LORA_W3_HORNER(0.09642177820205688f);
    LORA_W3_HORNER(-0.13908621668815613f);
    LORA_W3_HORNER(0.19946564733982086f);
    LORA_W3_HORNER(-0.33329859375953674f);
    LORA_W3_HORNER(0.9999993443489075f);
#undef LORA_W3_HORNER
    const v2f q0 = a0 * p0, q1 = a1 * p1;
    float r00 = q0.x, r01 = q0.y, r10 = q1.x, r11 = q1.y;
    r00 = (ay00 > ax00) ? 1.57079632679489662f - r00 : r00;
    r10 = (ay10 > ax10) ? 1.57079632679489662f - r10 : r10;
    r01 = (ay01 > ax01) ? 1.57079632679489662f - r01 : r01;
    r11 = (ay11 > ax11) ? 1.57079632679489662f - r11 : r11;
    r00 = (x0.x < 0.0f) ? 3.14159265358979324f - r00 : r00;
    r10 = (x1.x < 0.0f) ? 3.14159265358979324f - r10 : r10;
    r01 = (x0.y < 0.0f) ? 3.14159265358979324f - r01 : r01;
    r11 = (x1.y < 0.0f) ? 3.14159265358979324f - r11 : r11;
    r0_out = (v2f){copysignf(r00, y0.x), copysignf(r01, y0.y)};
    r1_out = (v2f){copysignf(r10, y1.x), copysignf(r11, y1.y)};
}

// the ifreq of a thread's 16 chunk samples a[c] (n = c CH + base): f[c] = ifreq[n - 1] = arg(x[n] conj(x[n-1])), with the
// predecessors x[n-1] loaded by the caller (a second, cache-hot round of coalesced loads: cheaper than moving the neighbour
// lane's sample over with DPP and patching lane 0 from a boundary load - 8 VALU instructions per sample pair)
template <bool FIRST, bool ZM = false>
__device__ __forceinline__ void w3_ifreq16(const v2f (&a)[16], const v2f (&ap)[16], bool n0_thread, float (&f)[16])
{
#pragma unroll
    for (int c = 0; c < 16; c += 4) {
        if (c == 8) __builtin_amdgcn_sched_barrier(0);
        v2f im0, re0, im1, re1, o0, o1;
        im0 = (v2f){a[c].y * ap[c].x - a[c].x * ap[c].y, a[c + 1].y * ap[c + 1].x - a[c + 1].x * ap[c + 1].y};
        re0 = (v2f){a[c].x * ap[c].x + a[c].y * ap[c].y, a[c + 1].x * ap[c + 1].x + a[c + 1].y * ap[c + 1].y};
        im1 = (v2f){a[c + 2].y * ap[c + 2].x - a[c + 2].x * ap[c + 2].y, a[c + 3].y * ap[c + 3].x - a[c + 3].x * ap[c + 3].y};
        re1 = (v2f){a[c + 2].x * ap[c + 2].x + a[c + 2].y * ap[c + 2].y, a[c + 3].x * ap[c + 3].x + a[c + 3].y * ap[c + 3].y};
        w3_atan2_x4(im0, re0, im1, re1, o0, o1);
        f[c] = (FIRST && c == 0 && n0_thread) ? 0.0f : o0.x; // n = 0 has no predecessor in the window
        f[c + 1] = o0.y; f[c + 2] = o1.x; f[c + 3] = o1.y;
    }
    if constexpr (ZM) { // the values next to a sample of exactly zero came out NaN: those as the reference forms them (ifreq_prod_z, out of line; rare)
#pragma unroll
        for (int c = 0; c < 16; c++)
            if (__builtin_amdgcn_ballot_w64(poisoned(f[c])) != 0ull) // (wave-uniform)
                if (poisoned(f[c])) f[c] = ifreq_prod_z(make_float2(ap[c].x, ap[c].y), make_float2(a[c].x, a[c].y));
    }
}

// ---- the cooperative symbol demodulator, one window per group -------------------------------------------------
// Group g evaluates the window at x_g (when valid_g): s[g] = get_shift_fft's value (:430-464); fine[g] = d_fine_sync after
// fine_sync(bin_idx, 2) (:300-338, :501-502; 0 when drift correction is off); en[g] = determine_energy (:368-375) when
// want_energy.  Called by all threads of the workgroup (barriers inside); the results of every group come back uniform.
struct W3DemodOut { uint32_t s[4]; int32_t fine[4]; float en[4]; int slot; };
struct W3DemodArgs { const float2 *down, *ctab; const float *up_ifreq_v; uint32_t enable_fine_sync, demod_mode; uint32_t ffs_on; float ffs_alpha, ffs_jump, ffs_tol; /* DevParams::ffs_* */ };
// fine[g] of a POISONED window (a sample of exactly zero: the NaN of its products has reached the window's fine_sync sums): the caller has the round evaluated
// again by the ZM = true instantiation, which forms every ifreq value as the reference does (std::arg(0) = 0: lora_kernels.hip, ifreq_prod_z)
constexpr int32_t kFinePoison = 0x7ffffff0;
template <int SF, int HV = 0, bool ZM = false>
__device__ __forceinline__ void w3_demod_round(const W3DemodArgs &P, const W3Lds<SF, HV> &L, const float2 *const (&xa)[W3Geom<SF, HV>::NG] /* every group's window (uniform) */,
                                               uint32_t vmask /* bit g: group g has a window */, bool want_energy, int &slot,
                                               uint32_t (&s_out)[W3Geom<SF, HV>::NG], int32_t (&fine_out)[W3Geom<SF, HV>::NG], float (&en_out)[W3Geom<SF, HV>::NG],
                                               long long *stamps = nullptr /* tools/demod_bench.py --stamps: clock per phase */)
{
#define LORA_W3STAMP(i) do { if (stamps) stamps[i] = clock64(); } while (0)
    LORA_W3STAMP(0);
    using G = W3Geom<SF, HV>;
    constexpr int N = G::N, SPS = G::SPS, TG = G::TG, CH = G::CH, M2 = G::M2, AR = G::AR, SA = G::SA, PAIRS = G::PAIRS, ROUNDS = G::ROUNDS, NG = G::NG, VT = G::VT, U = G::U, GW = G::GW;
    // fine_sync's ifreq: never kept from pass 1 (round 6).  The common decision - lag 0 - is taken in closed form from sign tests made while the samples are in
    // registers for the dechirp (FFS: wave_demod_symbol FMODE 2, lora_wave_demod.inc.hip, explains the rule; ffs_row is the per-sample part); a window the closed
    // form cannot vouch for has its three sums formed from a second read of the window (the path SF12 always took, and every ZM evaluation takes)
    constexpr bool FFS = !ZM;
    int tt = threadIdx.x;
    asm volatile("" : "+v"(tt)); // keeps per-thread table addresses out of the caller's loop-invariant set
    const int grp = __builtin_amdgcn_readfirstlane(tt / TG), t = tt % TG;
    const int gwave = __builtin_amdgcn_readfirstlane(t >> 6), r = t & 7;
    const bool want_fine = P.enable_fine_sync != 0u;
    const float2 *__restrict__ x = xa[0];
#pragma unroll
    for (int g = 1; g < NG; g++) x = grp == g ? xa[g] : x;
    const bool valid = ((vmask >> grp) & 1u) != 0u;
    const w3_buf_t xb = w3_buf(w3_uniform_ptr(x)), db = w3_buf(w3_uniform_ptr(P.down)), cb = w3_buf(w3_uniform_ptr(P.ctab));
    W3Shared &ws = *L.ws;
    v2f *data = L.data + (size_t)grp * G::data_entries;
    const uint32_t tu = (uint32_t)t;
    const int lane = tt & 63;

    // closed-form fine_sync: this thread's rows are (p, c) <-> n = c CH + p TG + t, row index 16 p + c; the lanes of a wavefront hold 64 consecutive samples of a row
    constexpr int NROW = 16 * PAIRS, NMW = (NROW + 31) / 32, CLS = kFfsClass<SF>;
    const bool ffs = FFS && want_fine && P.ffs_on != 0u; // (uniform)
    uint32_t mA[NMW], mC[NMW];
#pragma unroll
    for (int g = 0; g < NMW; g++) { mA[g] = 0u; mC[g] = 0u; }
    float zmin = 3.0e38f;
    v2f raw0 = (v2f){0.0f, 0.0f}, rawE = (v2f){0.0f, 0.0f}; // x[0] (thread 0) / x[sps-2], x[sps-1] (the group's last two threads)
    float ffs_W = 0.0f, ffs_th = 0.0f;                      // this wavefront's winding count; arg of raw0 / rawE
    int ffs_zb = 0;                                          // this wavefront's min of the class / non-zero test, as integer bits (<= 0: the closed form may not vouch)

    v2f hold[ROUNDS > 1 ? PAIRS : 1][8]; // SF12: rows 8..15 of pass 1 wait here for round 1
    float en = 0.0f;

    // ---- pass 1 ----
    if (valid) {
        // Pass 1 is a chain of memory round trips if written pair by pair (16 samples, then their 16 dechirp entries, pair after pair: 19 k of a round's 38 k
        // clocks at SF10 with under 4 k of arithmetic in them, LORA_HIP_W3_STAMPS).  LORA_W3_P1_PREFETCH requests pair p + 1's samples and pair p's dechirp
        // entries before pair p is worked on: with fine_sync's 32 ifreq registers gone it fits without a spill and takes block 0's pass 1 from 19 k to 12 k
        // clocks - and the device as a whole 4-7 % DOWN (standalone 0.283 / 0.273 / 0.253 -> 0.264 / 0.256 / 0.237 at SF10 / SF11 / SF12, walkers -1 ... -4 %):
        // every workgroup of the launch then asks for twice as much at once.  Off.
#if LORA_W3_P1_PREFETCH
        v2f nx[16];
#pragma unroll
        for (int c = 0; c < 16; c++) nx[c] = w3_ld2(xb, 8u * tu, (uint32_t)(c * CH * 8));
#endif
#pragma unroll
        for (int p = 0; p < PAIRS; p++) {
            const int base = p * TG + t;
            const uint32_t ob = 8u * ((uint32_t)(p * TG) + tu); // byte offset of this thread's sample inside a chunk
            v2f a[16];
#if LORA_W3_P1_PREFETCH
            v2f dd[16];
#pragma unroll
            for (int c = 0; c < 16; c++) dd[c] = w3_ld2(db, ob, (uint32_t)(c * CH * 8));
#pragma unroll
            for (int c = 0; c < 16; c++) a[c] = nx[c];
            if (p + 1 < PAIRS) {
#pragma unroll
                for (int c = 0; c < 16; c++) nx[c] = w3_ld2(xb, ob + 8u * (uint32_t)TG, (uint32_t)(c * CH * 8));
            }
            __builtin_amdgcn_sched_barrier(0); // (the requests stay in front of the work)
#else
#pragma unroll
            for (int c = 0; c < 16; c++) a[c] = w3_ld2(xb, ob, (uint32_t)(c * CH * 8));
#endif
            if (want_energy) {
#pragma unroll
                for (int c = 0; c < 16; c++) en += a[c].x * a[c].x + a[c].y * a[c].y;
            }
            if (ffs) { // (uniform) sign tests of this thread's 16 samples (ffs_row)
                if (p == 0) raw0 = a[0];
                if (p == PAIRS - 1) rawE = a[15];
#pragma unroll
                for (int c = 0; c < 16; c++) {
                    constexpr int dummy = 0; (void)dummy;
                    const int row = p * 16 + c;
                    float tq, re;
                    // the first and the last four products of the window are not held to the class bound (wave_demod_symbol), only to being non-zero
                    const bool ends = CLS != 0 && ((p == 0 && c == 0) || (p == PAIRS - 1 && c == 15));
                    if (ends) {
                        ffs_row<CLS, false>(a[c].x, a[c].y, mA[row >> 5], mC[row >> 5], zmin, tq, re);
                        float u = CLS == 1 ? re : __builtin_fmaf(-2.0f, fabsf(tq), re);
                        u = ((p == 0 && c == 0) ? (gwave == 0 && lane < 4) : (gwave == GW - 1 && lane >= 60)) ? 1.0f : u;
                        asm("v_min3_f32 %0, %1, |%2|, %3" : "=v"(zmin) : "v"(u), "v"(tq), "v"(zmin));
                    } else {
                        ffs_row<CLS, true>(a[c].x, a[c].y, mA[row >> 5], mC[row >> 5], zmin, tq, re);
                    }
                }
            }
#pragma unroll
            for (int h = 0; h < 2; h++) { // dechirp (:437)
#if LORA_W3_P1_PREFETCH
                cmul_batch<8>(a + 8 * h, dd + 8 * h);
#else
                v2f d[8];
#pragma unroll
                for (int c = 0; c < 8; c++) d[c] = w3_ld2(db, ob, (uint32_t)((8 * h + c) * CH * 8));
                cmul_batch<8>(a + 8 * h, d);
#endif
            }
            fft_inlane_dif_pk<16>(a);
            const int q0 = base >> 3;
            {
                v2f w[8];
#pragma unroll
                for (int m = 1; m < 8; m++) w[m] = w3_tw<SF, HV>(L, (uint32_t)(q0 * brev_bits(m, 4)));
                cmul_batch<7>(a + 1, w + 1);
#pragma unroll
                for (int m = 8; m < 16; m++) w[m - 8] = w3_tw<SF, HV>(L, (uint32_t)(q0 * brev_bits(m, 4)));
                cmul_batch<8>(a + 8, w);
            }
#pragma unroll
            for (int m = 0; m < AR; m++) data[m * SA + q0 * 8 + r] = a[m];
            if constexpr (ROUNDS > 1) {
#pragma unroll
                for (int i = 0; i < 8; i++) hold[p][i] = a[8 + i];
            }
        }
    }

    if (ffs && valid) { // this wavefront's share of the window's winding number, and its class / non-zero test
        zmin = lane == 0 ? 3.0e38f : zmin; // (lane 0's products were taken with lane 63's sample of its own row)
        int cnt = 0;
#pragma unroll
        for (int g = 0; g < NMW; g++) {
            const uint32_t A = mA[g], Cm = mC[g];
            const uint32_t B = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)A, kDppWaveRor1, 0xf, 0xf, true); // Im x[n-1] < 0: the neighbour's bit of the same row
            uint32_t q = ((Cm & B) | (~Cm & A)) & (A ^ B);
            q = lane == 0 ? 0u : q;
            cnt += __builtin_popcount(q) - 2 * __builtin_popcount(q & Cm);
        }
        { // the rows' first samples: lane l < NROW holds (x[n - 1], x[n]) of row l = 16 p + c, n = c CH + p TG + 64 gwave (the window's n = 0 has no predecessor)
            const int pl = lane >> 4, cl = lane & 15;
            const int nl = cl * CH + pl * TG + 64 * gwave;
            const bool mine = lane < NROW && nl > 0;
            typedef float f4u __attribute__((ext_vector_type(4), aligned(8)));
            const f4u pb = *reinterpret_cast<const __attribute__((address_space(1))) f4u *>((const __attribute__((address_space(1))) float *)x + (mine ? 2 * nl - 2 : 0));
            const float tq = pb.w * pb.x - pb.z * pb.y, re = pb.z * pb.x + pb.w * pb.y; // x[n] conj x[n-1]
            const uint32_t A = __builtin_bit_cast(uint32_t, pb.w), B = __builtin_bit_cast(uint32_t, pb.y), Cm = __builtin_bit_cast(uint32_t, tq);
            const uint32_t q = mine ? (((Cm & B) | (~Cm & A)) & (A ^ B)) : 0u;
            cnt += (int)(q >> 31) - 2 * (int)((q & Cm) >> 31);
            float u = fabsf(tq);
            if constexpr (CLS != 0) u = fminf(u, CLS == 1 ? re : __builtin_fmaf(-2.0f, u, re));
            zmin = mine ? fminf(zmin, u) : zmin;
        }
        ffs_W = wave_sum_u((float)cnt);
        ffs_zb = wave_min_u(__builtin_bit_cast(int, zmin));
        const v2f sel = (gwave == 0 && lane == 0) ? raw0 : rawE;
        ffs_th = lean_atan2_pk((v2f){sel.y, sel.y}, (v2f){sel.x, sel.x}).x; // arg x[0] (lane 0 of the group's first wavefront), arg x[sps-2], arg x[sps-1] (lanes 62, 63 of its last)
    }

    LORA_W3STAMP(1);
    float bv = -1.0f;
    int bi = 0x7fffffff;
#pragma unroll
    for (int g = 0; g < ROUNDS; g++) {
        if (g > 0) {
            if constexpr (ROUNDS > 1) {
                __syncthreads(); // round 0's pass 3 has read the rows
                if (valid) {
#pragma unroll
                    for (int p = 0; p < PAIRS; p++) {
                        const int q0 = (p * TG + t) >> 3;
#pragma unroll
                        for (int i = 0; i < 8; i++) data[i * SA + q0 * 8 + r] = hold[p][i];
                    }
                }
            }
        }
        __syncthreads();
        if (g == 0) LORA_W3STAMP(2);
        // ---- pass 2 (in place) ----
        if (valid) {
#pragma unroll
            for (int u = 0; u < U; u++) { // (unit vt of the group: what thread vt does at one unit per thread)
            const int vt = t + u * TG;
            const int row = (vt >> 3) % AR, q1 = __builtin_amdgcn_readfirstlane(vt / (8 * AR)); // (8 AR >= 64: the same in all lanes)
            v2f *pe = data + row * SA + q1 * 8 + r;
            v2f a[16];
#pragma unroll
            for (int c2 = 0; c2 < 16; c2++) a[c2] = pe[M2 * 8 * c2];
            fft_inlane_dif_pk<16>(a);
            if (q1 != 0) { // W_{N/16}^{q1 a2} = W_N^{16 q1 a2}; q1 is the same in all lanes of a wavefront
                v2f w[8];
#pragma unroll
                for (int m = 1; m < 8; m++) w[m] = w3_tw<SF, HV>(L, (uint32_t)(16 * q1 * brev_bits(m, 4)));
                cmul_batch<7>(a + 1, w + 1);
#pragma unroll
                for (int m = 8; m < 16; m++) w[m - 8] = w3_tw<SF, HV>(L, (uint32_t)(16 * q1 * brev_bits(m, 4)));
                cmul_batch<8>(a + 8, w);
            }
#pragma unroll
            for (int m = 0; m < 16; m++) pe[M2 * 8 * m] = a[m];
            }
        }
        if (g == 0) LORA_W3STAMP(3);
        __syncthreads();
        if (g == 0) LORA_W3STAMP(4);
        // ---- pass 3 + combine ----
        if (valid) {
#pragma unroll
            for (int u = 0; u < U; u++) {
            const int vt = t + u * TG;
            const int w = vt >> 3, row = w % AR, wl = w / AR;
            v2f out[16]; // the combine coefficients first (pass-3 unit order; requesting them before pass 2 costs more in registers than the latency it hides)
#pragma unroll
            for (int i = 0; i < 16; i++) out[i] = w3_ld2(cb, 8u * (uint32_t)vt, (uint32_t)((g * 16 + i) * VT * 8));
#pragma unroll
            for (int j = 0; j < G::NB; j++) {
                const int m2 = wl * G::NB + j;
                v2f b[M2];
#pragma unroll
                for (int q1 = 0; q1 < M2; q1++) b[q1] = data[row * SA + (q1 + M2 * m2) * 8 + r];
                fft_inlane_dif_pk<M2>(b);
                cmul_batch<M2>(out + j * M2, b);
            }
            float mag[16];
            {
                float o32[32]; // sum over r = lane bits 0, 1, 2
#pragma unroll
                for (int i = 0; i < 16; i++) { const float ox = out[i].x, oy = out[i].y; o32[2 * i] = ox; o32[2 * i + 1] = oy; }
                w3_sum_r32(o32);
#pragma unroll
                for (int i = 0; i < 16; i++) mag[i] = o32[2 * i] * o32[2 * i] + o32[2 * i + 1] * o32[2 * i + 1]; // |X|^2: monotone in std::abs (:454)
            }
            // first maximum in bin order among this thread's 16 bins: k1 = a + 16 a2 + 256 b2 with a = bitrev4(row + AR g),
            // a2 = bitrev4(wl NB + j) = bitrev4(wl NB) + bitrev4(j): a per-thread base plus a compile-time constant per value
            float mx = mag[0];
#pragma unroll
            for (int i = 1; i < 16; i++) mx = fmaxf(mx, mag[i]);
            int kc = 0x7fffffff;
#pragma unroll
            for (int j = 0; j < G::NB; j++)
#pragma unroll
                for (int m = 0; m < M2; m++) {
                    const int kconst = 16 * brev_bits(j, 4) + 256 * brev_bits(m, G::LOGM2);
                    kc = min(kc, mag[j * M2 + m] == mx ? kconst : 0x7fffffff);
                }
            const int k1 = kc + (int)(__brev((uint32_t)(row + AR * g)) >> 28) + 16 * (int)(__brev((uint32_t)(wl * G::NB)) >> 28);
            const bool better = mx > bv || (mx == bv && k1 < bi);
            bv = better ? mx : bv;
            bi = better ? k1 : bi;
            }
        }
    }
    LORA_W3STAMP(5);
    const bool all = grp == 0 && gwave == 0; // thread 0's wavefront: the replay needs every group's results
    float bvs[NG];
    int bis[NG];
    float ffs_F[NG];  // (closed form) F = sum_k ifreq[k] of every group's window
    bool ffs_ok[NG];  // (closed form) the group's window passed the class / non-zero test
    { // the group arg-max (first maximum in bin order) and the closed form's sums, one barrier; every thread gets every group's results
        float (*red)[72] = ws.red[slot];
        slot ^= 1;
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
            const float ov = __shfl_xor(bv, o, 64);
            const int oi = __shfl_xor(bi, o, 64);
            if (ov > bv || (ov == bv && oi < bi)) { bv = ov; bi = oi; }
        }
        if (lane == 0) {
            red[grp][gwave] = bv; ((int *)red[grp])[32 + gwave] = bi;
            if (FFS) { red[grp][8 + gwave] = ffs_W; ((int *)red[grp])[16 + gwave] = (ffs && valid) ? ffs_zb : 0; }
        }
        if (FFS && ffs && valid) {
            if (gwave == 0 && lane == 0) red[grp][24] = ffs_th;
            if (gwave == GW - 1 && lane >= 62) red[grp][25 + (lane - 62)] = ffs_th;
        }
        __syncthreads();
#pragma unroll
        for (int g = 0; g < NG; g++) {
            float gv = red[g][0];
            int gi = ((int *)red[g])[32];
#pragma unroll
            for (int w = 1; w < GW; w++) {
                const float ov = red[g][w];
                const int oi = ((int *)red[g])[32 + w];
                if (ov > gv || (ov == gv && oi < gi)) { gv = ov; gi = oi; }
            }
            bvs[g] = w3_uni(gv);
            bis[g] = __builtin_amdgcn_readfirstlane(gi);
            ffs_F[g] = 0.0f; ffs_ok[g] = false;
            if constexpr (FFS) {
                if (ffs) {
                    float Wg = 0.0f;
                    int zb = 0x7fffffff;
#pragma unroll
                    for (int w = 0; w < GW; w++) { Wg += red[g][8 + w]; zb = min(zb, ((int *)red[g])[16 + w]); }
                    const float th0 = red[g][24], th2 = red[g][25], the = red[g][26];
                    float last = the - th2; // ifreq[sps-1] = ifreq[sps-2] (:243)
                    last = last > 3.14159265358979324f ? last - 6.28318530717958648f : (last < -3.14159265358979324f ? last + 6.28318530717958648f : last);
                    ffs_F[g] = w3_uni((the - th0) + 6.28318530717958648f * Wg + last);
                    ffs_ok[g] = w3_ub(zb > 0 && ffs_F[g] == ffs_F[g]);
                }
            }
        }
    }
    LORA_W3STAMP(6);
#pragma unroll
    for (int g = 0; g < NG; g++) { s_out[g] = (uint32_t)bis[g]; fine_out[g] = 0; en_out[g] = 0.0f; }
    if (want_energy) {
        float e1[1] = {en}, eo[NG][1];
        w3_group_sums<SF, 1, HV>(e1, ws, slot, grp, gwave, eo, all);
#pragma unroll
        for (int g = 0; g < NG; g++) en_out[g] = eo[g][0];
    }
    if (!want_fine) return;
    // fine_sync (:300-338) with search = max(D/4, 2) = 2 -> lags -1, 0, +1.  First the closed form, for every group in every thread (the decisions are
    // uniform over the WORKGROUP, so that the sums below - a barrier - are skipped by everybody or by nobody)
    bool need[NG];
    bool any_need = false;
#pragma unroll
    for (int g = 0; g < NG; g++) {
        const bool gval = ((vmask >> g) & 1u) != 0u;
        bool sure = false;
        if constexpr (FFS) {
            const uint32_t sg = s_out[g];
            const uint32_t bin_g = (sg == 0u && P.demod_mode == 2u) ? 0u : (sg + (uint32_t)N - 1u) % (uint32_t)N;
            if (gval && ffs_ok[g] && bin_g != (uint32_t)N - 1u) { // (uniform)
                // ifreq[ka - 1], ifreq[ka] next to the template's step (ka = sps - 8 (bin_idx + 1)) from three samples read again, as the reference forms them
                const int ka = SPS - 8 * ((int)bin_g + 1);
                const auto xv = (const __attribute__((address_space(1))) v2f *)xa[g];
                const v2f xs = xv[ka - 1 + (lane < 2 ? lane : 2)];
                const float th = lean_atan2_pk((v2f){xs.y, xs.y}, (v2f){xs.x, xs.x}).x;
                float d = th - dpp_f<kDppWaveRor1>(th);
                d = d > 3.14159265358979324f ? d - 6.28318530717958648f : (d < -3.14159265358979324f ? d + 6.28318530717958648f : d);
                const float fb = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, d), 1)); // ifreq[ka - 1]
                const float fa = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, d), 2)); // ifreq[ka]
                const float D0 = P.ffs_alpha * ffs_F[g] + P.ffs_jump * fa, D1 = P.ffs_alpha * ffs_F[g] + P.ffs_jump * fb; // c(0) - c(-1), c(1) - c(0)
                sure = w3_ub(D0 > P.ffs_tol && D1 < -P.ffs_tol); // c(0) above c(-1), c(1) below c(0), both beyond the table's noise: lag 0 whatever the signs
            }
        }
        need[g] = gval && !sure;
        any_need = any_need || need[g];
    }
    if (!any_need) { // (uniform over the workgroup)
        LORA_W3STAMP(7);
        LORA_W3STAMP(8);
        return;
    }
    bool need_mine = need[0];
#pragma unroll
    for (int g = 1; g < NG; g++) need_mine = grp == g ? need[g] : need_mine;
    float cs[3] = {0.f, 0.f, 0.f};
    // this thread's taps: the window's ifreq from a second read of the window.  Z: the values next to a sample of exactly zero as the reference forms them
    // (w3_ifreq16 ZM).  (Re-evaluating a poisoned window's taps inside the wavefront, ahead of the group sums, instead of in a ZM round was measured as well:
    // the same to +-1 %, profiles/r05_ab_zero_samples.txt.)
    auto tap_sums = [&](auto z_t) {
        constexpr bool Z = decltype(z_t)::value;
        cs[0] = 0.f; cs[1] = 0.f; cs[2] = 0.f;
        if (!valid || !need_mine) return;
        uint32_t s = s_out[0];
#pragma unroll
        for (int g = 1; g < NG; g++) s = (grp == g) ? s_out[g] : s;
        const uint32_t bin_idx = (s == 0u && P.demod_mode == 2u) ? 0u : (s + (uint32_t)N - 1u) % (uint32_t)N;
        // descriptor base one element before v[shift_ref + sps]: lag -1 at k = 0 reads v[-1] of that origin, and buffer offsets are unsigned
        const w3_buf_t vb = w3_buf(w3_uniform_ptr(P.up_ifreq_v + ((int)(bin_idx + 1u) * 8 + SPS - 1)));
#pragma unroll
        for (int p = 0; p < PAIRS; p++) {
            const uint32_t nb = (uint32_t)(p * TG) + tu; // n inside chunk 0
            const uint32_t k0 = 4u * (nb >= 1u ? nb - 1u : 1u); // byte offset of k = n - 1 in chunk 0 (f is 0 for the non-existent k = -1)
            float v0[16], v1[16], v2[16];
            v0[0] = w3_ld1(vb, k0, 0u); v1[0] = w3_ld1(vb, k0, 4u); v2[0] = w3_ld1(vb, k0, 8u); // v[k-1], v[k], v[k+1]
#pragma unroll
            for (int c = 1; c < 16; c++) {
                v0[c] = w3_ld1(vb, 4u * nb, (uint32_t)(c * CH * 4 - 4)); v1[c] = w3_ld1(vb, 4u * nb, (uint32_t)(c * CH * 4)); v2[c] = w3_ld1(vb, 4u * nb, (uint32_t)(c * CH * 4 + 4));
            }
            float fl[16];
            { // second read of the window
                const uint32_t ob = 8u * nb;
                v2f a[16], ap[16];
#pragma unroll
                for (int c = 0; c < 16; c++) a[c] = w3_ld2(xb, ob, (uint32_t)(c * CH * 8));
                ap[0] = w3_ld2(xb, (p == 0) ? (ob >= 8u ? ob - 8u : 0u) : ob - 8u, 0u);
#pragma unroll
                for (int c = 1; c < 16; c++) ap[c] = w3_ld2(xb, ob, (uint32_t)(c * CH * 8 - 8));
                if (p == 0) w3_ifreq16<true, Z>(a, ap, t == 0, fl);
                else w3_ifreq16<false, Z>(a, ap, false, fl);
            }
#pragma unroll
            for (int c = 0; c < 16; c++) {
                const float fk = fl[c];
                cs[0] += fk * v0[c]; cs[1] += fk * v1[c]; cs[2] += fk * v2[c];
            }
            if (p == PAIRS - 1 && t == TG - 1) { // ifreq[sps-1] = ifreq[sps-2] (:243): the duplicated tap at k = sps-1
                const float flast = fl[15];
                const uint32_t ko = 4u * (uint32_t)(SPS - 1);
                cs[0] += flast * w3_ld1(vb, ko, 0u); cs[1] += flast * w3_ld1(vb, ko, 4u); cs[2] += flast * w3_ld1(vb, ko, 8u);
            }
        }
    };
    tap_sums(std::integral_constant<bool, ZM>{});
    LORA_W3STAMP(7);
    float co[NG][3];
    w3_group_sums<SF, 3, HV>(cs, ws, slot, grp, gwave, co, all);
    LORA_W3STAMP(8);
#undef LORA_W3STAMP
#pragma unroll
    for (int g = 0; g < NG; g++) {
        float mx = 0.0f;
        int32_t lag = 0;
        if (co[g][0] > mx) { mx = co[g][0]; lag = -1; }
        if (co[g][1] > mx) { mx = co[g][1]; lag = 0; }
        if (co[g][2] > mx) { mx = co[g][2]; lag = 1; }
        if (!ZM && poisoned(co[g][0] + co[g][1] + co[g][2])) lag = -kFinePoison; // (every group's sums reach thread 0's wavefront, whose replay looks at them)
        fine_out[g] = need[g] ? __builtin_amdgcn_readfirstlane(-lag) : 0;
    }
}

// ---- the reference's SHIPPED demodulator, one window per WAVEFRONT: max_frequency_gradient_idx (:466-491) + fine_sync (:300-338) ----------------------
// No FFT, so nothing has to cross a workgroup: the estimator is a reduction over the window - ifreq, means of D = 8 samples, the largest drop between
// neighbouring means (:474-488) - and the window's three fine_sync correlations another.  A wavefront walks its window in chunks of 1024 samples exactly as
// wave_demod_symbol_grad (lora_wave_demod.inc.hip) handles a whole SF7 symbol: lane l owns n = 1024 q + 64 j + l, every load instruction covers 512
// contiguous bytes, x[n + 1] comes from the neighbouring lane (lane 63: lane 0 of the next register / the next chunk, which is already in flight: the chunks
// are double-buffered in registers), the eight samples of bin i = 128 q + 8 j + (l >> 3) sit in eight adjacent lanes.  The bin is known only behind the
// last chunk, so fine_sync's sums are a SECOND pass over the window (L2-hot) that forms the ifreq values again: two arctangents per sample instead of one,
// and no barrier, no LDS, no idle wavefront - a workgroup's 8 wavefronts demodulate 8 consecutive windows per round at every spreading factor (the
// group-per-window form of rounds 2-4 took 4 / 2 / 1 / 1 windows behind five barriers; same-box A/B: profiles/r05_ab_wave_gradient.txt).
// bin_out is demodulate()'s bin_idx itself; fine_out = kFinePoison for a window with a sample of exactly zero (re-evaluated by ZM = true, which patches
// the poisoned values with ifreq_prod_z).  en_out: determine_energy (:368-375) when want_energy.
// fcache: this wavefront's kW3GradCacheChunks x 1024 floats of LDS - pass A leaves the ifreq of the window's first chunks there and pass B reads them back
// instead of forming them again (the whole window at SF9, half of it at SF10, ...): fewer arctangents and no second read for those chunks.
constexpr int kW3GradCacheChunks = 4;
template <int SF, bool ZM = false>
__device__ __forceinline__ void w3_wave_window_grad(const W3DemodArgs &P, const float2 *__restrict__ x, bool want_energy, uint32_t &bin_out, int32_t &fine_out, float &en_out, float *fcache)
{
    constexpr int N = 1 << SF, SPS = 8 * N, NCH = SPS / 1024, NCC = NCH < kW3GradCacheChunks ? NCH : kW3GradCacheChunks;
    typedef __attribute__((address_space(3))) float lds_f;
    lds_f *fc = (lds_f *)fcache;
    int lane = threadIdx.x & 63;
    asm volatile("" : "+v"(lane)); // (keeps per-lane addresses out of the caller's loop-invariant set)
    const auto xv = (const __attribute__((address_space(1))) v2f *)x;
    // one pass over the window: use(q, f) with f[j] = ifreq[1024 q + 64 j + lane]; ifreq[sps-1] = ifreq[sps-2] (:243)
    auto pass = [&](auto &&use, bool energy, int q_first) {
        v2f nxt[16];
#pragma unroll
        for (int j = 0; j < 16; j++) nxt[j] = xv[q_first * 1024 + j * 64 + lane];
        v2f e2 = (v2f){0.0f, 0.0f};
#pragma unroll 1
        for (int q = q_first; q < NCH; q++) {
            v2f a[16];
#pragma unroll
            for (int j = 0; j < 16; j++) a[j] = nxt[j];
            const bool last = q == NCH - 1;
            if (!last) {
#pragma unroll
                for (int j = 0; j < 16; j++) nxt[j] = xv[(q + 1) * 1024 + j * 64 + lane];
            }
            if (energy) {
#pragma unroll
                for (int j = 0; j < 16; j++) e2 = __builtin_elementwise_fma(a[j], a[j], e2);
            }
            float f[16];
            v2f cn = dpp2<kDppWaveRol1>(a[0]); // a[j] of lane + 1 (lane 63: of lane 0)
#pragma unroll
            for (int j = 0; j < 16; j += 2) {
                const v2f c0 = cn, c1 = dpp2<kDppWaveRol1>(a[j + 1]);
                const v2f c2 = (j + 2 < 16) ? dpp2<kDppWaveRol1>(a[j + 2]) : dpp2<kDppWaveRol1>(nxt[0]); // (behind the last chunk: never used, see below)
                cn = c2;
                const v2f s0 = (lane == 63) ? c1 : c0, s1 = (lane == 63) ? c2 : c1; // x[n + 1]
                const v2f fp = ifreq_prod_pk(a[j], s0, a[j + 1], s1);
                f[j] = fp.x; f[j + 1] = fp.y;
                if constexpr (ZM) { // the values next to a sample of exactly zero came out NaN: those as the reference forms them (rare)
                    if (__builtin_amdgcn_ballot_w64(poisoned(f[j]) || poisoned(f[j + 1])) != 0ull) {
                        if (poisoned(f[j])) f[j] = ifreq_prod_z(make_float2(a[j].x, a[j].y), make_float2(s0.x, s0.y));
                        if (poisoned(f[j + 1])) f[j + 1] = ifreq_prod_z(make_float2(a[j + 1].x, a[j + 1].y), make_float2(s1.x, s1.y));
                    }
                }
            }
            if (last) { // ifreq[sps-1] = ifreq[sps-2] (:243): the sample behind the window's last one is never looked at (it need not exist)
                const float dup = dpp_f<kDppWaveRor1>(f[15]);
                f[15] = (lane == 63) ? dup : f[15];
            }
            use(q, f);
        }
        return e2.x + e2.y;
    };
    // pass A: bin averages (:474-477) and the largest drop (:479-488)
    float bv = 0.1f; // max_gradient = 0.1f
    int bi = 0x7fffffff;
    float gs = 0.0f, prev_perm = 0.0f; // gs carries the poison of a zero sample when there is no fine_sync sum to carry it
    const int m = lane >> 3, perm_addr = ((lane - 8) & 63) << 2;
    // closed-form fine_sync (wave_demod_symbol FMODE 2 explains the rule): here the ifreq values exist anyway, so F = sum_k ifreq[k] is one add per sample and
    // the class bound one max; the two values next to the template's step are formed again behind the bin.  A window it vouches for skips pass B.
    constexpr int CLS = kFfsClass<SF>;
    const bool ffs = !ZM && P.enable_fine_sync != 0u && P.ffs_on != 0u; // (uniform)
    float fsum = 0.0f, amax = 0.0f;
    const float e = pass([&](int q, const float (&f)[16]) {
        if (q < NCC) { // (uniform) kept for pass B
#pragma unroll
            for (int j = 0; j < 16; j++) fc[(q * 16 + j) * 64 + lane] = f[j];
        }
        if (ffs) { // (uniform)
#pragma unroll
            for (int j = 0; j < 16; j++) fsum += f[j];
            if constexpr (CLS != 0) {
#pragma unroll
                for (int j = 0; j < 16; j++) {
                    float am = fabsf(f[j]);
                    // (the first three and the last five ifreq values - the products next to the window's ends - are not held to the class: wave_demod_symbol)
                    if (j == 0) am = (q == 0 && lane < 3) ? 0.0f : am;
                    if (j == 15) am = (q == NCH - 1 && lane >= 59) ? 0.0f : am;
                    amax = fmaxf(amax, am);
                }
            }
        }
#pragma unroll
        for (int j = 0; j < 16; j++) {
            float A = f[j];
            A += dpp_f<kDppQuadXor1>(A); A += dpp_f<kDppQuadXor2>(A); A += dpp_f<kDppRowHalfMirror>(A); // sum over the 8 lanes of the bin
            A *= 0.125f; // / d_decim_factor
            const float perm = __builtin_bit_cast(float, __builtin_amdgcn_ds_bpermute(perm_addr, __builtin_bit_cast(int, A))); // bin (m - 1) mod 8 of this register
            const float left = (m == 0) ? prev_perm : perm; // bin i - 1: for m = 0 bin 7 of the previous register
            prev_perm = perm;
            const float g = left - A; // samples_ifreq_avg[i - 1] - samples_ifreq_avg[i]
            gs += A;
            const int i = 128 * q + 8 * j + m;
            if (i >= 1 && g > bv) { bv = g; bi = i; } // i runs from 1; strict '>' keeps the first maximum
        }
    }, want_energy, 0);
    en_out = want_energy ? wave_sum_u(e) : 0.0f;
    const float best = wave_max_nonneg_u(bv);
    const int first = wave_min_u((bv == best) ? bi : 0x7fffffff);
    const uint32_t max_index = (first == 0x7fffffff) ? 0u : (uint32_t)first + 1u; // :486
    const uint32_t bin_idx = ((uint32_t)N - max_index) % (uint32_t)N;              // :490
    bin_out = bin_idx;
    fine_out = 0;
    if (P.enable_fine_sync == 0u) {
        if (!ZM && poisoned(wave_sum_u(gs))) fine_out = kFinePoison;
        return;
    }
    if (ffs && bin_idx != (uint32_t)N - 1u) { // (uniform)
        const float F = wave_sum_u(fsum); // (NaN: a sample of exactly zero - not vouched for; pass B then reports the poison)
        const float amx = CLS != 0 ? wave_max_nonneg_u(amax) : 0.0f;
        constexpr float kBound = CLS == 1 ? 1.57079632679489662f : 0.46364760900080609f; // pi / 2, atan(1/2): what ffs_tol is computed for (lora_hip_create)
        if (F == F && (CLS == 0 || amx < kBound)) {
            const int ka = SPS - 8 * ((int)bin_idx + 1);
            const v2f xs = xv[ka - 1 + (lane < 2 ? lane : 2)];
            const float th = lean_atan2_pk((v2f){xs.y, xs.y}, (v2f){xs.x, xs.x}).x;
            float d = th - dpp_f<kDppWaveRor1>(th);
            d = d > 3.14159265358979324f ? d - 6.28318530717958648f : (d < -3.14159265358979324f ? d + 6.28318530717958648f : d);
            const float fb = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, d), 1)); // ifreq[ka - 1]
            const float fa = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, d), 2)); // ifreq[ka]
            const float D0 = P.ffs_alpha * F + P.ffs_jump * fa, D1 = P.ffs_alpha * F + P.ffs_jump * fb; // c(0) - c(-1), c(1) - c(0)
            if (D0 > P.ffs_tol && D1 < -P.ffs_tol) return; // (uniform) lag 0 whatever the signs of the sums
        }
    }
    // pass B: fine_sync (:300-338), lags -1, 0, +1: c_lag = sum_k f[k] v[(bin_idx + 1) 8 + sps + lag + k]
    const float *__restrict__ vp = P.up_ifreq_v + ((int)(bin_idx + 1u) * 8 + SPS) + (lane - 1);
    float c0 = 0.f, c1 = 0.f, c2 = 0.f;
    auto taps = [&](int q, const float (&f)[16]) {
        const float *__restrict__ vq = vp + 1024 * q;
#pragma unroll
        for (int j = 0; j < 16; j++) { c0 += f[j] * vq[64 * j]; c1 += f[j] * vq[64 * j + 1]; c2 += f[j] * vq[64 * j + 2]; }
    };
#pragma unroll 1
    for (int q = 0; q < NCC; q++) { // the chunks pass A kept (this lane's own values: no synchronisation needed)
        float f[16];
#pragma unroll
        for (int j = 0; j < 16; j++) f[j] = fc[(q * 16 + j) * 64 + lane];
        taps(q, f);
    }
    if constexpr (NCC < NCH) (void)pass(taps, false, NCC);
    c0 = wave_sum_u(c0); c1 = wave_sum_u(c1); c2 = wave_sum_u(c2);
    if (!ZM && poisoned3(c0, c1, c2)) { fine_out = kFinePoison; return; } // (uniform) a sample of the window is exactly zero: the bin averages next to it are NaN as well
    float mx = 0.0f;
    int32_t lag = 0;
    if (c0 > mx) { mx = c0; lag = -1; }
    if (c1 > mx) { mx = c1; lag = 0; }
    if (c2 > mx) { mx = c2; lag = 1; }
    fine_out = -lag;
}

// copies W_N^t into LDS; all threads; the caller synchronises
template <int SF, int HV = 0>
__device__ __forceinline__ void w3_tables_to_lds(const DevParams &P, const W3Lds<SF, HV> &L)
{
    using G = W3Geom<SF, HV>;
    const v2f *__restrict__ src = reinterpret_cast<const v2f *>(P.w3_tw);
    for (int i = threadIdx.x; i < G::NTW; i += G::T) L.tw[i] = src[i];
}

// ---- acquisition rounds look at several windows per group (round 3).  At one window per group and round an acquisition was a
// chain of 5 DETECT + 1 SYNC + 10 FIND_SFD + 1 PAUSE rounds at SF11 (LORA_HIP_DEBUG), each a first-touch HBM round trip, a
// workgroup-wide reduction and thread 0's replay: a quarter of a job.  Now every group evaluates KD (DETECT) / KS (FIND_SFD)
// windows behind ONE barrier sequence, as walker2's workers do; the partial sums of all windows meet in an LDS scratch area that
// aliases the demodulator's data array (idle in these rounds), and thread 0 replays the NQ = NG K results in order.
#ifndef LORA_W3_SFD_INLINE
#define LORA_W3_SFD_INLINE __attribute__((noinline))
#endif
#ifndef LORA_W3_DET_K
#define LORA_W3_DET_K 0x1111   // DETECT windows per group and round, one hex digit per SF (SF9 lowest)
#endif
#ifndef LORA_W3_SFD_K
#define LORA_W3_SFD_K 0x1111   // FIND_SFD windows per group and round
#endif
// (WIDE: the header-only variants scan two DETECT windows per group and round - a header-only job spends most of its time looking for the next preamble;
// config 4 at 8 s per pass +6 %, at 2 s +1 %, a sparse SF12 pass +3.5 %.  The complete kernels keep one: their passes are payload rounds, and a
// trigger voids the later windows of a round - config 3 at SF9 -2.8 % with two.)
template <int SF, int HV = 0, bool WIDE = false> struct W3Acq {
    static constexpr int KD = WIDE ? 2 : (LORA_W3_DET_K >> (4 * (SF - 9))) & 15, KS = (LORA_W3_SFD_K >> (4 * (SF - 9))) & 15;
    static constexpr int NQD = KD * W3Geom<SF, HV>::NG, NQS = KS * W3Geom<SF, HV>::NG; // windows per round
    static_assert(KD >= 1 && KS >= 1 && NQD <= 8 && NQS <= 8, "results are kept for at most 8 windows per round");
};
struct alignas(16) W3AcqScratch { // in the demodulator's LDS data array
    float   part[8][16][4];    // [window][wavefront of its group][sum]
    double  dpart[8][16][2];   // FIND_SFD: G0, G1 of fine_sync's closed form
    float   edge[8][72];       // FIND_SFD: head / tail samples of the window's ifreq
    int32_t lag[8];
};

// ---- DETECT (:340-366): sums of c1 conj(c2), |c1|^2, |c2|^2 over the symbol pairs of the windows q = grp KD + k at x0 + q sps.
// A group's KD windows are consecutive: its KD + 1 symbols are read once, the energy of a symbol serves the two windows it
// belongs to (summed in the same order either way).  out[q] = {re, im, e1, e2}, uniform.
template <int SF, int HV = 0, bool WIDE = false>
__device__ __forceinline__ void w3_detect_round(const float2 *__restrict__ x0, int n_valid, W3AcqScratch *sc, float (&out)[W3Acq<SF, HV, WIDE>::NQD][4])
{
    using G = W3Geom<SF, HV>;
    constexpr int KD = W3Acq<SF, HV, WIDE>::KD, NQ = W3Acq<SF, HV, WIDE>::NQD, GW = G::GW;
    int tt = threadIdx.x;
    asm volatile("" : "+v"(tt)); // keeps per-thread offsets out of the caller's loop-invariant set (they would be parked in scratch)
    const int grp = __builtin_amdgcn_readfirstlane(tt / G::TG), t = tt % G::TG, gwave = __builtin_amdgcn_readfirstlane(t >> 6), lane = t & 63;
    const int q0 = grp * KD;
    float d0[KD], d1[KD], e[KD + 1];
#pragma unroll
    for (int k = 0; k < KD; k++) { d0[k] = 0.f; d1[k] = 0.f; }
#pragma unroll
    for (int k = 0; k <= KD; k++) e[k] = 0.f;
    if (q0 < n_valid) {
        const w3_buf_t xb = w3_buf(w3_uniform_ptr(x0 + (int64_t)q0 * G::SPS));
#pragma unroll 1
        for (int p = 0; p < G::PAIRS; p++) {
            const uint32_t ob = 8u * (uint32_t)(p * G::TG + t);
            v2f u[16], w[16];
#pragma unroll
            for (int c = 0; c < 16; c++) u[c] = w3_ld2(xb, ob, (uint32_t)(c * G::CH * 8));
#pragma unroll
            for (int c = 0; c < 16; c++) e[0] += u[c].x * u[c].x + u[c].y * u[c].y;
#pragma unroll
            for (int k = 0; k < KD; k++) {
                if (q0 + k < n_valid) { // (uniform per group)
                    const w3_buf_t wb = w3_buf(w3_uniform_ptr(x0 + (int64_t)(q0 + k + 1) * G::SPS));
#pragma unroll
                    for (int c = 0; c < 16; c++) w[c] = w3_ld2(wb, ob, (uint32_t)(c * G::CH * 8));
#pragma unroll
                    for (int c = 0; c < 16; c++) {
                        const v2f c1 = u[c], c2 = w[c];
                        d0[k] += c1.x * c2.x + c1.y * c2.y;
                        d1[k] += c1.y * c2.x - c1.x * c2.y;
                        e[k + 1] += c2.x * c2.x + c2.y * c2.y;
                    }
#pragma unroll
                    for (int c = 0; c < 16; c++) u[c] = w[c];
                }
            }
        }
    }
#pragma unroll
    for (int k = 0; k <= KD; k++) e[k] = wave_sum_rows(e[k]);
#pragma unroll
    for (int k = 0; k < KD; k++) {
        const float s0 = wave_sum_rows(d0[k]), s1 = wave_sum_rows(d1[k]);
        if (lane == 0) { float *o = sc->part[q0 + k][gwave]; o[0] = s0; o[1] = s1; o[2] = e[k]; o[3] = e[k + 1]; }
    }
    __syncthreads();
#pragma unroll
    for (int q = 0; q < NQ; q++)
#pragma unroll
        for (int j = 0; j < 4; j++) {
            float v = 0.0f;
#pragma unroll
            for (int w = 0; w < GW; w++) v += sc->part[q][w][j];
            out[q][j] = w3_uni(v);
        }
}

// ---- SYNC (:770-783, detect_upchirp :392-413), all threads of the workgroup together --------------------------
// C[i] = sum_{k<n} f[i+k] u[k], n = sps-1, i < sps, f = ifreq of x[0 .. 2 sps).  d_upchirp_ifreq is the line a + b k (up
// to float noise ~1e-5 of the peak), so C[i] = a S0[i] + b S1[i] with S0 = sum_k f[i+k], S1 = sum_k k f[i+k], both from
// prefix sums of f and (t - sps) f in double.  Thread tau owns the LEN shifts i0 = LEN tau ..: it computes f on
// A = [i0, i0 + LEN) and on B = [i0 + n, i0 + n + LEN) straight from the samples, the workgroup scans the A and B sums,
// and the thread slides over its shifts.  Returns the best correlation and its (first) shift.
struct W3SyncOut { float bv; int bi; int slot; int pz; }; // pz: the window holds a sample of exactly zero (the sums are NaN): to be evaluated again with ZM = true
// ifreq[base .. base + 16) straight from the samples: f[j] = arg(x[base + j + 1] conj(x[base + j]))
template <bool ZM>
__device__ __forceinline__ void w3_sync_ifreq16(const __attribute__((address_space(1))) v2f *xv, int base, float (&f)[16])
{
    v2f xs[17];
#pragma unroll
    for (int j = 0; j <= 16; j++) xs[j] = xv[base + j];
#pragma unroll
    for (int j = 0; j < 16; j += 2) {
        const v2f fp = ZM ? ifreq_prod_pk_z(xs[j], xs[j + 1], xs[j + 1], xs[j + 2]) : ifreq_prod_pk(xs[j], xs[j + 1], xs[j + 1], xs[j + 2]);
        f[j] = fp.x; f[j + 1] = fp.y;
    }
}
template <int SF, int HV = 0, bool ZM = false>
__device__ __attribute__((noinline)) W3SyncOut w3_sync(double sync_a, double sync_b, const float2 *__restrict__ x, W3Shared *wsp, int slot,
                                                       const float *__restrict__ strict_u /* d_upchirp_ifreq, or nullptr: closed form only */, float *strict_buf)
{
    using G = W3Geom<SF, HV>;
    constexpr int SPS = G::SPS, LEN = G::LEN, WAVES = G::T / 64;
    constexpr int n = SPS - 1;
    W3Shared &ws = *wsp;
    const auto xv = (const __attribute__((address_space(1))) v2f *)x;
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const int i0 = t * LEN;
    if (t == 0) strict::cands_reset(ws.sc); // (ordered before the pushes by the scan's barrier)
    double s0A = 0.0, gA = 0.0, s0B = 0.0, gB = 0.0; // sums of f and of (pos - sps) f over A and B
    // More than 16 shifts per thread (the 512-thread geometry at SF11 / SF12): the two ifreq segments are not kept (2 x 32 / 2 x 64
    // floats beside as many samples: a 0.5-1.1 KB stack frame per lane, 60-180 spilled registers) but computed twice, 16 at a time -
    // once for the sums the scan needs, once for the slide; same values, same order of every sum.
    constexpr bool CHUNKED = LEN > 16;
    float fa[CHUNKED ? 16 : LEN], fb[CHUNKED ? 16 : LEN];
    float f_last_a = 0.0f;
    if constexpr (CHUNKED) {
#pragma unroll 1
        for (int q = 0; q < LEN; q += 16) {
            w3_sync_ifreq16<ZM>(xv, i0 + q, fa);
            w3_sync_ifreq16<ZM>(xv, i0 + n + q, fb);
#pragma unroll
            for (int j = 0; j < 16; j++) {
                s0A += (double)fa[j]; gA += (double)(i0 + q + j - SPS) * (double)fa[j];
                s0B += (double)fb[j]; gB += (double)(i0 + n + q + j - SPS) * (double)fb[j];
            }
            f_last_a = fa[15];
        }
    } else {
    {
        v2f xa[LEN + 1];
#pragma unroll
        for (int j = 0; j <= LEN; j++) xa[j] = xv[i0 + j];
#pragma unroll
        for (int j = 0; j < LEN; j += 2) {
            const v2f fp = ZM ? ifreq_prod_pk_z(xa[j], xa[j + 1], xa[j + 1], xa[j + 2 <= LEN ? j + 2 : LEN]) : ifreq_prod_pk(xa[j], xa[j + 1], xa[j + 1], xa[j + 2 <= LEN ? j + 2 : LEN]);
            fa[j] = fp.x; fa[j + 1] = fp.y;
        }
    }
    {
        v2f xb[LEN + 1];
#pragma unroll
        for (int j = 0; j <= LEN; j++) xb[j] = xv[i0 + n + j];
#pragma unroll
        for (int j = 0; j < LEN; j += 2) {
            const v2f fp = ZM ? ifreq_prod_pk_z(xb[j], xb[j + 1], xb[j + 1], xb[j + 2 <= LEN ? j + 2 : LEN]) : ifreq_prod_pk(xb[j], xb[j + 1], xb[j + 1], xb[j + 2 <= LEN ? j + 2 : LEN]);
            fb[j] = fp.x; fb[j + 1] = fp.y;
        }
    }
#pragma unroll
    for (int j = 0; j < (CHUNKED ? 0 : LEN); j++) {
        s0A += (double)fa[j]; gA += (double)(i0 + j - SPS) * (double)fa[j];
        s0B += (double)fb[j]; gB += (double)(i0 + n + j - SPS) * (double)fb[j];
    }
    f_last_a = fa[LEN - 1];
    }
    // block-wide exclusive scans of the four sums
    double in[4] = {s0A, gA, s0B, gB}, ex[4], tot[4];
    double *dr = ws.dred[slot & 1];
#pragma unroll
    for (int q = 0; q < 4; q++) {
        double v = in[q];
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
            const double u = __shfl_up(v, o, 64);
            if (lane >= o) v += u;
        }
        ex[q] = v - in[q];
        if (lane == 63) dr[wave * 4 + q] = v;
    }
    if (t == G::T - 1) dr[64] = (double)f_last_a; // f[sps-1]: the last element of the last A segment
    __syncthreads();
#pragma unroll
    for (int q = 0; q < 4; q++) {
        double pre = 0.0, all = 0.0;
        for (int w = 0; w < WAVES; w++) { const double v = dr[w * 4 + q]; if (w < wave) pre += v; all += v; }
        ex[q] += pre; tot[q] = all;
    }
    const double flast = dr[64];
    if (!ZM && w3_ub(poisoned((float)(tot[0] + tot[2])))) return W3SyncOut{0.0f, 0x7fffffff, slot, 1}; // (the same totals in every thread)
    // prefix sums over t < i0 and t < i0 + n:  F(i0) = exA,  F(i0 + n) = F(sps - 1) + exB,  F(sps - 1) = totA - f[sps-1]
    const double F0 = ex[0], G0 = ex[1];
    const double F1 = (tot[0] - flast) + ex[2], G1 = (tot[1] - (double)(SPS - 1 - SPS) * flast) + ex[3];
    double s0 = F1 - F0, s1 = (G1 - G0) + (double)(SPS - i0) * s0;
    float bv = 0.0f; // max_correlation = 0 (:400)
    int bi = 0x7fffffff;
    float b2 = 0.0f; // this thread's second-best shift: a near-tie's other half (lora_strict_sync.inc.hip)
    int i2 = 0x7fffffff;
    if constexpr (CHUNKED) {
#pragma unroll 1
        for (int q = 0; q < LEN; q += 16) {
            w3_sync_ifreq16<ZM>(xv, i0 + q, fa);
            w3_sync_ifreq16<ZM>(xv, i0 + n + q, fb);
#pragma unroll
            for (int rr = 0; rr < 16; rr++) {
                const float c = (float)(sync_a * s0 + sync_b * s1);
                if (c > bv) { b2 = bv; i2 = bi; bv = c; bi = i0 + q + rr; }
                else if (c > b2) { b2 = c; i2 = i0 + q + rr; }
                const double fin = (double)fb[rr], fout = (double)fa[rr];
                s0 += fin - fout;
                s1 += (double)n * fin - s0;
            }
        }
    } else {
#pragma unroll
    for (int rr = 0; rr < LEN; rr++) {
        const float c = (float)(sync_a * s0 + sync_b * s1);
        if (c > bv) { b2 = bv; i2 = bi; bv = c; bi = i0 + rr; }
        else if (c > b2) { b2 = c; i2 = i0 + rr; }
        const double fin = (double)fb[rr], fout = (double)fa[rr];
        s0 += fin - fout;
        s1 += (double)n * fin - s0;
    }
    }
    // first maximum over the workgroup
    float *red = ws.red[slot][0];
    slot ^= 1;
    const float my_bv = bv;
    const int my_bi = bi;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        const float ov = __shfl_xor(bv, o, 64);
        const int oi = __shfl_xor(bi, o, 64);
        if (ov > bv || (ov == bv && oi < bi)) { bv = ov; bi = oi; }
    }
    if (lane == 0) { red[wave] = bv; ((int *)red)[32 + wave] = bi; }
    __syncthreads();
    bv = red[0]; bi = ((int *)red)[32];
    for (int w = 1; w < WAVES; w++) {
        const float ov = red[w];
        const int oi = ((int *)red)[32 + w];
        if (ov > bv || (ov == bv && oi < bi)) { bv = ov; bi = oi; }
    }
    if (strict_u) { // shifts within rounding of the maximum: the reference's own float sums decide (:399-407)
        strict::cands_push(ws.sc, bv, my_bv, my_bi, b2, i2);
        __syncthreads();
        const int nc = ws.sc.n;
        if (nc >= 2 && nc <= strict::kK) {
            float ev;
            bi = strict::resolve<G::T, G::STRICT_CH, true>(x, SPS, strict_u, &ws.sc, strict_buf, &ev);
            bv = ev;
        }
    }
    return W3SyncOut{bv, bi, slot, 0};
}

// ---- FIND_SFD (:385-390, :283-298, :801-803): the windows q = k NG + grp, k < KS, at x0 + q sps ---------------------------
// Pearson correlation of each window's ifreq with the ideal downchirp ifreq (one pass); for an upchirp (c < -0.97)
// fine_sync(-1, 4 D) over the 63 lags in closed form (see w2_sfd_window for the derivation).  c[q] and fine[q] come back uniform.
struct W3SfdOut { float c[8]; int32_t fine[8]; uint32_t pz; }; // pz bit q: window q holds a sample of exactly zero (its sums are NaN): to be evaluated again with ZM = true
struct W3SfdArgs { const float *down_ifreq, *up_ifreq_v; float down_ifreq_avg, down_ifreq_sd, down_ifreq_dsum; double sync_a, sync_b; };
template <int SF, int HV = 0, bool ZM = false>
__device__ LORA_W3_SFD_INLINE W3SfdOut w3_sfd_round(W3SfdArgs P, const float2 *__restrict__ x0, int n_valid, W3AcqScratch *sc)
{
    using G = W3Geom<SF, HV>;
    constexpr int SPS = G::SPS, TG = G::TG, CH = G::CH, PAIRS = G::PAIRS, GW = G::GW, NG = G::NG, KS = W3Acq<SF, HV>::KS, NQ = W3Acq<SF, HV>::NQS;
    const w3_buf_t ddb = w3_buf(P.down_ifreq);
    int tt = threadIdx.x;
    asm volatile("" : "+v"(tt));
    const int grp = __builtin_amdgcn_readfirstlane(tt / TG), t = tt % TG, lane = t & 63, gwave = __builtin_amdgcn_readfirstlane(t >> 6);
#pragma unroll 1
    for (int k = 0; k < KS; k++) {
        const int q = k * NG + grp;
        float a3[3] = {0.f, 0.f, 0.f};
        double g0 = 0.0, g1 = 0.0;
        float f_first = 0.0f, f_last = 0.0f; // chunk 0 of pair 0, chunk 15 of the last pair
        if (q < n_valid) { // (uniform per group)
            const w3_buf_t xb = w3_buf(w3_uniform_ptr(x0 + (int64_t)q * SPS));
#pragma unroll 1
            for (int p = 0; p < PAIRS; p++) {
                const int base = p * TG + t;
                const uint32_t ob = 8u * (uint32_t)base;
                v2f a[16], ap[16];
                float dd[16], f[16];
#pragma unroll
                for (int c = 0; c < 16; c++) a[c] = w3_ld2(xb, ob, (uint32_t)(c * CH * 8));
                ap[0] = w3_ld2(xb, (p == 0) ? (ob >= 8u ? ob - 8u : 0u) : ob - 8u, 0u);
#pragma unroll
                for (int c = 1; c < 16; c++) ap[c] = w3_ld2(xb, ob, (uint32_t)(c * CH * 8 - 8));
                dd[0] = w3_ld1(ddb, base >= 1 ? 4u * (uint32_t)(base - 1) : 0u, 0u);
#pragma unroll
                for (int c = 1; c < 16; c++) dd[c] = w3_ld1(ddb, 4u * (uint32_t)base, (uint32_t)(c * CH * 4 - 4));
                if (p == 0) w3_ifreq16<true, ZM>(a, ap, t == 0, f);
                else w3_ifreq16<false, ZM>(a, ap, false, f);
#pragma unroll
                for (int c = 0; c < 16; c++) {
                    const float fk = f[c];
                    const float d = dd[c] - P.down_ifreq_avg;
                    a3[0] += fk; a3[1] += fk * fk; a3[2] += fk * d; // f is 0 for the non-existent k = -1
                    const int kk = c * CH + base - 1;
                    g0 += (double)fk; g1 += (double)kk * (double)fk;
                }
                if (p == 0) f_first = f[0];
                if (p == PAIRS - 1) f_last = f[15];
            }
            if (t == TG - 1) { g0 += (double)f_last; g1 += (double)(SPS - 1) * (double)f_last; } // duplicated last tap (:243); only the lags use g0, g1
        }
#pragma unroll
        for (int j = 0; j < 3; j++) a3[j] = wave_sum_rows(a3[j]);
        g0 = w3_wave_sum_d(g0); g1 = w3_wave_sum_d(g1);
        if (lane == 0) {
            float *o = sc->part[q][gwave];
            o[0] = a3[0]; o[1] = a3[1]; o[2] = a3[2];
            sc->dpart[q][gwave][0] = g0; sc->dpart[q][gwave][1] = g1;
        }
        // head[k] = fe[k], k < 32 (threads 1 .. 32 of chunk 0); tail[j] = fe[sps-1-j], j <= 32 (the last threads of chunk 15)
        float *scr = sc->edge[q];
        if (t >= 1 && t <= 32) scr[t - 1] = f_first;
        if (t >= TG - 32) scr[32 + 1 + (TG - 1 - t)] = f_last;
        if (t == TG - 1) scr[32] = f_last;
    }
    __syncthreads();
    W3SfdOut R;
    bool any_up = false;
#pragma unroll
    for (int q = 0; q < 8; q++) { R.c[q] = 0.0f; R.fine[q] = 0; }
    R.pz = 0u;
#pragma unroll
    for (int q = 0; q < NQ; q++) {
        float s0 = 0.f, s1 = 0.f, s2 = 0.f;
#pragma unroll
        for (int w = 0; w < GW; w++) { s0 += sc->part[q][w][0]; s1 += sc->part[q][w][1]; s2 += sc->part[q][w][2]; }
        s0 = w3_uni(s0); s1 = w3_uni(s1); s2 = w3_uni(s2);
        if (!ZM && q < n_valid && poisoned(s0)) R.pz |= 1u << q;
        const float nf = (float)(SPS - 1);
        const float average = s0 / nf;
        const float var = fmaxf(s1 / nf - average * average, 0.0f);
        const float sd = sqrtf(var) * P.down_ifreq_sd;
        R.c[q] = (s2 - average * P.down_ifreq_dsum) / sd / nf;
        any_up = any_up || (R.c[q] < -0.97f);
    }
    if (!any_up) return R; // (uniform)
    // fine_sync(-1, 32): c_i = sum_{k<sps} fe[k] v[sps + i + k], i = -31 .. 31, fe[sps-1] = fe[sps-2]; one wavefront per group
    if (gwave == 0) {
#pragma unroll 1
        for (int k = 0; k < KS; k++) {
            const int q = k * NG + grp;
            if (!(R.c[q] < -0.97f)) continue; // (uniform)
            double G0 = 0.0, G1 = 0.0;
            for (int w = 0; w < GW; w++) { G0 += sc->dpart[q][w][0]; G1 += sc->dpart[q][w][1]; }
            const float *scr = sc->edge[q];
            const int i = lane - 31; // this lane's lag
            float ps = scr[lane];
#pragma unroll
            for (int o = 1; o < 32; o <<= 1) {
                const float up = __shfl_up(ps, o, 32);
                if ((lane & 31) >= o) ps += up;
            }
            const float hsum = __shfl(ps, lane < 31 ? 30 - lane : 0, 64); // H_{-i} for the negative lags
            float c_i = -3.0e38f;
            if (lane <= 62) {
                const double a = P.sync_a, b = P.sync_b;
                const double wd = (double)P.up_ifreq_v[2 * SPS - 1] - (a + b * (double)(SPS - 1));
                double edge;
                float wrap_f;
                if (i >= 0) { edge = i > 0 ? -(double)ps : 0.0; wrap_f = scr[32 + i]; }
                else { edge = (double)hsum; wrap_f = scr[-i - 1]; }
                c_i = (float)(a * G0 + b * ((double)i * G0 + G1) + b * (double)SPS * edge + wd * (double)wrap_f);
            }
            int li = lane;
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) { // first maximum in lag order (strict '>' scan from 0, :311)
                const float ov = __shfl_xor(c_i, o, 64);
                const int oi = __shfl_xor(li, o, 64);
                if (ov > c_i || (ov == c_i && oi < li)) { c_i = ov; li = oi; }
            }
            const int32_t lag = (c_i > 0.0f) ? li - 31 : 0;
            if (lane == 0) sc->lag[q] = lag;
        }
    }
    __syncthreads();
#pragma unroll
    for (int q = 0; q < NQ; q++) R.fine[q] = (R.c[q] < -0.97f) ? -sc->lag[q] : 0;
    return R;
}

// ---- thread-0 bookkeeping ---------------------------------------------------------------------------------------
// everything demodulate() / work() do once the bin is known (:506-529, :826-886), explicit or implicit header.
// Returns true when the payload is complete: the caller requests the workgroup-wide finalisation round.
template <bool FAST_MOD>
__device__ __forceinline__ bool w3_post_symbol(const DevParams &P, W2State &S, Shared &sh, bool do_demod, uint32_t bin_idx, bool is_first)
{
    bool block_done = false;
    if (do_demod) {
        const bool reduced = is_first || P.reduced_rate; // :495
        if (reduced) { // :507-509 (% N/4: a power of two, of a value >= 0)
            if constexpr (FAST_MOD) bin_idx = (uint32_t)lroundf((float)bin_idx / 4.0f) & (P.nbins_hdr - 1u);
            else bin_idx = (uint32_t)(lroundf((float)bin_idx / 4.0f) % (long)P.nbins_hdr);
        }
        const uint32_t word = bin_idx ^ (bin_idx >> 1u); // :512
        const uint32_t need = 4u + (is_first ? 4u : S.cr); // :521
        if (S.n_words < 16u) sh.words[S.n_words] = word;
        S.n_words++;
        S.n_sym++;
        if (S.n_words == need) {
            const uint32_t ppm = reduced ? P.sf - 2u : P.sf;
            uint32_t tmp = S.n_cw;
            deinterleave_block(sh, need, ppm, tmp);
            S.n_cw = (S.n_cw + ppm <= (uint32_t)kMaxCodewords) ? S.n_cw + ppm : (uint32_t)kMaxCodewords;
            S.n_words = 0;
            block_done = true;
        }
    }
    if (is_first) {
        if (block_done) {
            if (P.implicit) {
                S.payload_symbols = 1; // :829
            } else { // decode(true) and header parse (:831-847)
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
                S.payload_length = (uint32_t)S.phdr[0] + 2u * S.has_crc; // MAC_CRC_SIZE (:838)
                const uint32_t redundancy = P.reduced_rate ? 2u : 0u; // :842-847
                const int symbols_per_block = (int)S.cr + 4;
                const float bits_needed = (float)S.payload_length * 8.0f;
                const float symbols_needed = bits_needed * ((float)symbols_per_block / 4.0f) / (float)(P.sf - redundancy);
                const int blocks_needed = (int)ceilf(symbols_needed / (float)symbols_per_block);
                S.payload_symbols = blocks_needed * symbols_per_block;
            }
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

// L2 touch of the symbol a group will most likely evaluate in the NEXT round (one dword per 128-byte line and thread; the value
// is never used).  A round reads its NG windows in one burst at its very start - every CU at about the same time, so the burst
// runs at the HBM limit (~11 B/clk/CU) while the rest of the round moves nothing; touched a round ahead, the lines come from L2 /
// MALL instead.  The value must stay live until the data has landed (the caller consumes it at the top of the next round).
template <int SF, int HV = 0>
__device__ __forceinline__ void w3_touch(const float2 *X, int64_t first_item, int64_t fallback_item, int64_t n_items, float (&v)[4])
{
    using G = W3Geom<SF, HV>;
    // (always a load - past the end of the stream it re-reads the current window - and nothing done to the value: any use,
    // even a select against 0, makes the compiler wait for it right here)
    const int64_t fi = first_item + (int64_t)G::SPS <= n_items ? first_item : fallback_item; // uniform per group
    const w3_buf_t xb = w3_buf(w3_uniform_ptr(X + fi));
    const uint32_t t = threadIdx.x % G::TG;
#pragma unroll
    for (int p = 0; p < G::PAIRS; p++) v[p] = w3_ld1(xb, 128u * ((uint32_t)(p * G::TG) + t), 0u);
}

// ---- the kernel -----------------------------------------------------------------------------------------------
// ROUNDS.  The decoder state (W2State, as in walker2) lives in LDS and belongs to thread 0.  Every round starts from a
// PLAN (position, what to evaluate, how many windows): in DETECT, FIND_SFD and DECODE_* the NG groups evaluate the NG
// upcoming windows pos + g sps (zero drift assumed) - the results come back uniform in every thread - and thread 0
// replays the reference's per-call logic over them in order on a register copy of the state, stopping at the first
// outcome that invalidates the later windows (a trigger, a state change, d_fine_sync != 0, a loop-top check, the end of
// the data); it then writes the state back and the next plan.  The accepted sequence is exactly the serial one.  Keeping
// the state out of the other wavefronts' registers is what lets the demodulator run without spills.
// SKIP (LaunchCfg.skip_payload): the header-only variant.  A packet's attempt ends behind its header: the record (kAttemptHeaderOnly) carries d_phdr, the
// header block's spare codewords and d_payload_symbols, and the job goes on in DETECT where DECODE_PAYLOAD would end if no symbol moved the symbol
// clock.  The payload symbols are demodulated by the payload pass - all of them at once, over the whole device - and the host checks the assumption.
// ---- the ZM evaluation of a decode round (a window of the previous round holds a sample of exactly zero: kFinePoison), out of line: it runs in a round of its
// own, for the rare window only, and must neither grow the ordinary round's code nor take part in its register allocation ----------------------------------
#ifndef LORA_W3_ZM_ATTR
#define LORA_W3_ZM_ATTR __attribute__((noinline)) // (inlined into the round loop it cost SF9-SF12 another 1.5-4 %: profiles/r05_ab_zero_samples.txt)
#endif
template <int SF, int HV>
__device__ LORA_W3_ZM_ATTR W3DemodOut w3_demod_round_zm(W3DemodArgs DA, W3Lds<SF, HV> L, const float2 *x0, int64_t gstride /* group g's window: x0 + g gstride */, uint32_t vmask, bool want_energy, int slot)
{
    constexpr int NG = W3Geom<SF, HV>::NG;
    uint32_t sq[NG];
    int32_t fq[NG];
    float eq[NG];
    const float2 *xa[NG];
#pragma unroll
    for (int g = 0; g < NG; g++) xa[g] = x0 + (int64_t)g * gstride;
    w3_demod_round<SF, HV, true>(DA, L, xa, vmask, want_energy, slot, sq, fq, eq);
    W3DemodOut o{};
#pragma unroll
    for (int g = 0; g < NG; g++) { o.s[g] = sq[g]; o.fine[g] = fq[g]; o.en[g] = eq[g]; }
    o.slot = slot;
    return o;
}

template <int SF, bool GRAD, int HV = 0, bool SKIP = false>
__device__ __forceinline__ void walker3_body(const DevParams &P, const LaunchCfg &C)
{
    using G = W3Geom<SF, HV>;
    constexpr uint32_t sps = G::SPS;
    constexpr int T = G::T, NG = G::NG;
    // windows a decode round evaluates: one per GROUP with the FFT demodulators (the pruned DFT needs the group's LDS array), one per WAVEFRONT with the
    // gradient demodulator (w3_wave_window_grad: no FFT, nothing crosses a wavefront) - 8 at every spreading factor (4 in the half-size workgroups)
    constexpr bool WFFT = G::WFFT && !GRAD; // SF9: the FFT demodulator one window per wavefront as well (wave_demod_symbol<9>)
    constexpr int NWIN = (GRAD || WFFT) ? T / 64 : NG;
    static_assert(NWIN <= 16, "W3Shared::wres");
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const W3Lds<SF, HV> L = w3_carve<SF, HV>(smem);
    W3Shared &ws = *L.ws;
    Shared &sh = ws.sh;
    W2State &S = ws.st;

    const uint32_t jid = blockIdx.x;
    if (jid >= C.n_jobs) return;
    Job job = C.jobs[jid];             // (phase 1 rewrites start / limits: the job carries on as the next segment's probe)
    if constexpr (G::UNIFORM_JOB) job = uniform_job(job);
    uint32_t rec_cap = C.recs_per_job; // ... with what is left of the attempt-record capacity
    const float2 *__restrict__ X = C.iq + job.stream_off;
    const int64_t n_items = (int64_t)job.stream_len;
    AttemptRec *recs = C.recs + (size_t)jid * C.recs_per_job;
    StepRec *trace = C.trace ? C.trace + (size_t)jid * C.trace_cap : nullptr;
    const bool t0 = threadIdx.x == 0;
    int slot = 0;
    const W3DemodArgs DA{P.down, P.w3_ctab, P.up_ifreq_v, P.enable_fine_sync, P.demod_mode, P.ffs_on, P.ffs_alpha, P.ffs_jump, P.ffs_tol};

    WaveTabs WT{};
    if constexpr (WFFT) WT = wave_tabs_to_lds<SF>(P, smem + G::kWfftScratch, (uint32_t)T); // (visible behind the round loop's first barrier)
    else w3_tables_to_lds<SF, HV>(P, L);

    // plan for the next round from the TRUE state (thread 0 only)
    auto plan_from = [&](W2State &St, W2Plan &pl, bool zreq = false /* a window of this round came back poisoned (a sample of exactly zero): the next round is a ZM one */) {
        pl.buf = 0; pl.resolve_prev = 0; pl.n_win = NG; pl.pos = St.pos; pl.prev_n = 0; pl.zmode = zreq ? 1 : 0;
        if (!St.done) (void)w2_pre_step(St, job, rec_cap, sps);
        if (St.done) { pl.mode = kPlanExit; return; }
        if (St.fin_pending) { pl.mode = kPlanFinalize; return; }
        switch (St.state) {
        case kDetect: pl.mode = kPlanDetect; break;
        case kSync: pl.mode = kPlanSync; break;
        case kFindSfd: pl.mode = kPlanSfd; break;
        case kPause: pl.mode = kPlanPause; break;
        default:
            pl.mode = kPlanDecode; pl.n_win = NWIN;
            if (St.state == kDecodePayload && !P.implicit) { // symbols left in the packet (:866-870)
                const int32_t rem = St.payload_symbols - (int32_t)St.n_words;
                pl.n_win = rem < NWIN ? (rem > 0 ? rem : 1) : NWIN;
            }
            break;
        }
    };

    bool stop_sfd = false; // phase 1 with Job.tail_stop_sfd: the probe stops behind its first FIND_SFD step
    for (int phase = 0; phase < 2; phase++) {
    if (t0) {
        S = W2State{};
        S.state = kDetect; S.pos = job.start; S.cr = job.cr_prev; S.has_crc = P.ctor_crc;
        S.phdr[1] = (uint8_t)((P.ctor_cr << 5) | (P.ctor_crc << 4));
        S.att_start = job.start; S.att_trig = -1; S.att_hdr = -1; S.att_cr_prev = job.cr_prev;
        sh.n_sfd = 0u;
        if (job.start_at_header && phase == 0 && rec_cap > 0u) recs[0].n_sfd = 0u;
        if (job.start_at_header && phase == 0) { S.state = kDecodeHeader; S.in_attempt = 1; S.att_trig = job.start; S.att_hdr = job.start; } // (acquired elsewhere)
        if (phase == 0) ws.stats = W2Stats{};
        ws.stats.prev_state = -1;
        plan_from(S, ws.plan[0]);
    }

    for (uint32_t it = 0;; it++) {
        __syncthreads(); // plan[it & 1] and everything thread 0 wrote are visible; plan[(it + 1) & 1] is free
        const W2Plan &pl_in = ws.plan[it & 1u];
        const uint64_t pl_pp = (uint64_t)pl_in.pos;
        const int64_t pos = (int64_t)(((uint64_t)__builtin_amdgcn_readfirstlane((uint32_t)(pl_pp >> 32)) << 32) | __builtin_amdgcn_readfirstlane((uint32_t)pl_pp));
        const int32_t plan_mode = __builtin_amdgcn_readfirstlane(pl_in.mode), plan_n_win = __builtin_amdgcn_readfirstlane(pl_in.n_win);
        const int32_t plan_z = __builtin_amdgcn_readfirstlane(pl_in.zmode); // kPlanDecode / kPlanSfd: this round's windows by the ZM instantiations (W2Plan.zmode)
        W2Plan &next = ws.plan[(it + 1u) & 1u];
        if (plan_mode == kPlanExit) break;
        const long long t_start = clock64();
        if (t0) {
            W2Stats &Q = ws.stats;
            const int sidx = plan_mode == kPlanDetect ? 0 : plan_mode == kPlanSync ? 1 : plan_mode == kPlanSfd ? 2 : plan_mode == kPlanPause ? 3 : 5;
            if (Q.prev_state >= 0) { Q.cyc[Q.prev_state] += (uint32_t)((t_start - Q.prev_t) >> 6); Q.rounds[Q.prev_state]++; }
            Q.prev_state = sidx; Q.prev_t = t_start;
        }

        // windows of this round that lie inside the data (:91): the first n_in_data of pos, pos + sps, ...
        const int64_t fit64 = (n_items - pos) / (int64_t)sps - 1;
        const int n_in_data = fit64 < 0 ? 0 : (fit64 > 64 ? 64 : (int)fit64);
        W3AcqScratch *acq = reinterpret_cast<W3AcqScratch *>(L.data);

        if (plan_mode == kPlanDetect) { // :752-768, detect_preamble_autocorr :340-366
            constexpr int NQ = W3Acq<SF, HV, SKIP>::NQD;
            float a[NQ][4];
            w3_detect_round<SF, HV, SKIP>(X + pos, n_in_data < NQ ? n_in_data : NQ, acq, a);
            if (t0) {
                W2State St = S;
                for (int g = 0; g < NQ; g++) {
                    if (g > 0 && (St.state != kDetect || !w2_pre_step(St, job, rec_cap, sps))) break;
                    float a0 = a[0][0], a1 = a[0][1], a2 = a[0][2], a3 = a[0][3];
#pragma unroll
                    for (int q = 1; q < NQ; q++) if (g == q) { a0 = a[q][0]; a1 = a[q][1]; a2 = a[q][2]; a3 = a[q][3]; }
                    St.energy_threshold = a3 / 2.0f; // :357
                    const float pushed = a2 / (float)sps; // d_pwr_queue.push_back (:360)
                    if (St.npush >= 4u) { St.push_tail[0] = St.push_tail[1]; St.push_tail[1] = St.push_tail[2]; St.push_tail[2] = St.push_tail[3]; St.push_tail[3] = pushed; }
                    else {
                        const uint32_t k = St.npush;
                        if (k == 0u) St.push_tail[0] = pushed; else if (k == 1u) St.push_tail[1] = pushed; else if (k == 2u) St.push_tail[2] = pushed; else St.push_tail[3] = pushed;
                    }
                    St.npush++;
                    const float sq = sqrtf(a2 * a3);
                    const float autocorr = hypotf(a0 / sq, a1 / sq); // :363
                    int32_t consumed = 0;
                    if (autocorr >= 0.90f) { // :755
                        St.corr_fails = 0u;
                        St.state = kSync;
                        St.in_attempt = 1;
                        St.att_trig = St.pos; St.att_hdr = -1; St.att_cr_prev = St.cr; St.att_ambig = 0; St.n_sym = 0;
                        sh.n_sfd = 0u;
                        if (St.n_att < rec_cap) recs[St.n_att].n_sfd = 0u;
                    } else {
                        consumed = (int32_t)sps;
                    }
                    w2_end_step<true>(St, job, C, recs, trace, kDetect, consumed, -1, 0, autocorr, t_start);
                    if (St.done) break;
                }
                W2Plan np;
                plan_from(St, np);
                next = np; S = St;
            }
            continue;
        }

        if (plan_mode == kPlanSync) { // :770-783
            W3SyncOut so = w3_sync<SF, HV>(P.sync_a, P.sync_b, X + pos, &ws, slot, P.strict_sync ? P.up_ifreq : nullptr, reinterpret_cast<float *>(L.data));
            if (__builtin_amdgcn_readfirstlane(so.pz)) { // a sample of exactly zero in the window (the closed form's sums are NaN): once more, every ifreq value as the reference forms it
                __syncthreads(); // (the scan's exchange area is written again)
                so = w3_sync<SF, HV, true>(P.sync_a, P.sync_b, X + pos, &ws, slot, P.strict_sync ? P.up_ifreq : nullptr, reinterpret_cast<float *>(L.data));
            }
            slot = __builtin_amdgcn_readfirstlane(so.slot);
            if (t0) {
                W2State St = S;
                const int32_t consumed = (so.bi == 0x7fffffff) ? 0 : so.bi; // `int i = 0` stays when nothing exceeds 0 (:771)
                St.state = kFindSfd;
                w2_end_step<true>(St, job, C, recs, trace, kSync, consumed, -1, 0, so.bv, t_start);
                W2Plan np;
                plan_from(St, np);
                next = np; S = St;
            }
            continue;
        }

        if (plan_mode == kPlanSfd) { // :785-818
            constexpr int NQ = W3Acq<SF, HV>::NQS;
            const W3SfdArgs sfa{P.down_ifreq, P.up_ifreq_v, P.down_ifreq_avg, P.down_ifreq_sd, P.down_ifreq_dsum, P.sync_a, P.sync_b};
            const W3SfdOut fo = plan_z ? w3_sfd_round<SF, HV, true>(sfa, X + pos, n_in_data < NQ ? n_in_data : NQ, acq)
                                       : w3_sfd_round<SF, HV>(sfa, X + pos, n_in_data < NQ ? n_in_data : NQ, acq);
            if (t0) {
                W2State St = S;
                bool zreq = false;
                for (int g = 0; g < NQ; g++) {
                    if (g > 0 && (St.state != kFindSfd || !w2_pre_step(St, job, rec_cap, sps))) break;
                    if ((fo.pz >> g) & 1u) { zreq = true; break; } // a sample of exactly zero in this window: it opens a round of ZM evaluations
                    // a tail probe with Job.tail_stop_sfd has seen enough once it stands at the start of its SECOND FIND_SFD step: the first
                    // one's fine_sync(-1, 4 D) (:801-803) has pulled it onto the chirp boundary its successor's own attempt passes through,
                    // and (position, d_corr_fails) is all that :785-818 read - the stitch matches it against that attempt's record
                    const uint32_t nsf = sh.n_sfd; // (thread 0's own counter, in LDS: not part of the register-resident state)
                    if (stop_sfd && nsf >= 1u) { St.stop_reason = 4; St.done = 1; break; }
                    if (St.n_att < rec_cap && nsf < (uint32_t)kMaxSfdRec) { // the state this step starts in, for such a match
                        recs[St.n_att].sfd_pos[nsf] = St.pos;
                        recs[St.n_att].sfd_fails[nsf] = (uint8_t)St.corr_fails;
                        recs[St.n_att].n_sfd = nsf + 1u;
                    }
                    sh.n_sfd = nsf + 1u;
                    float c = fo.c[0];
                    int32_t fs = fo.fine[0];
#pragma unroll
                    for (int q = 1; q < NQ; q++) if (g == q) { c = fo.c[q]; fs = fo.fine[q]; }
                    int32_t fine = 0;
                    if (c > 0.96f) { // :792
                        St.state = kPause;
                    } else {
                        if (c < -0.97f) fine = fs; // :801-803
                        else St.corr_fails++;
                        if (St.corr_fails > 4u) St.state = kDetect; // :808-809
                    }
                    w2_end_step<true>(St, job, C, recs, trace, kFindSfd, (int32_t)sps + fine, -1, fine, c, t_start); // :816
                    if (St.done || fine != 0) break; // (fine != 0: the later windows started at the wrong sample)
                }
                // PAUSE (:820-824) looks at no sample: the step is taken here, behind its own loop-top checks, instead of in a
                // round of its own (as walker2 does)
                if (St.state == kPause && !St.done && w2_pre_step(St, job, rec_cap, sps)) {
                    St.state = kDecodeHeader;
                    const int32_t consumed = (int32_t)(sps + P.delay_after_sync);
                    St.att_hdr = St.pos + consumed;
                    w2_end_step<true>(St, job, C, recs, trace, kPause, consumed, -1, 0, 0.0f, t_start);
                }
                W2Plan np;
                plan_from(St, np, zreq);
                next = np; S = St;
            }
            continue;
        }

        if (plan_mode == kPlanPause) { // :820-824
            if (t0) {
                W2State St = S;
                St.state = kDecodeHeader;
                const int32_t consumed = (int32_t)(sps + P.delay_after_sync);
                St.att_hdr = St.pos + consumed;
                w2_end_step<true>(St, job, C, recs, trace, kPause, consumed, -1, 0, 0.0f, t_start);
                W2Plan np;
                plan_from(St, np);
                next = np; S = St;
            }
            continue;
        }

        if (plan_mode == kPlanFinalize) { // decode(false) + frame bytes (:870-881), all threads
            const uint32_t n_cw = S.n_cw, cr = S.cr, n_bytes = S.fin_n_bytes, plen = S.fin_plen, n_att = S.n_att;
            decode_payload_bytes(sh, n_cw, cr, n_bytes);
            AttemptRec &r = recs[n_att];
            for (uint32_t i = threadIdx.x; i < plen; i += (uint32_t)T) r.frame[3u + i] = (i < n_bytes) ? sh.dec[i] : 0;
            __syncthreads();
            if (t0) {
                W2State St = S;
                r.frame[0] = St.phdr[0]; r.frame[1] = St.phdr[1]; r.frame[2] = St.phdr[2]; // d_phdr (:600)
                r.frame_len = 3u + plen;
                St.frame_ok = 1;
                St.state = kDetect;
                St.n_words = 0; St.n_cw = 0;
                St.fin_pending = 0;
                w2_end_step<true>(St, job, C, recs, trace, St.fin_st, St.fin_consumed, St.fin_bin, St.fin_fine, 0.0f, t_start);
                W2Plan np;
                plan_from(St, np);
                next = np; S = St;
            }
            continue;
        }

        // ---- kPlanDecode: DECODE_HEADER / DECODE_PAYLOAD rounds (:826-886, demodulate :493-529) ----
        {
            uint32_t sq[NWIN];
            int32_t fq[NWIN];
            float eq[NWIN];
            if constexpr (GRAD || WFFT) { // one window per wavefront: no barrier until the results are in LDS
                const int wave = __builtin_amdgcn_readfirstlane((int)threadIdx.x >> 6);
                const int64_t wpos = pos + (int64_t)wave * sps;
                const bool wvalid = wave < plan_n_win && wpos + 2 * (int64_t)sps <= n_items;
                uint32_t wb = 0u;
                int32_t wf = 0;
                float we = 0.0f;
                if constexpr (GRAD) {
                if (wvalid) {
                    float *fcache = reinterpret_cast<float *>(L.data) + wave * (kW3GradCacheChunks * 1024); // (the FFT demodulators' LDS array, idle in these kernels)
                    if (plan_z) w3_wave_window_grad<SF, true>(DA, X + wpos, P.implicit != 0u, wb, wf, we, fcache); // (uniform) a round of ZM evaluations (W2Plan.zmode)
                    else w3_wave_window_grad<SF>(DA, X + wpos, P.implicit != 0u, wb, wf, we, fcache);            // wf = kFinePoison: a sample of exactly zero in the window
                }
                } else {
                if (wvalid) {
                    float *en_p = P.implicit != 0u ? &we : nullptr;
                    if (plan_z) wave_demod_symbol<SF, 1, true>(P, WT, X + wpos, wb, wf, en_p);
                    else wave_demod_symbol<SF, kWaveFmode<SF>>(P, WT, X + wpos, wb, wf, en_p);
                    if (wb == kPoisonBin) wf = kFinePoison; // a sample of exactly zero in the window: the replay asks for a ZM round
                }
                }
                if ((threadIdx.x & 63u) == 0u) { ws.wres[wave][0] = (int32_t)wb; ws.wres[wave][1] = wf; ws.wres[wave][2] = __builtin_bit_cast(int32_t, we); ws.wres[wave][3] = wvalid ? 1 : 0; }
                __syncthreads();
                if (t0) {
#pragma unroll
                    for (int g = 0; g < NWIN; g++) { sq[g] = (uint32_t)ws.wres[g][0]; fq[g] = ws.wres[g][1]; eq[g] = __builtin_bit_cast(float, ws.wres[g][2]); }
#pragma unroll
                    for (int g = 0; g < NWIN; g++) { // (thread 0's replay lives in scalar registers)
                        sq[g] = (uint32_t)__builtin_amdgcn_readfirstlane((int)sq[g]); fq[g] = __builtin_amdgcn_readfirstlane(fq[g]); eq[g] = w3_uni(eq[g]);
                    }
                }
            } else {
            // every group's window and whether it has one (uniform): group g reads at pos + g sps while that lies inside the data (:91), up to plan_n_win windows
            uint32_t dmask = 0u;
            const float2 *xall[NG];
#pragma unroll
            for (int g = 0; g < NG; g++) {
                const bool gv = pos + (int64_t)(g + 2) * sps <= n_items && g < plan_n_win;
                dmask |= gv ? (1u << g) : 0u;
                xall[g] = X + (gv ? pos + (int64_t)g * sps : pos);
            }
            if (plan_z) { // (uniform) a round of ZM evaluations: a window of the previous round holds a sample of exactly zero
                const W3DemodOut zo = w3_demod_round_zm<SF, HV>(DA, L, X + pos, (int64_t)sps, dmask, P.implicit != 0u, slot);
#pragma unroll
                for (int g = 0; g < NG; g++) {
                    sq[g] = (uint32_t)__builtin_amdgcn_readfirstlane((int)zo.s[g]); fq[g] = __builtin_amdgcn_readfirstlane(zo.fine[g]); eq[g] = w3_uni(zo.en[g]);
                }
                slot = __builtin_amdgcn_readfirstlane(zo.slot);
            } else
            w3_demod_round<SF, HV>(DA, L, xall, dmask, P.implicit != 0u, slot, sq, fq, eq); // fq = kFinePoison: see W2Plan.zmode
            }
            if (t0) {
                bool zreq = false;
#if LORA_W3_REPLAY_STATS
                const long long tr0 = clock64(); // (LORA_HIP_DEBUG accounting: ctl[0] = the demodulation, ctl[1] = thread 0's replay, of the decode rounds)
                ws.stats.ctl[0] += (uint32_t)((tr0 - t_start) >> 6);
#endif
                W2State St = S;
#if LORA_W3_REPLAY_STATS > 1 // (finer: ctl[2] = the state copy-in, ctl[3] = the symbol loop)
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                const long long tr1 = clock64();
                ws.stats.ctl[2] += (uint32_t)((tr1 - tr0) >> 6);
#endif
                // The common round - NG payload symbols inside the data, none of which moves the symbol clock - taken on its own:
                // what the general loop below does with them is the symbol's word, the block's deinterleave when it is full and
                // three counters, but as one loop over every case (header, implicit mode, traces, limits, a break per outcome) the
                // compiler carries the whole state through ~300 instructions per symbol: 1.7 k clocks each on the one thread
                // that runs them, with the workgroup waiting (LORA_W3_REPLAY_STATS: 6.7 k of a 50 k round at SF9).
                bool fast = !trace && P.implicit == 0u && St.state == kDecodePayload && plan_n_win == NWIN &&
                            pos + (int64_t)(NWIN + 1) * (int64_t)sps <= n_items; // (every window passes the loop-top check, :91)
#pragma unroll
                for (int q = 0; q < NWIN; q++) fast = fast && fq[q] == 0;
                if (fast) {
#pragma unroll
                    for (int g = 0; g < NWIN; g++) {
                        const uint32_t sg = sq[g];
                        uint32_t bin_idx;
                        if constexpr (GRAD) bin_idx = sg;
                        else if constexpr (G::FAST_MOD) bin_idx = (sg == 0u && P.demod_mode == 2u) ? 0u : (sg + (uint32_t)G::N - 1u) % (uint32_t)G::N;
                        else bin_idx = (sg == 0u && P.demod_mode == 2u) ? 0u : (sg + P.nbins - 1u) % P.nbins;
                        if (w3_post_symbol<G::FAST_MOD>(P, St, sh, true, bin_idx, false)) { // payload complete
                            St.fin_pending = 1; St.fin_st = kDecodePayload; St.fin_consumed = (int32_t)sps; St.fin_bin = (int32_t)bin_idx; St.fin_fine = 0;
                            break;
                        }
                        St.n_steps++; // w2_end_step of a payload step without a trace
                        St.pos += (int64_t)sps;
                    }
                } else
                for (int g = 0; g < NWIN; g++) {
                    if (g >= plan_n_win) break;
                    if (g > 0 && (!(St.state == kDecodeHeader || St.state == kDecodePayload) || !w2_pre_step(St, job, rec_cap, sps))) break;
                    uint32_t sg = sq[0];
                    int32_t fg = fq[0];
                    float eg = eq[0];
#pragma unroll
                    for (int q = 1; q < NWIN; q++) if (g == q) { sg = sq[q]; fg = fq[q]; eg = eq[q]; }
                    if (fg == kFinePoison) { zreq = true; break; } // a sample of exactly zero in this window: it opens a round of ZM evaluations
                    const bool is_first = St.state == kDecodeHeader;
                    const int32_t st_w = St.state;
                    bool do_demod = true;
                    if (!is_first && P.implicit && eg < St.energy_threshold) { St.payload_symbols = 0; St.payload_length = St.n_cw / 2u; do_demod = false; } // :861-864
                    uint32_t bin_idx = 0;
                    int32_t fine = 0, step_bin = -1;
                    if (do_demod) { // :500, bin_idx = (s-1) mod N; compat keeps the s==0 -> 0 quirk of the gradient path
                        if constexpr (GRAD) bin_idx = sg;
                        else if constexpr (G::FAST_MOD) bin_idx = (sg == 0u && P.demod_mode == 2u) ? 0u : (sg + (uint32_t)G::N - 1u) % (uint32_t)G::N;
                        else bin_idx = (sg == 0u && P.demod_mode == 2u) ? 0u : (sg + P.nbins - 1u) % P.nbins;
                        step_bin = (int32_t)bin_idx;
                        fine = fg; // :501-502
                    }
                    if (w3_post_symbol<G::FAST_MOD>(P, St, sh, do_demod, bin_idx, is_first)) { // payload complete: finalise with all threads
                        St.fin_pending = 1; St.fin_st = st_w; St.fin_consumed = (int32_t)sps + fine; St.fin_bin = step_bin; St.fin_fine = fine;
                        break;
                    }
                    w2_end_step<true>(St, job, C, recs, trace, st_w, (int32_t)sps + fine, step_bin, fine, 0.0f, t_start); // :856,:883
                    if constexpr (SKIP) {
                        if (is_first && St.state == kDecodePayload && !St.done) { // header parsed (:831-847): the payload is the payload pass's
                            AttemptRec &r = recs[St.n_att];
                            SkippedPayload sk;
                            sk.phdr[0] = St.phdr[0]; sk.phdr[1] = St.phdr[1]; sk.phdr[2] = St.phdr[2];
                            sk.n_left = (uint8_t)(St.n_cw < 8u ? St.n_cw : 8u);
                            for (uint32_t i = 0; i < 8u; i++) sk.left[i] = i < St.n_cw ? sh.cw[i] : (uint8_t)0;
                            sk.payload_symbols = St.payload_symbols;
                            *reinterpret_cast<SkippedPayload *>(r.frame) = sk;
                            const int32_t n_walk = St.payload_symbols > 0 ? St.payload_symbols : 1; // (:866-870 are reached behind the first symbol at the earliest)
                            St.state = kDetect; St.frame_ok = 0; St.n_words = 0; St.n_cw = 0;
                            w2_end_step<true>(St, job, C, recs, nullptr, kDecodePayload, n_walk * (int32_t)sps, -1, 0, 0.0f, t_start); // (the attempt is closed: status, positions, pushes)
                            r.status = kAttemptHeaderOnly; r.frame_len = (uint32_t)sizeof(SkippedPayload);
                            break;
                        }
                    }
                    if (St.done || fine != 0) break; // (fine != 0: the later windows started at the wrong sample)
                }
#if LORA_W3_REPLAY_STATS > 1
                ws.stats.ctl[3] += (uint32_t)((clock64() - tr1) >> 6);
#endif
                W2Plan np;
                plan_from(St, np, zreq);
                next = np; S = St;
#if LORA_W3_REPLAY_STATS
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                ws.stats.ctl[1] += (uint32_t)((clock64() - tr0) >> 6);
#endif
            }
        }
    }

    // an attempt cut short (out of data, or probe stop) is reported but not counted as complete
    __syncthreads();
    if (t0) {
        const bool in_attempt = S.in_attempt != 0;
        if (in_attempt && S.n_att < rec_cap) {
            AttemptRec &r = recs[S.n_att];
            r.status = (S.stop_reason == 3) ? kAttemptAtHeader : (S.stop_reason == 4) ? kAttemptAtSfd : kAttemptOutOfData;
            r.start_pos = S.att_start; r.trig_pos = S.att_trig; r.hdr_pos = S.att_hdr; r.end_pos = S.pos;
            r.npush = S.npush;
            for (int i = 0; i < 4; i++) r.push_tail[i] = S.push_tail[i];
            r.cr_prev = S.att_cr_prev; r.hdr_ambig = S.att_ambig; r.n_symbols = S.n_sym; r.frame_len = 0;
            uint32_t ns = sh.n_sfd < (uint32_t)kMaxSfdRec ? sh.n_sfd : (uint32_t)kMaxSfdRec;
            if (S.stop_reason == 4 && ns < (uint32_t)kMaxSfdRec) { r.sfd_pos[ns] = S.pos; r.sfd_fails[ns] = (uint8_t)S.corr_fails; ns++; } // the state the probe stopped in
            r.n_sfd = ns;
            if (S.stop_reason == 4) S.stop_reason = 3; // (reported like any other probe stop)
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
            // Job.probe_limit: having reached its scan limit the workgroup runs what a separate probe job started from its end state
            // would run - a FRESH job (state re-initialised, tables kept) that stops at the next header - and reports it as the tail
            go = job.probe_limit > job.scan_limit && S.stop_reason == 0 && !in_attempt && !trace && e_natt < C.recs_per_job;
            ws.ph_go = go ? 1u : 0u; ws.ph_start = e_pos; ws.ph_cr = S.cr; ws.ph_natt = e_natt;
        } else {
            jr.tail_valid = 1; jr.tail_first_rec = ws.ph_natt;
            jr.tail_final_pos = e_pos; jr.tail_n_attempts = e_natt; jr.tail_final_cr = S.cr; jr.tail_npush = e_npush;
            for (int i = 0; i < 4; i++) jr.tail_push_tail[i] = S.push_tail[i];
            jr.tail_stop_reason = (uint32_t)S.stop_reason; jr.tail_pad = e_pad; jr.tail_rsv = 0;
        }
        if (!go) { // last phase of this job: the per-state time accounting goes out with it
            W2Stats &Q = ws.stats;
            if (Q.prev_state >= 0) { Q.cyc[Q.prev_state] += (uint32_t)((clock64() - Q.prev_t) >> 6); Q.rounds[Q.prev_state]++; }
            for (int i = 0; i < 6; i++) { jr.cyc[i] = Q.cyc[i]; jr.rounds[i] = Q.rounds[i]; }
            for (int i = 0; i < 4; i++) jr.ctl[i] = Q.ctl[i];
            for (int i = 0; i < 6; i++) jr.dbg[i] = 0;
        }
    }
    if (phase == 1) break;
    __syncthreads();
    if (!ws.ph_go) break;
    { // the probe: a job from the end state of the job proper to the next header (what decode_streams would launch)
        const int64_t st = ws.ph_start;
        const uint32_t lo = __builtin_amdgcn_readfirstlane((uint32_t)st), hi = __builtin_amdgcn_readfirstlane((uint32_t)((uint64_t)st >> 32));
        const uint32_t natt = __builtin_amdgcn_readfirstlane(ws.ph_natt);
        job.start = (int64_t)(((uint64_t)hi << 32) | lo);
        job.cr_prev = __builtin_amdgcn_readfirstlane(ws.ph_cr);
        job.scan_limit = job.probe_limit; job.stop_at_header = 1; job.max_attempts = 0;
        stop_sfd = job.tail_stop_sfd != 0u;
        recs += natt;
        rec_cap = C.recs_per_job - natt;
    }
    __syncthreads(); // everyone has read the hand-over before thread 0 re-initialises the state
    } // phase
}

__global__ __launch_bounds__(W3Geom<9>::T, W3Geom<9>::T512 ? 2 : 4) void walker3_kernel_sf9(DevParams P, LaunchCfg C) { walker3_body<9, false>(P, C); }
__global__ __launch_bounds__(W3Geom<10>::T, W3Geom<10>::T512 ? 2 : 4) void walker3_kernel_sf10(DevParams P, LaunchCfg C) { walker3_body<10, false>(P, C); }
__global__ __launch_bounds__(W3Geom<11>::T, W3Geom<11>::T512 ? 2 : 4) void walker3_kernel_sf11(DevParams P, LaunchCfg C) { walker3_body<11, false>(P, C); }
__global__ __launch_bounds__(W3Geom<12>::T, W3Geom<12>::T512 ? 2 : 4) void walker3_kernel_sf12(DevParams P, LaunchCfg C) { walker3_body<12, false>(P, C); }
// half-size workgroups, two per CU (W3Geom HV = 1): launched when a pass holds more jobs than full-size workgroups fit at once
__global__ __launch_bounds__((W3Geom<10, 1>::T), 2) void walker3_kernel_sf10_half(DevParams P, LaunchCfg C) { walker3_body<10, false, 1>(P, C); }
__global__ __launch_bounds__((W3Geom<9, 1>::T), 2) void walker3_kernel_sf9_grad_half(DevParams P, LaunchCfg C) { walker3_body<9, true, 1>(P, C); }
__global__ __launch_bounds__((W3Geom<10, 1>::T), 2) void walker3_kernel_sf10_grad_half(DevParams P, LaunchCfg C) { walker3_body<10, true, 1>(P, C); }
// header-only variants (LaunchCfg.skip_payload), full-size workgroups: the launches of a decoupled pass hold fewer jobs than the device has CUs
__global__ __launch_bounds__(W3Geom<9>::T, W3Geom<9>::T512 ? 2 : 4) void walker3_kernel_sf9_skip(DevParams P, LaunchCfg C) { walker3_body<9, false, 0, true>(P, C); }
__global__ __launch_bounds__(W3Geom<10>::T, W3Geom<10>::T512 ? 2 : 4) void walker3_kernel_sf10_skip(DevParams P, LaunchCfg C) { walker3_body<10, false, 0, true>(P, C); }
__global__ __launch_bounds__(W3Geom<11>::T, W3Geom<11>::T512 ? 2 : 4) void walker3_kernel_sf11_skip(DevParams P, LaunchCfg C) { walker3_body<11, false, 0, true>(P, C); }
__global__ __launch_bounds__(W3Geom<12>::T, W3Geom<12>::T512 ? 2 : 4) void walker3_kernel_sf12_skip(DevParams P, LaunchCfg C) { walker3_body<12, false, 0, true>(P, C); }
__global__ __launch_bounds__(W3Geom<9>::T, W3Geom<9>::T512 ? 2 : 4) void walker3_kernel_sf9_grad_skip(DevParams P, LaunchCfg C) { walker3_body<9, true, 0, true>(P, C); }
__global__ __launch_bounds__(W3Geom<10>::T, W3Geom<10>::T512 ? 2 : 4) void walker3_kernel_sf10_grad_skip(DevParams P, LaunchCfg C) { walker3_body<10, true, 0, true>(P, C); }
__global__ __launch_bounds__(W3Geom<11>::T, W3Geom<11>::T512 ? 2 : 4) void walker3_kernel_sf11_grad_skip(DevParams P, LaunchCfg C) { walker3_body<11, true, 0, true>(P, C); }
__global__ __launch_bounds__(W3Geom<12>::T, W3Geom<12>::T512 ? 2 : 4) void walker3_kernel_sf12_grad_skip(DevParams P, LaunchCfg C) { walker3_body<12, true, 0, true>(P, C); }
// demod_mode 0: the gradient demodulator (the reference's shipped default, :499) in the decode rounds
__global__ __launch_bounds__(W3Geom<9>::T, W3Geom<9>::T512 ? 2 : 4) void walker3_kernel_sf9_grad(DevParams P, LaunchCfg C) { walker3_body<9, true>(P, C); }
__global__ __launch_bounds__(W3Geom<10>::T, W3Geom<10>::T512 ? 2 : 4) void walker3_kernel_sf10_grad(DevParams P, LaunchCfg C) { walker3_body<10, true>(P, C); }
__global__ __launch_bounds__(W3Geom<11>::T, W3Geom<11>::T512 ? 2 : 4) void walker3_kernel_sf11_grad(DevParams P, LaunchCfg C) { walker3_body<11, true>(P, C); }
__global__ __launch_bounds__(W3Geom<12>::T, W3Geom<12>::T512 ? 2 : 4) void walker3_kernel_sf12_grad(DevParams P, LaunchCfg C) { walker3_body<12, true>(P, C); }

// ---- symbol-level kernel: one group per symbol, for lora_hip_demod_symbols_device and the payload pass -----------------
template <int SF, bool GRAD, int HV = 0>
__global__ __launch_bounds__((W3Geom<SF, HV>::T), (W3Geom<SF, HV>::T512 ? 2 : 4)) void demod_symbols_w3_kernel(DevParams P, const float2 *iq, const int64_t *offsets, uint32_t n,
                                                                    uint32_t *bins, int32_t *fine, long long *stamps_out, DemodAlt alt)
{
    using G = W3Geom<SF, HV>;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const W3Lds<SF, HV> L = w3_carve<SF, HV>(smem);
    w3_tables_to_lds<SF, HV>(P, L);
    __syncthreads();
    int slot = 0;
    const W3DemodArgs DA{P.down, P.w3_ctab, P.up_ifreq_v, P.enable_fine_sync, P.demod_mode, P.ffs_on, P.ffs_alpha, P.ffs_jump, P.ffs_tol};
    for (uint32_t s0 = blockIdx.x * G::NG; s0 < n; s0 += gridDim.x * G::NG) {
        uint32_t b[G::NG];
        int32_t fs[G::NG];
        float en[G::NG];
        long long stamps[9];
        static_assert(!GRAD, "the gradient demodulator has a kernel of its own: demod_symbols_w3_grad_kernel");
        const float2 *xa[G::NG]; // every group's window (uniform)
        uint32_t vmask = 0u;
#pragma unroll
        for (int g = 0; g < G::NG; g++) {
            const bool gv = s0 + (uint32_t)g < n;
            vmask |= gv ? (1u << g) : 0u;
            xa[g] = iq + offsets[gv ? s0 + (uint32_t)g : s0];
        }
        w3_demod_round<SF, HV>(DA, L, xa, vmask, false, slot, b, fs, en, stamps_out ? stamps : nullptr);
        if (stamps_out && blockIdx.x == 0 && (threadIdx.x & 63) == 0 && s0 == blockIdx.x * G::NG + gridDim.x * G::NG) // (second round of block 0: every wavefront's stamps)
            for (int i = 0; i < 9; i++) stamps_out[(threadIdx.x >> 6) * 9 + i] = stamps[i];
        int32_t *fs_all = L.ws->sh.ibuf; // every group's d_fine_sync (the demodulator hands all groups' results to thread 0's wavefront only)
        if (threadIdx.x == 0) {
            for (int g = 0; g < G::NG; g++) {
                if (s0 + (uint32_t)g < n) { bins[s0 + g] = b[g]; if (fine) fine[s0 + g] = fs[g]; }
                fs_all[g] = fs[g];
            }
        }
        __syncthreads();
        { // a window with a sample of exactly zero comes back as kFinePoison: the round once more, by the ZM instantiations (every ifreq value as the reference forms it)
            bool pz = false;
#pragma unroll
            for (int g = 0; g < G::NG; g++) pz = pz || __builtin_amdgcn_readfirstlane(fs_all[g]) == kFinePoison;
            if (pz) { // (uniform over the workgroup)
                w3_demod_round<SF, HV, true>(DA, L, xa, vmask, false, slot, b, fs, en, nullptr);
                if (threadIdx.x == 0) {
                    for (int g = 0; g < G::NG; g++) {
                        if (s0 + (uint32_t)g < n) { bins[s0 + g] = b[g]; if (fine) fine[s0 + g] = fs[g]; }
                        fs_all[g] = fs[g];
                    }
                }
                __syncthreads();
            }
        }
        if (alt.shift) { // second reads (DemodAlt): the successors of the symbols that moved the symbol clock, that far further on
#pragma unroll
            for (int g = 0; g < G::NG; g++) fs[g] = __builtin_amdgcn_readfirstlane(fs_all[g]);
            const float2 *xa2[G::NG]; // the groups' second windows
            uint32_t vmask2 = 0u;
#pragma unroll
            for (int g = 0; g < G::NG; g++) {
                xa2[g] = iq + offsets[s0];
                if (s0 + (uint32_t)g + 1u >= n || fs[g] == 0) continue; // (fs: uniform over the workgroup)
                const int64_t o0 = offsets[s0 + g], o1 = offsets[s0 + g + 1];
                const int64_t a = o1 + (int64_t)fs[g];
                if (o1 != o0 + (int64_t)G::SPS || a < 0 || a > alt.max_start) continue;
                vmask2 |= 1u << g;
                xa2[g] = iq + a;
            }
            if (vmask2 != 0u) {
                uint32_t b2[G::NG];
                int32_t f2[G::NG];
                w3_demod_round<SF, HV>(DA, L, xa2, vmask2, false, slot, b2, f2, en, nullptr);
                if (threadIdx.x == 0) {
                    for (int g = 0; g < G::NG; g++) {
                        if (!((vmask2 >> g) & 1u)) continue;
                        if (f2[g] == kFinePoison) { vmask2 &= ~(1u << g); continue; } // (a zero sample in the second window: no second read on offer - the walk asks for this shift as a read of its own)
                        alt.bins[s0 + g + 1] = b2[g]; alt.fine[s0 + g + 1] = f2[g];
                    }
                }
                __syncthreads();
            }
            // DemodAlt.shift[s + 1] is written by the round that demodulated symbol s, taken or not (and shift[0] by the first round): the caller clears nothing
            if (threadIdx.x == 0) {
                for (int g = 0; g < G::NG; g++)
                    if (s0 + (uint32_t)g + 1u < n) alt.shift[s0 + g + 1] = ((vmask2 >> g) & 1u) ? fs[g] : 0;
                if (s0 == 0u) alt.shift[0] = 0;
            }
        }
    }
}

// ---- host side: tables -----------------------------------------------------------------------------------------
// w3_tw: W_N^t, t < NTW.  w3_ctab: the combine coefficient of value i of thread t3 in round g at [(g 16 + i) T + t3]:
// W_sps^{k r} for the signed bin k of k1 = a + 16 a2 + 256 b2 (+ the fold at k1 = N/2, :450).
// The gradient demodulator on caller-given windows (lora_hip_demod_symbols_device, the payload pass of a decoupled pass): one WAVEFRONT per symbol
// (w3_wave_window_grad), no LDS, no barrier; a poisoned window (a sample of exactly zero) is evaluated again in place by the ZM instantiation.  Second
// reads (DemodAlt): the wavefront whose symbol moved the symbol clock reads that symbol's successor again, that far on - wavefront-local as well.
template <int SF>
__global__ __launch_bounds__(512, 2) void demod_symbols_w3_grad_kernel(DevParams P, const float2 *iq, const int64_t *offsets, uint32_t n, uint32_t *bins, int32_t *fine, DemodAlt alt)
{
    constexpr int SPS = 8 << SF;
    const W3DemodArgs DA{P.down, P.w3_ctab, P.up_ifreq_v, P.enable_fine_sync, P.demod_mode, P.ffs_on, P.ffs_alpha, P.ffs_jump, P.ffs_tol};
    const uint32_t wave = threadIdx.x >> 6, lane = threadIdx.x & 63u;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    float *fcache = reinterpret_cast<float *>(smem) + wave * (kW3GradCacheChunks * 1024);
    for (uint32_t s = blockIdx.x * 8u + wave; s < n; s += gridDim.x * 8u) {
        const int64_t o0 = offsets[s];
        uint32_t b;
        int32_t fs;
        float en;
        w3_wave_window_grad<SF>(DA, iq + o0, false, b, fs, en, fcache);
        if (fs == kFinePoison) w3_wave_window_grad<SF, true>(DA, iq + o0, false, b, fs, en, fcache); // (uniform over the wavefront)
        if (lane == 0u) { bins[s] = b; if (fine) fine[s] = fs; }
        int32_t sh = 0; // the shift this wavefront's second read of its successor was made at (0: none)
        if (alt.shift && fs != 0 && s + 1u < n) {
            const int64_t o1 = offsets[s + 1u], a = o1 + (int64_t)fs;
            if (o1 == o0 + (int64_t)SPS && a >= 0 && a <= alt.max_start) {
                uint32_t b2;
                int32_t f2;
                w3_wave_window_grad<SF>(DA, iq + a, false, b2, f2, en, fcache);
                if (f2 == kFinePoison) w3_wave_window_grad<SF, true>(DA, iq + a, false, b2, f2, en, fcache);
                if (lane == 0u) { alt.bins[s + 1u] = b2; alt.fine[s + 1u] = f2; }
                sh = fs;
            }
        }
        // DemodAlt.shift[s + 1] is this wavefront's to write, taken or not (and shift[0] the first one's): the caller clears nothing
        if (alt.shift && lane == 0u) { if (s + 1u < n) alt.shift[s + 1u] = sh; if (s == 0u) alt.shift[0] = 0; }
    }
}

template <int SF, int HV = 0>
static void build_w3_tables_sf(float2 *tw, float2 *ctab)
{
    using G = W3Geom<SF, HV>;
    constexpr int N = G::N, SPS = G::SPS, T = G::VT;
    constexpr int NTWB = G::WFFT ? N : G::NTW; // (SF9: the full-size workgroup keeps none in LDS, the half-size one all N)
    for (int t = 0; t < NTWB; t++) {
        const double a = -2.0 * M_PI * (double)t / (double)N;
        tw[t] = make_float2((float)std::cos(a), (float)std::sin(a));
    }
    for (int g = 0; g < G::ROUNDS; g++)
        for (int t3 = 0; t3 < T; t3++) {
            const int r = t3 & 7, w = t3 >> 3, row = w % G::AR, wl = w / G::AR;
            const int a_bin = brev_bits(row + G::AR * g, 4);
            for (int j = 0; j < G::NB; j++) {
                const int a2 = brev_bits(wl * G::NB + j, 4);
                for (int m = 0; m < G::M2; m++) {
                    const int k1 = a_bin + 16 * a2 + 256 * brev_bits(m, G::LOGM2);
                    const int k = (k1 < N / 2) ? k1 : k1 - N;
                    const int e = ((k * r) % SPS + SPS) % SPS;
                    const double ang = -2.0 * M_PI * (double)e / (double)SPS;
                    double re = std::cos(ang), im = std::sin(ang);
                    if (k1 == N / 2) { // tmp[N/2] += F[N/2] (:450)
                        const int e2 = ((N / 2) * r) % SPS;
                        const double a2r = -2.0 * M_PI * (double)e2 / (double)SPS;
                        re += std::cos(a2r); im += std::sin(a2r);
                    }
                    ctab[(size_t)(g * 16 + j * G::M2 + m) * T + t3] = make_float2((float)re, (float)im);
                }
            }
        }
}

bool walker3_covers(uint32_t sf) { return sf >= 9u && sf <= 12u; }
uint32_t w3_tw_entries(uint32_t sf) { return sf == 9u ? W3Geom<9, 1>::NTW : sf == 10u ? W3Geom<10>::NTW : sf == 11u ? W3Geom<11>::NTW : sf == 12u ? W3Geom<12>::NTW : 0u; }
void build_w3_tables(uint32_t sf, float2 *tw, float2 *ctab /* sps entries */)
{
    if (sf == 9u) build_w3_tables_sf<9>(tw, ctab);
    else if (sf == 10u) build_w3_tables_sf<10>(tw, ctab);
    else if (sf == 11u) build_w3_tables_sf<11>(tw, ctab);
    else if (sf == 12u) build_w3_tables_sf<12>(tw, ctab);
}

static uint32_t walker3_threads_half(uint32_t sf) { return (uint32_t)(sf == 9u ? W3Geom<9, 1>::T : W3Geom<10, 1>::T); }
static uint32_t walker3_lds_half(uint32_t sf) { return sf == 9u ? w3_lds_bytes<9, 1>() : w3_lds_bytes<10, 1>(); }
static uint32_t walker3_threads(uint32_t sf) { return (uint32_t)(sf == 9u ? W3Geom<9>::T : sf == 10u ? W3Geom<10>::T : sf == 11u ? W3Geom<11>::T : W3Geom<12>::T); }
static uint32_t walker3_groups(uint32_t sf) { return sf == 9u ? W3Geom<9>::NG : sf == 10u ? W3Geom<10>::NG : 1u; }
static uint32_t walker3_lds(uint32_t sf) { return sf == 9u ? w3_lds_bytes<9>() : sf == 10u ? w3_lds_bytes<10>() : sf == 11u ? w3_lds_bytes<11>() : w3_lds_bytes<12>(); }
