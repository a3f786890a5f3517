// lora_walker3.inc.hip -- the SF9 .. SF12 walker (decimation 8, FFT demodulators): one workgroup demodulates one
// symbol window at a time, cooperatively.  Included by lora_kernels.hip.
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
// the samples while they are in registers for the dechirp (x[n-1] is the neighbouring lane's sample: one DPP wave
// rotate; lane 0 takes it from one extra load per wavefront that fetches all 16 chunk-boundary samples at once).
//
// The state machine is the reference's work() (:740-903), replicated uniformly in every thread like the generic
// walker's; the symbol clock's serial dependency (d_fine_sync, :321,:856,:883) therefore needs no speculation.
// DETECT, SYNC (closed form over block-wide prefix sums) and FIND_SFD (one-pass Pearson, closed-form 63-lag
// fine_sync) are workgroup-cooperative too.  Latency is covered by the other workgroups of the CU (SF9: 4 x 256
// threads, SF10: 2 x 512, SF11 / SF12: 1 x 1024 - 16 wavefronts per CU in every case).

template <int SF> struct W3Geom {
    static constexpr int N = 1 << SF, SPS = 8 * N;
    static constexpr int T = SF >= 11 ? 1024 : N / 2;   // threads per workgroup
    static constexpr int WAVES = T / 64;
    static constexpr int PAIRS = SPS / (16 * T);        // (q0, r) pairs per thread: 1; SF12: 2
    static constexpr int ROUNDS = PAIRS;                // passes over the rows of pass 1
    static constexpr int AR = 16 / ROUNDS;              // rows resident in LDS per round
    static constexpr int M = N / 16, M2 = M / 16, LOGM2 = ilog2(M2);
    static constexpr int SA = 8 * M + 8;                // entries per row
    static constexpr int NTW = N > 2048 ? N / 2 : N;    // W_N^t entries kept in LDS (SF12: half, the rest by sign)
    static constexpr int CH = SPS / 16;                 // samples between a thread's consecutive loads
    static constexpr int NWL = T / (8 * AR), NB = 16 / NWL; // pass 3: butterflies per thread
    static constexpr int LEN = SPS / T;                 // SYNC: shifts per thread
    static constexpr uint32_t data_entries = (uint32_t)AR * SA;
    static_assert(NB * M2 == 16, "pass 3 covers 16 values per thread");
    static_assert(AR * M2 * 8 == T, "pass 2 uses every thread once per round");
};

struct alignas(16) W3Shared {
    float    red[2][16 * 8 + 8];   // block reductions, double-buffered (one barrier each)
    double   dred[2][16 * 4 + 4];  // block scans / sums of doubles (SYNC, FIND_SFD)
    float    sfd[72];              // FIND_SFD: head / tail samples of the window's ifreq
    int32_t  ibc[8];               // broadcasts
    Shared   sh;                   // integer chain (words / codewords / decoded bytes)
};

template <int SF> struct W3Lds {
    v2f      *data;  // [AR][SA]
    v2f      *tw;    // [NTW]  W_N^t
    W3Shared *ws;
};

template <int SF>
__device__ __forceinline__ W3Lds<SF> w3_carve(unsigned char *smem)
{
    using G = W3Geom<SF>;
    W3Lds<SF> L;
    L.data = reinterpret_cast<v2f *>(smem);
    L.tw = L.data + G::data_entries;
    L.ws = reinterpret_cast<W3Shared *>(L.tw + G::NTW);
    return L;
}
template <int SF> constexpr uint32_t w3_lds_bytes()
{
    using G = W3Geom<SF>;
    return (uint32_t)((G::data_entries + G::NTW) * sizeof(v2f) + ((sizeof(W3Shared) + 15) & ~(size_t)15));
}

template <int SF>
__device__ __forceinline__ v2f w3_tw(const W3Lds<SF> &L, uint32_t idx)
{
    using G = W3Geom<SF>;
    if constexpr (G::NTW == G::N) return L.tw[idx];
    else {
        const v2f w = L.tw[idx & (uint32_t)(G::NTW - 1)];
        return (idx & (uint32_t)G::NTW) ? -w : w;
    }
}

__device__ __forceinline__ v2f cmul2(v2f a, v2f w) { return __builtin_elementwise_fma(a.xx, w, a.yy * (v2f){-w.y, w.x}); }

// ---- block reductions (T threads, slot alternates between consecutive calls) ---------------------------------
template <int WAVES, int K>
__device__ __forceinline__ void w3_block_sum(float (&v)[K], W3Shared &ws, int &slot)
{
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    float *red = ws.red[slot];
    slot ^= 1;
#pragma unroll
    for (int k = 0; k < K; k++) v[k] = wave_sum_rows(v[k]);
    if (lane == 0) {
#pragma unroll
        for (int k = 0; k < K; k++) red[wave * K + k] = v[k];
    }
    __syncthreads();
#pragma unroll
    for (int k = 0; k < K; k++) {
        float s = 0.0f;
#pragma unroll
        for (int w = 0; w < WAVES; w++) s += red[w * K + k];
        v[k] = __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, s)));
    }
}

template <int WAVES>
__device__ __forceinline__ void w3_block_argmax_first(float &v, int &idx, W3Shared &ws, int &slot)
{
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    float *red = ws.red[slot];
    slot ^= 1;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        const float ov = __shfl_xor(v, o, 64);
        const int oi = __shfl_xor(idx, o, 64);
        if (ov > v || (ov == v && oi < idx)) { v = ov; idx = oi; }
    }
    if (lane == 0) { red[wave] = v; ((int *)red)[64 + wave] = idx; }
    __syncthreads();
    float bv = red[0];
    int bi = ((int *)red)[64];
#pragma unroll
    for (int w = 1; w < WAVES; w++) {
        const float ov = red[w];
        const int oi = ((int *)red)[64 + w];
        if (ov > bv || (ov == bv && oi < bi)) { bv = ov; bi = oi; }
    }
    v = __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, bv)));
    idx = __builtin_amdgcn_readfirstlane(bi);
}

__device__ __forceinline__ double w3_wave_sum_d(double v)
{
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}

// ---- the cooperative symbol demodulator ----------------------------------------------------------------------
// s_out = get_shift_fft's value for the window x[0 .. sps) (:430-464); fine_out = d_fine_sync after fine_sync(bin_idx, 2)
// (:300-338, :501-502; 0 when drift correction is off); energy_out = determine_energy (:368-375) when want_energy.
// Called by all T threads of the workgroup; the LDS twiddle table must be in place.
template <int SF>
__device__ __forceinline__ void w3_demod_symbol(const DevParams &P, const W3Lds<SF> &L, const float2 *__restrict__ x, bool want_energy, int &slot,
                                                uint32_t &s_out, int32_t &fine_out, float &energy_out)
{
    using G = W3Geom<SF>;
    constexpr int N = G::N, SPS = G::SPS, T = G::T, CH = G::CH, M2 = G::M2, AR = G::AR, SA = G::SA, PAIRS = G::PAIRS, ROUNDS = G::ROUNDS;
    int t = threadIdx.x;
    asm volatile("" : "+v"(t)); // keeps per-thread table addresses out of the caller's loop-invariant set
    const int lane = t & 63, wave = t >> 6, r = t & 7;
    const bool want_fine = P.enable_fine_sync != 0u;
    const auto xv = (const __attribute__((address_space(1))) v2f *)x;
    const auto dv = (const __attribute__((address_space(1))) v2f *)P.down;
    const auto cv = (const __attribute__((address_space(1))) v2f *)P.w3_ctab;
    W3Shared &ws = *L.ws;

    float f[PAIRS][16];     // ifreq[n - 1] of this thread's samples
    v2f hold[ROUNDS > 1 ? PAIRS : 1][8]; // SF12: rows 8..15 of pass 1 wait here for round 1
    float en = 0.0f;

    // ---- pass 1 ----
#pragma unroll
    for (int p = 0; p < PAIRS; p++) {
        const int base = p * T + t;
        v2f a[16], d[16];
#pragma unroll
        for (int c = 0; c < 16; c++) a[c] = xv[c * CH + base];
        v2f bnd = (v2f){0.0f, 0.0f}; // lane c: the sample before this wavefront's first one in chunk c
        if (want_fine) {
            const int bi = (lane & 15) * CH + p * T + 64 * wave - 1;
            bnd = xv[bi < 0 ? 0 : bi];
        }
#pragma unroll
        for (int c = 0; c < 16; c++) d[c] = dv[c * CH + base];
        if (want_fine) {
            const float bnd_xf = bnd.x, bnd_yf = bnd.y; // scalars first: bit_cast on a vector element reads element 0 with this compiler
            const int bnd_xi = __builtin_bit_cast(int, bnd_xf), bnd_yi = __builtin_bit_cast(int, bnd_yf);
#pragma unroll
            for (int c = 0; c < 16; c += 2) {
                const float bx0 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(bnd_xi, c));
                const float by0 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(bnd_yi, c));
                const float bx1 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(bnd_xi, c + 1));
                const float by1 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(bnd_yi, c + 1));
                const v2f r0 = dpp2<kDppWaveRor1>(a[c]), r1 = dpp2<kDppWaveRor1>(a[c + 1]);
                const v2f p0 = (lane == 0) ? (v2f){bx0, by0} : r0, p1 = (lane == 0) ? (v2f){bx1, by1} : r1;
                const v2f fp = ifreq_prod_pk(p0, a[c], p1, a[c + 1]);
                f[p][c] = (c == 0 && p == 0 && t == 0) ? 0.0f : fp.x; // n = 0 has no predecessor in the window
                f[p][c + 1] = fp.y;
            }
        }
        if (want_energy) {
#pragma unroll
            for (int c = 0; c < 16; c++) en += a[c].x * a[c].x + a[c].y * a[c].y;
        }
#pragma unroll
        for (int c = 0; c < 16; c++) a[c] = cmul2(a[c], d[c]); // dechirp (:437)
        fft_inlane_dif_pk<16>(a);
        const int q0 = base >> 3;
#pragma unroll
        for (int m = 1; m < 16; m++) a[m] = cmul2(a[m], w3_tw<SF>(L, (uint32_t)(q0 * brev_bits(m, 4))));
#pragma unroll
        for (int m = 0; m < AR; m++) L.data[m * SA + q0 * 8 + r] = a[m];
        if constexpr (ROUNDS > 1) {
#pragma unroll
            for (int i = 0; i < 8; i++) hold[p][i] = a[8 + i];
        }
    }

    float bv = -1.0f;
    int bi = 0x7fffffff;
#pragma unroll
    for (int g = 0; g < ROUNDS; g++) {
        if (g > 0) {
            if constexpr (ROUNDS > 1) {
                __syncthreads(); // round 0's pass 3 has read the rows
#pragma unroll
                for (int p = 0; p < PAIRS; p++) {
                    const int q0 = (p * T + t) >> 3;
#pragma unroll
                    for (int i = 0; i < 8; i++) L.data[i * SA + q0 * 8 + r] = hold[p][i];
                }
            }
        }
        __syncthreads();
        // ---- pass 2 (in place) ----
        v2f cc[16]; // combine coefficients of pass 3: loaded here, used there
#pragma unroll
        for (int i = 0; i < 16; i++) cc[i] = cv[(g * 16 + i) * T + t];
        {
            const int row = (t >> 3) % AR, q1 = t / (8 * AR);
            v2f *pe = L.data + row * SA + q1 * 8 + r;
            v2f a[16];
#pragma unroll
            for (int c2 = 0; c2 < 16; c2++) a[c2] = pe[M2 * 8 * c2];
            fft_inlane_dif_pk<16>(a);
            if (q1 != 0) { // W_{N/16}^{q1 a2} = W_N^{16 q1 a2}; q1 is the same in all lanes of a wavefront
#pragma unroll
                for (int m = 1; m < 16; m++) a[m] = cmul2(a[m], w3_tw<SF>(L, (uint32_t)(16 * q1 * brev_bits(m, 4))));
            }
#pragma unroll
            for (int m = 0; m < 16; m++) pe[M2 * 8 * m] = a[m];
        }
        __syncthreads();
        // ---- pass 3 + combine ----
        {
            const int w = t >> 3, row = w % AR, wl = w / AR;
            v2f out[16];
#pragma unroll
            for (int j = 0; j < G::NB; j++) {
                const int m2 = wl * G::NB + j;
                v2f b[M2];
#pragma unroll
                for (int q1 = 0; q1 < M2; q1++) b[q1] = L.data[row * SA + (q1 + M2 * m2) * 8 + r];
                fft_inlane_dif_pk<M2>(b);
#pragma unroll
                for (int m = 0; m < M2; m++) out[j * M2 + m] = cmul2(b[m], cc[j * M2 + m]);
            }
#pragma unroll
            for (int i = 0; i < 16; i++) { // sum over r = lane bits 0, 1, 2
                v2f s = out[i];
                s += dpp2<kDppQuadXor1>(s);
                s += dpp2<kDppQuadXor2>(s);
                const float ox = s.x, oy = s.y;
                out[i] = (v2f){ox + lane_xor<4>(ox), oy + lane_xor<4>(oy)};
            }
            const int a_bin = (int)(__brev((uint32_t)(row + AR * g)) >> 28);
#pragma unroll
            for (int j = 0; j < G::NB; j++) {
                const int a2 = (int)(__brev((uint32_t)(wl * G::NB + j)) >> 28);
#pragma unroll
                for (int m = 0; m < M2; m++) {
                    const int k1 = a_bin + 16 * a2 + 256 * brev_bits(m, G::LOGM2);
                    const v2f o = out[j * M2 + m];
                    const float mag = o.x * o.x + o.y * o.y; // monotone in std::abs (:454)
                    if (mag > bv || (mag == bv && k1 < bi)) { bv = mag; bi = k1; }
                }
            }
        }
    }
    w3_block_argmax_first<G::WAVES>(bv, bi, ws, slot);
    const uint32_t s = (uint32_t)bi;
    s_out = s;
    fine_out = 0;
    energy_out = 0.0f;
    if (want_energy) {
        float e1[1] = {en};
        w3_block_sum<G::WAVES, 1>(e1, ws, slot);
        energy_out = e1[0];
    }
    if (!want_fine) return;
    // fine_sync (:300-338) with search = max(D/4, 2) = 2 -> lags -1, 0, +1
    const uint32_t bin_idx = (s == 0u && P.demod_mode == 2u) ? 0u : (s + (uint32_t)N - 1u) % (uint32_t)N;
    const auto vv = (const __attribute__((address_space(1))) float *)P.up_ifreq_v + ((int)(bin_idx + 1u) * 8 + SPS);
    float cs[3] = {0.f, 0.f, 0.f};
#pragma unroll
    for (int p = 0; p < PAIRS; p++) {
        float v0[16], v1[16], v2[16];
#pragma unroll
        for (int c = 0; c < 16; c++) {
            const int n = c * CH + p * T + t;
            const int k = (n >= 1) ? n - 1 : 1; // f is 0 for the non-existent k = -1
            v0[c] = vv[k - 1]; v1[c] = vv[k]; v2[c] = vv[k + 1];
        }
#pragma unroll
        for (int c = 0; c < 16; c++) {
            const float fk = f[p][c];
            cs[0] += fk * v0[c]; cs[1] += fk * v1[c]; cs[2] += fk * v2[c];
        }
        if (p == PAIRS - 1 && t == T - 1) { // ifreq[sps-1] = ifreq[sps-2] (:243): the duplicated tap at k = sps-1
            const float fl = f[p][15];
            const int k = SPS - 1;
            cs[0] += fl * vv[k - 1]; cs[1] += fl * vv[k]; cs[2] += fl * vv[k + 1];
        }
    }
    w3_block_sum<G::WAVES, 3>(cs, ws, slot);
    float mx = 0.0f;
    int32_t lag = 0;
    if (cs[0] > mx) { mx = cs[0]; lag = -1; }
    if (cs[1] > mx) { mx = cs[1]; lag = 0; }
    if (cs[2] > mx) { mx = cs[2]; lag = 1; }
    fine_out = -lag;
}

// copies W_N^t into LDS; all threads; the caller synchronises
template <int SF>
__device__ __forceinline__ void w3_tables_to_lds(const DevParams &P, const W3Lds<SF> &L)
{
    using G = W3Geom<SF>;
    const v2f *__restrict__ src = reinterpret_cast<const v2f *>(P.w3_tw);
    for (int i = threadIdx.x; i < G::NTW; i += G::T) L.tw[i] = src[i];
}

// ---- DETECT (:340-366): sums of c1 conj(c2), |c1|^2, |c2|^2 over the symbol pair at x ------------------------
template <int SF>
__device__ __forceinline__ void w3_detect_window(const float2 *__restrict__ x, W3Shared &ws, int &slot, float (&a)[4])
{
    using G = W3Geom<SF>;
    const auto xv = (const __attribute__((address_space(1))) v2f *)x;
    const int t = threadIdx.x;
    a[0] = a[1] = a[2] = a[3] = 0.0f;
#pragma unroll
    for (int p = 0; p < G::PAIRS; p++) {
        v2f u[16], w[16];
#pragma unroll
        for (int c = 0; c < 16; c++) u[c] = xv[c * G::CH + p * G::T + t];
#pragma unroll
        for (int c = 0; c < 16; c++) w[c] = xv[G::SPS + c * G::CH + p * G::T + t];
#pragma unroll
        for (int c = 0; c < 16; c++) {
            const v2f c1 = u[c], c2 = w[c];
            a[0] += c1.x * c2.x + c1.y * c2.y;
            a[1] += c1.y * c2.x - c1.x * c2.y;
            a[2] += c1.x * c1.x + c1.y * c1.y;
            a[3] += c2.x * c2.x + c2.y * c2.y;
        }
    }
    w3_block_sum<G::WAVES, 4>(a, ws, slot);
}

// ---- SYNC (:770-783, detect_upchirp :392-413) -----------------------------------------------------------------
// C[i] = sum_{k<n} f[i+k] u[k], n = sps-1, i < sps, f = ifreq of x[0 .. 2 sps).  d_upchirp_ifreq is the line a + b k (up
// to float noise ~1e-5 of the peak), so C[i] = a S0[i] + b S1[i] with S0 = sum_k f[i+k], S1 = sum_k k f[i+k], both from
// prefix sums of f and (t - sps) f in double.  Thread tau owns the LEN shifts i0 = LEN tau ..: it computes f on
// A = [i0, i0 + LEN) and on B = [i0 + n, i0 + n + LEN) straight from the samples, the workgroup scans the A and B sums,
// and the thread slides over its shifts.  Returns the best correlation and its (first) shift.
struct W3SyncOut { float bv; int bi; int slot; };
template <int SF>
__device__ __attribute__((noinline)) W3SyncOut w3_sync(double sync_a, double sync_b, const float2 *__restrict__ x, W3Shared *wsp, int slot)
{
    W3Shared &ws = *wsp;
    using G = W3Geom<SF>;
    constexpr int SPS = G::SPS, LEN = G::LEN, WAVES = G::WAVES;
    constexpr int n = SPS - 1;
    const auto xv = (const __attribute__((address_space(1))) v2f *)x;
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const int i0 = t * LEN;
    double s0A = 0.0, gA = 0.0, s0B = 0.0, gB = 0.0; // sums of f and of (pos - sps) f over A and B
    float fa[LEN], fb[LEN];
    {
        v2f xa[LEN + 1];
#pragma unroll
        for (int j = 0; j <= LEN; j++) xa[j] = xv[i0 + j];
#pragma unroll
        for (int j = 0; j < LEN; j += 2) {
            const v2f fp = ifreq_prod_pk(xa[j], xa[j + 1], xa[j + 1], xa[j + 2 <= LEN ? j + 2 : LEN]);
            fa[j] = fp.x; fa[j + 1] = fp.y;
        }
    }
    {
        v2f xb[LEN + 1];
#pragma unroll
        for (int j = 0; j <= LEN; j++) xb[j] = xv[i0 + n + j];
#pragma unroll
        for (int j = 0; j < LEN; j += 2) {
            const v2f fp = ifreq_prod_pk(xb[j], xb[j + 1], xb[j + 1], xb[j + 2 <= LEN ? j + 2 : LEN]);
            fb[j] = fp.x; fb[j + 1] = fp.y;
        }
    }
#pragma unroll
    for (int j = 0; j < LEN; j++) {
        s0A += (double)fa[j]; gA += (double)(i0 + j - SPS) * (double)fa[j];
        s0B += (double)fb[j]; gB += (double)(i0 + n + j - SPS) * (double)fb[j];
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
    if (t == G::T - 1) dr[64] = (double)fa[LEN - 1]; // f[sps-1]: the last element of the last A segment
    __syncthreads();
#pragma unroll
    for (int q = 0; q < 4; q++) {
        double pre = 0.0, all = 0.0;
        for (int w = 0; w < WAVES; w++) { const double v = dr[w * 4 + q]; if (w < wave) pre += v; all += v; }
        ex[q] += pre; tot[q] = all;
    }
    const double flast = dr[64];
    // prefix sums over t < i0 and t < i0 + n:  F(i0) = exA,  F(i0 + n) = F(sps - 1) + exB,  F(sps - 1) = totA - f[sps-1]
    const double F0 = ex[0], G0 = ex[1];
    const double F1 = (tot[0] - flast) + ex[2], G1 = (tot[1] - (double)(SPS - 1 - SPS) * flast) + ex[3];
    double s0 = F1 - F0, s1 = (G1 - G0) + (double)(SPS - i0) * s0;
    float bv = 0.0f; // max_correlation = 0 (:400)
    int bi = 0x7fffffff;
#pragma unroll
    for (int rr = 0; rr < LEN; rr++) {
        const float c = (float)(sync_a * s0 + sync_b * s1);
        if (c > bv) { bv = c; bi = i0 + rr; }
        const double fin = (double)fb[rr], fout = (double)fa[rr];
        s0 += fin - fout;
        s1 += (double)n * fin - s0;
    }
    __syncthreads(); // dred is free again
    w3_block_argmax_first<WAVES>(bv, bi, ws, slot);
    return W3SyncOut{bv, bi, slot};
}

// ---- FIND_SFD (:385-390, :283-298, :801-803) ------------------------------------------------------------------
// Pearson correlation of the window's ifreq with the ideal downchirp ifreq (one pass); for an upchirp (c < -0.97)
// fine_sync(-1, 4 D) over the 63 lags in closed form (see w2_sfd_window for the derivation).
struct W3SfdOut { float c; int32_t fine; int slot; };
struct W3SfdArgs { const float *down_ifreq, *up_ifreq_v; float down_ifreq_avg, down_ifreq_sd, down_ifreq_dsum; double sync_a, sync_b; };
template <int SF>
__device__ __attribute__((noinline)) W3SfdOut w3_sfd_window(W3SfdArgs P, const float2 *__restrict__ x, W3Shared *wsp, int slot)
{
    W3Shared &ws = *wsp;
    using G = W3Geom<SF>;
    constexpr int SPS = G::SPS, T = G::T, CH = G::CH, PAIRS = G::PAIRS, WAVES = G::WAVES;
    const auto xv = (const __attribute__((address_space(1))) v2f *)x;
    const auto ddv = (const __attribute__((address_space(1))) float *)P.down_ifreq;
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    float a3[3] = {0.f, 0.f, 0.f};
    double g0 = 0.0, g1 = 0.0;
    float f_first = 0.0f, f_last = 0.0f; // chunk 0 of pair 0, chunk 15 of the last pair
#pragma unroll
    for (int p = 0; p < PAIRS; p++) {
        const int base = p * T + t;
        v2f a[16];
        float dd[16], f[16];
#pragma unroll
        for (int c = 0; c < 16; c++) a[c] = xv[c * CH + base];
        const int bi = (lane & 15) * CH + p * T + 64 * wave - 1;
        const v2f bnd = xv[bi < 0 ? 0 : bi];
#pragma unroll
        for (int c = 0; c < 16; c++) { const int k = c * CH + base - 1; dd[c] = ddv[k < 0 ? 0 : k]; }
        const float bnd_xf = bnd.x, bnd_yf = bnd.y; // (scalars first, as in w3_demod_symbol)
        const int bnd_xi = __builtin_bit_cast(int, bnd_xf), bnd_yi = __builtin_bit_cast(int, bnd_yf);
#pragma unroll
        for (int c = 0; c < 16; c += 2) {
            const float bx0 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(bnd_xi, c));
            const float by0 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(bnd_yi, c));
            const float bx1 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(bnd_xi, c + 1));
            const float by1 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(bnd_yi, c + 1));
            const v2f r0 = dpp2<kDppWaveRor1>(a[c]), r1 = dpp2<kDppWaveRor1>(a[c + 1]);
            const v2f p0 = (lane == 0) ? (v2f){bx0, by0} : r0, p1 = (lane == 0) ? (v2f){bx1, by1} : r1;
            const v2f fp = ifreq_prod_pk(p0, a[c], p1, a[c + 1]);
            f[c] = (c == 0 && p == 0 && t == 0) ? 0.0f : fp.x;
            f[c + 1] = fp.y;
        }
#pragma unroll
        for (int c = 0; c < 16; c++) {
            const float fk = f[c];
            const float d = dd[c] - P.down_ifreq_avg;
            a3[0] += fk; a3[1] += fk * fk; a3[2] += fk * d; // f is 0 for the non-existent k = -1
            const int k = c * CH + base - 1;
            g0 += (double)fk; g1 += (double)k * (double)fk;
        }
        if (p == 0) f_first = f[0];
        if (p == PAIRS - 1) f_last = f[15];
    }
    w3_block_sum<WAVES, 3>(a3, ws, slot);
    const float nf = (float)(SPS - 1);
    const float average = a3[0] / nf;
    const float var = fmaxf(a3[1] / nf - average * average, 0.0f);
    const float sd = sqrtf(var) * P.down_ifreq_sd;
    const float c = (a3[2] - average * P.down_ifreq_dsum) / sd / nf;
    if (!(c < -0.97f) || c > 0.96f) return W3SfdOut{c, 0, slot}; // (uniform)
    // fine_sync(-1, 32): c_i = sum_{k<sps} fe[k] v[sps + i + k], i = -31 .. 31, fe[sps-1] = fe[sps-2]
    if (t == T - 1) { g0 += (double)f_last; g1 += (double)(SPS - 1) * (double)f_last; } // duplicated last tap (:243)
    g0 = w3_wave_sum_d(g0); g1 = w3_wave_sum_d(g1);
    double *dr = ws.dred[slot & 1];
    if (lane == 0) { dr[wave * 2] = g0; dr[wave * 2 + 1] = g1; }
    // head[k] = fe[k], k < 32 (threads 1 .. 32 of chunk 0); tail[q] = fe[sps-1-q], q <= 32 (the last threads of chunk 15)
    float *scr = ws.sfd;
    if (t >= 1 && t <= 32) scr[t - 1] = f_first;
    if (t >= T - 32) scr[32 + 1 + (T - 1 - t)] = f_last;
    if (t == T - 1) scr[32] = f_last;
    __syncthreads();
    double G0 = 0.0, G1 = 0.0;
    for (int w = 0; w < WAVES; w++) { G0 += dr[w * 2]; G1 += dr[w * 2 + 1]; }
    int32_t lag = 0;
    if (wave == 0) {
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
        lag = (c_i > 0.0f) ? li - 31 : 0;
        if (lane == 0) ws.ibc[0] = lag;
    }
    __syncthreads();
    return W3SfdOut{c, -__builtin_amdgcn_readfirstlane(ws.ibc[0]), slot};
}

// ---- the kernel -----------------------------------------------------------------------------------------------
template <int SF>
__device__ __forceinline__ void walker3_body(const DevParams &P, const LaunchCfg &C)
{
    using G = W3Geom<SF>;
    constexpr uint32_t sps = G::SPS;
    constexpr int T = G::T;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const W3Lds<SF> L = w3_carve<SF>(smem);
    W3Shared &ws = *L.ws;
    Shared &sh = ws.sh;

    const uint32_t jid = blockIdx.x;
    if (jid >= C.n_jobs) return;
    const Job job = C.jobs[jid];
    const float2 *__restrict__ X = C.iq + job.stream_off;
    const int64_t n_items = (int64_t)job.stream_len;
    AttemptRec *recs = C.recs + (size_t)jid * C.recs_per_job;
    StepRec *trace = C.trace ? C.trace + (size_t)jid * C.trace_cap : nullptr;
    const bool t0 = threadIdx.x == 0;
    int slot = 0;

    w3_tables_to_lds<SF>(P, L);
    __syncthreads();

    // uniform (replicated) decoder state -- decoder_impl.h:70-123
    int32_t state = kDetect;
    int64_t pos = job.start;
    uint32_t corr_fails = 0, cr = job.cr_prev, has_crc = P.ctor_crc;
    int32_t payload_symbols = 0;
    uint32_t payload_length = 0, n_words = 0, n_cw = 0, n_sym = 0;
    float energy_threshold = 0.0f;
    uint8_t phdr0 = 0, phdr1 = (uint8_t)((P.ctor_cr << 5) | (P.ctor_crc << 4)), phdr2 = 0;
    uint32_t n_att = 0, npush = 0, n_steps = 0, stop_reason = 0;
    float push_tail[4] = {0.f, 0.f, 0.f, 0.f};
    int64_t att_start = pos, att_trig = -1, att_hdr = -1;
    uint32_t att_cr_prev = cr, att_ambig = 0;
    bool in_attempt = false, frame_ok = false;

    auto end_step = [&](int32_t st_in, int32_t consumed, int32_t step_bin, int32_t fine, float step_val, long long t_start) -> bool {
        if (trace && t0 && n_steps < C.trace_cap) {
            StepRec &s = trace[n_steps];
            s.state = st_in; s.consumed = consumed; s.pos = pos; s.bin = step_bin; s.fine = fine; s.value = step_val;
            s.stream = job.stream_id; s.cycles = (uint32_t)(clock64() - t_start); s.pad = 0;
        }
        n_steps++;
        pos += consumed;
        if (in_attempt && state == kDetect) { // attempt finished: frame published, or sync lost
            if (t0) {
                AttemptRec &r = recs[n_att];
                r.status = frame_ok ? kAttemptFrame : kAttemptLostSync;
                if (!frame_ok) r.frame_len = 0;
                r.start_pos = att_start; r.trig_pos = att_trig; r.hdr_pos = att_hdr; r.end_pos = pos;
                r.npush = npush;
                for (int i = 0; i < 4; i++) r.push_tail[i] = push_tail[i];
                r.cr_prev = att_cr_prev; r.hdr_ambig = att_ambig; r.n_symbols = n_sym;
            }
            n_att++;
            in_attempt = false;
            frame_ok = false;
            npush = 0;
            att_start = pos;
        } else if (in_attempt && job.stop_at_header && state == kDecodeHeader) {
            stop_reason = 3;
            return true;
        }
        return false;
    };

    // everything demodulate() / work() do after the bin is known (:506-529, :826-886)
    auto post_symbol = [&](bool do_demod, uint32_t bin_idx, bool is_first) {
        bool block_done = false;
        if (do_demod) {
            const bool reduced = is_first || P.reduced_rate; // :495
            if (reduced) bin_idx = (uint32_t)(lroundf((float)bin_idx / 4.0f) % (long)P.nbins_hdr); // :507-509
            const uint32_t word = bin_idx ^ (bin_idx >> 1u); // :512
            const uint32_t need = 4u + (is_first ? 4u : cr); // :521
            if (t0 && n_words < 16u) sh.words[n_words] = word;
            n_words++;
            n_sym++;
            if (n_words == need) {
                const uint32_t ppm = reduced ? P.sf - 2u : P.sf;
                if (t0) { uint32_t tmp = n_cw; deinterleave_block(sh, need, ppm, tmp); }
                n_cw = (n_cw + ppm <= (uint32_t)kMaxCodewords) ? n_cw + ppm : (uint32_t)kMaxCodewords;
                n_words = 0;
                block_done = true;
            }
        }
        if (is_first) {
            if (block_done) {
                if (P.implicit) {
                    payload_symbols = 1; // :829
                } else { // decode(true) and header parse (:831-847)
                    __syncthreads();
                    if (t0) {
                        uint8_t hA[3], hB[3], h0[3] = {0, 0, 0};
                        decode_header_bytes(sh, n_cw, 2, hA);
                        decode_header_bytes(sh, n_cw, 1, hB);
                        const uint8_t *use = (cr >= 3u) ? hA : (cr >= 1u ? hB : h0);
                        sh.hdr[0] = use[0]; sh.hdr[1] = use[1]; sh.hdr[2] = use[2];
                        sh.hdr[3] = (uint8_t)((hA[0] != hB[0]) || (hA[1] != hB[1]) || (hA[2] != hB[2]));
                        const uint32_t rem = n_cw > 5u ? n_cw - 5u : 0u; // erase the 5 header codewords (:632)
                        for (uint32_t i = 0; i < rem; i++) sh.cw[i] = sh.cw[i + 5u];
                    }
                    __syncthreads();
                    n_cw = n_cw > 5u ? n_cw - 5u : 0u;
                    { // (uniform: keeps the state machine in scalar registers)
                        const uint32_t hw = __builtin_amdgcn_readfirstlane(*reinterpret_cast<const uint32_t *>(sh.hdr));
                        phdr0 = (uint8_t)hw; phdr1 = (uint8_t)(hw >> 8); phdr2 = (uint8_t)(hw >> 16);
                        att_ambig = hw >> 24;
                    }
                    if ((phdr1 >> 5) > 4) phdr1 = (uint8_t)((phdr1 & 0x1f) | (4u << 5)); // :834-835
                    cr = phdr1 >> 5;
                    has_crc = (phdr1 >> 4) & 1u;
                    payload_length = (uint32_t)phdr0 + 2u * has_crc; // MAC_CRC_SIZE (:838)
                    const uint32_t redundancy = P.reduced_rate ? 2u : 0u; // :842-847
                    const int symbols_per_block = (int)cr + 4;
                    const float bits_needed = (float)payload_length * 8.0f;
                    const float symbols_needed = bits_needed * ((float)symbols_per_block / 4.0f) / (float)(P.sf - redundancy);
                    const int blocks_needed = (int)ceilf(symbols_needed / (float)symbols_per_block);
                    payload_symbols = blocks_needed * symbols_per_block;
                }
                state = kDecodePayload;
            }
        } else {
            if (block_done && !P.implicit) payload_symbols -= (int32_t)(4u + cr); // :866-867
            if (payload_symbols <= 0) { // :870-881
                uint32_t n_bytes;
                if (cr >= 3u) n_bytes = (uint32_t)ceilf((float)n_cw * 4.0f / (4.0f + (float)cr)); // :658
                else n_bytes = (n_cw + 1u) / 2u;
                if (n_bytes > (uint32_t)(kMaxCodewords / 2 + 8)) n_bytes = kMaxCodewords / 2 + 8;
                decode_payload_bytes(sh, n_cw, cr, n_bytes);
                const uint32_t plen = payload_length > 257u ? 257u : payload_length;
                AttemptRec &r = recs[n_att];
                for (uint32_t i = threadIdx.x; i < plen; i += (uint32_t)T) r.frame[3u + i] = (i < n_bytes) ? sh.dec[i] : 0;
                if (t0) {
                    r.frame[0] = phdr0; r.frame[1] = phdr1; r.frame[2] = phdr2; // d_phdr (:600)
                    r.frame_len = 3u + plen;
                }
                frame_ok = true;
                state = kDetect;
                n_words = 0; n_cw = 0;
            }
        }
    };

    while (true) {
        if (state == kDetect && !in_attempt) {
            if (pos >= job.scan_limit) { stop_reason = 0; break; }
            if (n_att >= C.recs_per_job) { stop_reason = 2; break; }
            if (job.max_attempts && n_att >= job.max_attempts) { stop_reason = 3; break; }
        }
        if (pos + 2 * (int64_t)sps > n_items) { stop_reason = 1; break; } // set_output_multiple(2*sps) (:91)
        const float2 *__restrict__ x = X + pos;
        int32_t consumed = 0, fine = 0, step_bin = -1; // d_fine_sync = 0 on every call (:749)
        float step_val = 0.0f;
        const int32_t st_in = state;
        const long long t_start = trace ? clock64() : 0;

        switch (state) {
        case kDetect: { // :752-768, detect_preamble_autocorr :340-366
            float a[4];
            w3_detect_window<SF>(x, ws, slot, a);
            energy_threshold = a[3] / 2.0f; // :357
            const float pushed = a[2] / (float)sps; // d_pwr_queue.push_back (:360)
            if (npush >= 4u) { push_tail[0] = push_tail[1]; push_tail[1] = push_tail[2]; push_tail[2] = push_tail[3]; push_tail[3] = pushed; }
            else {
                const uint32_t k = npush;
                if (k == 0u) push_tail[0] = pushed; else if (k == 1u) push_tail[1] = pushed; else if (k == 2u) push_tail[2] = pushed; else push_tail[3] = pushed;
            }
            npush++;
            const float s = sqrtf(a[2] * a[3]);
            const float autocorr = hypotf(a[0] / s, a[1] / s); // :363
            step_val = autocorr;
            if (autocorr >= 0.90f) { // :755
                corr_fails = 0u;
                state = kSync;
                in_attempt = true;
                att_trig = pos; att_hdr = -1; att_cr_prev = cr; att_ambig = 0; n_sym = 0;
            } else {
                consumed = (int32_t)sps;
            }
            break;
        }
        case kSync: { // :770-783
            const W3SyncOut so = w3_sync<SF>(P.sync_a, P.sync_b, x, &ws, slot);
            slot = __builtin_amdgcn_readfirstlane(so.slot);
            step_val = __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, so.bv)));
            const int sbi = __builtin_amdgcn_readfirstlane(so.bi);
            consumed = (sbi == 0x7fffffff) ? 0 : sbi; // `int i = 0` stays when nothing exceeds 0 (:771)
            state = kFindSfd;
            break;
        }
        case kFindSfd: { // :785-818
            const W3SfdOut fo = w3_sfd_window<SF>(W3SfdArgs{P.down_ifreq, P.up_ifreq_v, P.down_ifreq_avg, P.down_ifreq_sd, P.down_ifreq_dsum, P.sync_a, P.sync_b}, x, &ws, slot);
            slot = __builtin_amdgcn_readfirstlane(fo.slot);
            const float c = __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, fo.c)));
            const int32_t fs = __builtin_amdgcn_readfirstlane(fo.fine);
            step_val = c;
            if (c > 0.96f) { // :792
                state = kPause;
            } else {
                if (c < -0.97f) fine = fs; // :801-803
                else corr_fails++;
                if (corr_fails > 4u) state = kDetect; // :808-809
            }
            consumed = (int32_t)sps + fine; // :816
            break;
        }
        case kPause: // :820-824
            state = kDecodeHeader;
            consumed = (int32_t)(sps + P.delay_after_sync);
            att_hdr = pos + consumed;
            break;
        case kDecodeHeader:
        case kDecodePayload: { // :826-886, demodulate :493-529
            const bool is_first = state == kDecodeHeader;
            const bool want_energy = !is_first && P.implicit; // determine_energy (:861-864)
            uint32_t s;
            int32_t fs;
            float en;
            w3_demod_symbol<SF>(P, L, x, want_energy, slot, s, fs, en);
            bool do_demod = true;
            if (want_energy && en < energy_threshold) { payload_symbols = 0; payload_length = n_cw / 2u; do_demod = false; }
            uint32_t bin_idx = 0;
            if (do_demod) { // :500, bin_idx = (s-1) mod N; compat keeps the s==0 -> 0 quirk of the gradient path
                bin_idx = (s == 0u && P.demod_mode == 2u) ? 0u : (s + P.nbins - 1u) % P.nbins;
                step_bin = (int32_t)bin_idx;
                fine = fs; // :501-502
            }
            post_symbol(do_demod, bin_idx, is_first);
            consumed = (int32_t)sps + fine; // :856,:883
            break;
        }
        default:
            consumed = (int32_t)sps;
            break;
        }
        if (end_step(st_in, consumed, step_bin, fine, step_val, t_start)) break;
    }

    if (in_attempt && n_att < C.recs_per_job && t0) {
        AttemptRec &r = recs[n_att];
        r.status = (stop_reason == 3u) ? kAttemptAtHeader : kAttemptOutOfData;
        r.start_pos = att_start; r.trig_pos = att_trig; r.hdr_pos = att_hdr; r.end_pos = pos;
        r.npush = npush;
        for (int i = 0; i < 4; i++) r.push_tail[i] = push_tail[i];
        r.cr_prev = att_cr_prev; r.hdr_ambig = att_ambig; r.n_symbols = n_sym; r.frame_len = 0;
    }
    if (t0) {
        JobResult &jr = C.results[jid];
        jr.final_pos = in_attempt ? att_start : pos;
        jr.n_attempts = n_att + (in_attempt ? 1u : 0u);
        jr.final_cr = cr;
        jr.npush = in_attempt ? 0u : npush;
        for (int i = 0; i < 4; i++) jr.push_tail[i] = push_tail[i];
        jr.stop_reason = stop_reason;
        jr.n_steps = n_steps < C.trace_cap ? n_steps : C.trace_cap;
        jr.pad = in_attempt ? 1u : 0u;
        for (int i = 0; i < 6; i++) { jr.cyc[i] = 0; jr.rounds[i] = 0; }
        jr.tail_valid = 0;
    }
}

__global__ __launch_bounds__(W3Geom<9>::T, 4) void walker3_kernel_sf9(DevParams P, LaunchCfg C) { walker3_body<9>(P, C); }
__global__ __launch_bounds__(W3Geom<10>::T, 4) void walker3_kernel_sf10(DevParams P, LaunchCfg C) { walker3_body<10>(P, C); }
__global__ __launch_bounds__(W3Geom<11>::T, 4) void walker3_kernel_sf11(DevParams P, LaunchCfg C) { walker3_body<11>(P, C); }
__global__ __launch_bounds__(W3Geom<12>::T, 4) void walker3_kernel_sf12(DevParams P, LaunchCfg C) { walker3_body<12>(P, C); }

// ---- symbol-level kernel: one workgroup per symbol, for lora_hip_demod_symbols_device --------------------------
template <int SF>
__global__ __launch_bounds__(W3Geom<SF>::T, 4) void demod_symbols_w3_kernel(DevParams P, const float2 *iq, const int64_t *offsets, uint32_t n,
                                                                             uint32_t *bins, int32_t *fine)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const W3Lds<SF> L = w3_carve<SF>(smem);
    w3_tables_to_lds<SF>(P, L);
    __syncthreads();
    int slot = 0;
    for (uint32_t s = blockIdx.x; s < n; s += gridDim.x) {
        uint32_t b;
        int32_t fs;
        float en;
        w3_demod_symbol<SF>(P, L, iq + offsets[s], false, slot, b, fs, en);
        if (threadIdx.x == 0) { bins[s] = b; if (fine) fine[s] = fs; }
        __syncthreads();
    }
}

// ---- host side: tables -----------------------------------------------------------------------------------------
// w3_tw: W_N^t, t < NTW.  w3_ctab: the combine coefficient of value i of thread t3 in round g at [(g 16 + i) T + t3]:
// W_sps^{k r} for the signed bin k of k1 = a + 16 a2 + 256 b2 (+ the fold at k1 = N/2, :450).
template <int SF>
static void build_w3_tables_sf(float2 *tw, float2 *ctab)
{
    using G = W3Geom<SF>;
    constexpr int N = G::N, SPS = G::SPS, T = G::T;
    for (int t = 0; t < G::NTW; t++) {
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
uint32_t w3_tw_entries(uint32_t sf) { return sf == 9u ? W3Geom<9>::NTW : sf == 10u ? W3Geom<10>::NTW : sf == 11u ? W3Geom<11>::NTW : sf == 12u ? W3Geom<12>::NTW : 0u; }
void build_w3_tables(uint32_t sf, float2 *tw, float2 *ctab /* sps entries */)
{
    if (sf == 9u) build_w3_tables_sf<9>(tw, ctab);
    else if (sf == 10u) build_w3_tables_sf<10>(tw, ctab);
    else if (sf == 11u) build_w3_tables_sf<11>(tw, ctab);
    else if (sf == 12u) build_w3_tables_sf<12>(tw, ctab);
}

static uint32_t walker3_threads(uint32_t sf) { return sf == 9u ? W3Geom<9>::T : sf == 10u ? W3Geom<10>::T : 1024u; }
static uint32_t walker3_lds(uint32_t sf) { return sf == 9u ? w3_lds_bytes<9>() : sf == 10u ? w3_lds_bytes<10>() : sf == 11u ? w3_lds_bytes<11>() : w3_lds_bytes<12>(); }
