// lora_wave_decim.inc.hip -- the wave-per-symbol demodulators at decimation D = 2 and 4 (BW 500 / 250 kHz at 1 Msps, 250 / 125 kHz at
// 500 ksps ...): decoder::make takes any samp_rate / bandwidth (lib/decoder_impl.cc:57, :79-87), and until round 6 everything but D = 8 ran
// the generic kernels.  Included by lora_kernels.hip behind lora_wave_demod.inc.hip, whose helpers it uses.
//
// Same plan as wave_demod_symbol (get_shift_fft, :430-464, + fine_sync, :300-338): sps = D N samples, a lane owns n = 64 j + lane, j < J =
// sps / 64 - every load covers 512 contiguous bytes.  What changes with D is where the lane index splits into polyphase branch and position:
// n = D q + r with r = lane & (D - 1) and q = LQ j + lq, lq = lane >> LD, LQ = 64 / D lanes per branch.  The N-point FFT of a branch is the
// J-point DIF in registers, the twiddle W_N^{lq k1} and an LQ-point DIF over lane bits 5 .. LD - two more (D = 4: one more) cross-lane
// stages than at D = 8: bits 5 and 4 as register-pair transposes (v_permlane{32,16}_swap), the rest as butterflies on a DPP operand, each
// lane its own output, with a lane twiddle behind every stage but the last.  The polyphase combine is a reduce-scatter over the LD low bits.
// tools/wave_decim_model.py is this dataflow in numpy, register by register, against the pruned DFT (tests/test_wave_decim_model.py);
// wave_layout_bin_d below and the host's tables follow it.
//
// fine_sync's ifreq comes from the registers loaded for the dechirp (the ZM evaluation: from memory); the closed
// form of the D = 8 kernels is not used here (its bounds are derived for |ifreq| <= pi / 8 of a D = 8 chirp).  A window with a sample of exactly zero comes back as
// kPoisonBin and is evaluated again by the ZM instantiation, as everywhere.

template <int SF, int LD> struct WaveGeomD {
    static_assert(LD >= 1 && LD <= 3, "decimation 2, 4 - or 8 for SF6, which lora_wave_demod.inc.hip's SF7-SF9 instantiations do not cover");
    static constexpr int N = 1 << SF, D = 1 << LD, SPS = N << LD, J = SPS / 64, LQ = 64 >> LD, LOGJ = ilog2(J), NXT = 5 - LD;
    static_assert(J >= 4 && J <= 64, "4 .. 64 samples per lane");
    static constexpr uint32_t n_down = SPS, n_twn = J * LQ, n_tws = J * 64, n_xst = NXT * 64;
    static constexpr uint32_t n_ent = n_down + n_twn + n_tws + n_xst; // 8-byte entries of the packed table block
    static constexpr uint32_t n_v = (3u * SPS + 40u + 3u) & ~3u;      // floats of the ifreq template behind it
    static constexpr uint32_t lds_bytes = n_ent * 8u + n_v * 4u;
};

// Bin held by register g of `lane` after the cross-lane FFT (tools/wave_decim_model.py layout_bin): stages 1 and 2 leave the output bit in
// the register slot (s, t) and move lane bits 5, 4 into the in-lane register index; lane bit b of the later stages is output bit 5 - b
__device__ __host__ constexpr int wave_layout_bin_d(int J, int logj, int ld, int g, int lane)
{
    const int b5 = (lane >> 5) & 1, b4 = (lane >> 4) & 1;
    const int s = g / (J / 2), t = (g / (J / 4)) & 1, i = g % (J / 4);
    const int e = i + (J / 4) * b4 + (J / 2) * b5;
    int k2 = s + 2 * t;
    for (int b = 3; b >= ld; b--) k2 += ((lane >> b) & 1) << (5 - b);
    return brev_bits(e, logj) + J * k2;
}

// lane ^ 4 of both halves: row_shl:4 into banks 0, 2, row_shr:4 into banks 1, 3
__device__ __forceinline__ v2f dpp2_xor4(v2f v)
{
    const float x = v.x, y = v.y;
    v2f r;
    r.x = dpp_rows_bank_f<0x114, 0xA>(dpp_rows_bank_f<0x104, 0x5>(x, x), x);
    r.y = dpp_rows_bank_f<0x114, 0xA>(dpp_rows_bank_f<0x104, 0x5>(y, y), y);
    return r;
}

template <int SF, int LD>
__device__ __forceinline__ WaveTabs wave_tabs_to_lds_d(const DevParams &P, v2f *l2, float *lv, uint32_t nthreads)
{ // [down | twn | tws | xst] -> l2 (n_ent entries), the ifreq template -> lv (n_v floats)
    using G = WaveGeomD<SF, LD>;
    const v2f *__restrict__ src = reinterpret_cast<const v2f *>(P.wave_tabs);
    for (uint32_t i = threadIdx.x; i < G::n_ent; i += nthreads) l2[i] = src[i];
    for (uint32_t i = threadIdx.x; i < 3u * G::SPS + 40u; i += nthreads) lv[i] = P.up_ifreq_v[i];
    WaveTabs T{};
    T.down = l2; T.twn = l2 + G::n_down; T.tws = T.twn + G::n_twn; T.xst = T.tws + G::n_tws; T.v = lv;
    return T;
}
template <int SF, int LD>
__device__ __forceinline__ WaveTabs wave_tabs_to_lds_d(const DevParams &P, unsigned char *lds, uint32_t nthreads)
{ // WaveGeomD::lds_bytes bytes
    v2f *l2 = reinterpret_cast<v2f *>(lds);
    return wave_tabs_to_lds_d<SF, LD>(P, l2, reinterpret_cast<float *>(l2 + WaveGeomD<SF, LD>::n_ent), nthreads);
}

// fine_sync (:300-338) with search = max(D / 4, 2) = 2: lags -1, 0, +1, the window's ifreq from memory.  ZM: every value as the reference forms it
// next to a sample of exactly zero (ifreq_prod_z), rolled.  Returns false when the sums are poisoned (a zero sample met by the fast evaluation).
template <int SPS, bool ZM>
__device__ __forceinline__ bool wave_fine_sync_reload(const float2 *__restrict__ x, const float *__restrict__ v, int lane, int32_t &fine_out)
{
    constexpr int J = SPS / 64;
    const v2f *__restrict__ xv = reinterpret_cast<const v2f *>(x);
    // sample n = 64 j + lane carries ifreq[k], k = n - 1, against v[k - 1], v[k], v[k + 1] (lane 0, j = 0 has no k: its f is 0 and it reads the finite entries in front of v)
    const float *__restrict__ vp = v + (lane - 2);
    float c0 = 0.f, c1 = 0.f, c2 = 0.f;
    if constexpr (ZM) {
#pragma unroll 1
        for (int j = 0; j < J; j++) {
            const int n = 64 * j + lane;
            const float fj = n >= 1 ? ifreq_prod_z(x[n - 1], x[n]) : 0.0f;
            c0 += fj * vp[64 * j]; c1 += fj * vp[64 * j + 1]; c2 += fj * vp[64 * j + 2];
            if (j == J - 1 && lane == 63) { c0 += fj * vp[64 * j + 1]; c1 += fj * vp[64 * j + 2]; c2 += fj * vp[64 * j + 3]; } // ifreq[sps-1] = ifreq[sps-2] (:243)
        }
    } else {
#pragma unroll
        for (int j = 0; j < J; j += 2) {
            const int n0 = j * 64 + lane, n1 = n0 + 64;
            const v2f fp = ifreq_prod_pk(xv[n0 >= 1 ? n0 - 1 : 0], xv[n0], xv[n1 - 1], xv[n1]);
            const float f0 = (n0 >= 1) ? fp.x : 0.0f, f1 = fp.y;
            c0 += f0 * vp[64 * j]; c1 += f0 * vp[64 * j + 1]; c2 += f0 * vp[64 * j + 2];
            c0 += f1 * vp[64 * j + 64]; c1 += f1 * vp[64 * j + 65]; c2 += f1 * vp[64 * j + 66];
            if (j == J - 2) { // the lane that owns n = sps-1 adds the duplicated tap (:243)
                const float fl = (lane == 63) ? f1 : 0.0f;
                c0 += fl * vp[64 * j + 65]; c1 += fl * vp[64 * j + 66]; c2 += fl * vp[64 * j + 67];
            }
        }
    }
    c0 = wave_sum_u(c0); c1 = wave_sum_u(c1); c2 = wave_sum_u(c2);
    if (!ZM && poisoned3(c0, c1, c2)) return false; // (uniform)
    float mx = 0.0f;
    int32_t lag = 0;
    if (c0 > mx) { mx = c0; lag = -1; }
    if (c1 > mx) { mx = c1; lag = 0; }
    if (c2 > mx) { mx = c2; lag = 1; }
    fine_out = -lag;
    return true;
}

// ... and from ifreq values that are in registers already: f[j] = ifreq[64 j + lane - KOFF].  KOFF = 1: the FFT demodulator's predecessor form (lane 0 of row 0 holds 0; the
// duplicated last tap, :243, is added here); KOFF = 0: the gradient demodulator's successor form (lane 63 of the last row holds the duplicate already)
template <int J, int KOFF>
__device__ __forceinline__ bool wave_fine_sync_regs(const float (&f)[J], const float *__restrict__ v, int lane, int32_t &fine_out)
{
    const float *__restrict__ vp = v + (lane - 1 - KOFF);
    float c0 = 0.f, c1 = 0.f, c2 = 0.f;
#pragma unroll
    for (int j = 0; j < J; j++) {
        const float fj = f[j];
        c0 += fj * vp[64 * j]; c1 += fj * vp[64 * j + 1]; c2 += fj * vp[64 * j + 2];
    }
    if constexpr (KOFF == 1) {
        const float fl = (lane == 63) ? f[J - 1] : 0.0f;
        c0 += fl * vp[64 * (J - 1) + 1]; c1 += fl * vp[64 * (J - 1) + 2]; c2 += fl * vp[64 * (J - 1) + 3];
    }
    c0 = wave_sum_u(c0); c1 = wave_sum_u(c1); c2 = wave_sum_u(c2);
    if (poisoned3(c0, c1, c2)) return false; // (uniform) a sample of the window is exactly zero
    float mx = 0.0f;
    int32_t lag = 0;
    if (c0 > mx) { mx = c0; lag = -1; }
    if (c1 > mx) { mx = c1; lag = 0; }
    if (c2 > mx) { mx = c2; lag = 1; }
    fine_out = -lag;
    return true;
}

template <int SF, int LD, bool ZM = false>
__device__ __forceinline__ void wave_demod_symbol_d(const DevParams &P, const WaveTabs &T, const float2 *__restrict__ x, uint32_t &s_out, int32_t &fine_out,
                                                    float *en_out = nullptr /* implicit header: the window's energy (determine_energy, :368-375) */)
{
    using G = WaveGeomD<SF, LD>;
    constexpr int N = G::N, D = G::D, J = G::J, SPS = G::SPS, LOGJ = G::LOGJ, LQ = G::LQ;
    int lane = threadIdx.x & 63;
    asm volatile("" : "+v"(lane)); // (as in wave_demod_symbol: keeps the per-lane table addresses out of the caller's loop-invariant set)
    const int lq = lane >> LD;
    const v2f *__restrict__ xv = reinterpret_cast<const v2f *>(x);

    v2f a[J];
#pragma unroll
    for (int j = 0; j < J; j++) a[j] = xv[j * 64 + lane];
    if (en_out) { // (a uniform branch: only implicit-header decoders ask)
        v2f e2 = (v2f){0.0f, 0.0f};
#pragma unroll
        for (int j = 0; j < J; j++) e2 = __builtin_elementwise_fma(a[j], a[j], e2);
        *en_out = wave_sum_u(e2.x + e2.y);
    }
    // fine_sync's ifreq from the registers loaded for the dechirp (sample n - 1 sits in the neighbouring lane, lane 0's in lane 63 of the previous register - one
    // wave rotate per register; measured against a second, cache-hot read behind the FFT: walkers +6 ... +12 % at decimation 4).  J = 32 (SF9 at decimation 4) runs
    // at the 256-register budget, where the 32 values fit beside the FFT's
    constexpr bool EARLY_F = !ZM;
    float f[EARLY_F ? J : 1];
    if constexpr (EARLY_F) {
        if (P.enable_fine_sync != 0u) {
            v2f bprev = (v2f){0.0f, 0.0f};
#pragma unroll
            for (int j = 0; j < J; j += 2) {
                const v2f b0 = dpp2<kDppWaveRor1>(a[j]), b1 = dpp2<kDppWaveRor1>(a[j + 1]);
                const v2f p0 = (lane == 0) ? bprev : b0, p1 = (lane == 0) ? b0 : b1;
                bprev = b1;
                const v2f fp = ifreq_prod_pk(p0, a[j], p1, a[j + 1]);
                f[j] = (j == 0 && lane == 0) ? 0.0f : fp.x; // n = 0 has no predecessor in the window
                f[j + 1] = fp.y;
            }
        }
    }
#pragma unroll
    for (int j = 0; j < J; j++) a[j] = cmul2(a[j], T.down[j * 64 + lane]); // dechirp (:437)
    fft_inlane_dif_pk<J>(a);
#pragma unroll
    for (int m = 1; m < J; m++) a[m] = cmul2(a[m], T.twn[m * LQ + lq]); // W_N^{lq k1}
    { // LQ-point DIF over lq = lane bits 5 .. LD
        const v2f w1 = T.xst[lane], w2 = T.xst[64 + lane]; // W_LQ^{lq mod LQ/2}, W_{LQ/2}^{lq mod LQ/4}: the same on both lanes of a pair
#pragma unroll
        for (int i = 0; i < J / 2; i++) { // lane bit 5
            const float dx = a[i].x, dy = a[i].y, sx = a[i + J / 2].x, sy = a[i + J / 2].y;
            const auto px = __builtin_amdgcn_permlane32_swap(__builtin_bit_cast(int, dx), __builtin_bit_cast(int, sx), false, false);
            const auto py = __builtin_amdgcn_permlane32_swap(__builtin_bit_cast(int, dy), __builtin_bit_cast(int, sy), false, false);
            const int x0 = px[0], x1 = px[1], y0 = py[0], y1 = py[1]; // scalars first: bit_cast on a vector element is miscompiled
            const v2f lo = (v2f){__builtin_bit_cast(float, x0), __builtin_bit_cast(float, y0)};
            const v2f hi = (v2f){__builtin_bit_cast(float, x1), __builtin_bit_cast(float, y1)};
            a[i] = lo + hi;
            a[i + J / 2] = cmul2(lo - hi, w1);
        }
#pragma unroll
        for (int h = 0; h < 2; h++)
#pragma unroll
            for (int i = 0; i < J / 4; i++) { // lane bit 4
                const int g = i + h * (J / 2);
                const float dx = a[g].x, dy = a[g].y, sx = a[g + J / 4].x, sy = a[g + J / 4].y;
                const auto px = __builtin_amdgcn_permlane16_swap(__builtin_bit_cast(int, dx), __builtin_bit_cast(int, sx), false, false);
                const auto py = __builtin_amdgcn_permlane16_swap(__builtin_bit_cast(int, dy), __builtin_bit_cast(int, sy), false, false);
                const int x0 = px[0], x1 = px[1], y0 = py[0], y1 = py[1];
                const v2f lo = (v2f){__builtin_bit_cast(float, x0), __builtin_bit_cast(float, y0)};
                const v2f hi = (v2f){__builtin_bit_cast(float, x1), __builtin_bit_cast(float, y1)};
                a[g] = lo + hi;
                a[g + J / 4] = cmul2(lo - hi, w2);
            }
        if constexpr (LD == 3) { // lane bit 3: the last stage (decimation 8)
            const v2f sg = (lane & 8) ? (v2f){-1.f, -1.f} : (v2f){1.f, 1.f};
            xstage_last_pk<kDppRor8, J>(a, sg);
        } else { // lane bit 3: a' = +-a + partner, then the lane twiddle (1 on the lower lane of a pair)
            const v2f sg = (lane & 8) ? (v2f){-1.f, -1.f} : (v2f){1.f, 1.f};
            const v2f w3 = T.xst[2 * 64 + lane];
#pragma unroll
            for (int m = 0; m < J; m++) a[m] = cmul2(__builtin_elementwise_fma(sg, a[m], dpp2<kDppRor8>(a[m])), w3);
        }
        if constexpr (LD == 3) {
        } else if constexpr (LD == 2) { // lane bit 2: the last stage
            const v2f sg = (lane & 4) ? (v2f){-1.f, -1.f} : (v2f){1.f, 1.f};
#pragma unroll
            for (int m = 0; m < J; m++) a[m] = __builtin_elementwise_fma(sg, a[m], dpp2_xor4(a[m]));
        } else {
            {
                const v2f sg = (lane & 4) ? (v2f){-1.f, -1.f} : (v2f){1.f, 1.f};
                const v2f w4 = T.xst[3 * 64 + lane];
#pragma unroll
                for (int m = 0; m < J; m++) a[m] = cmul2(__builtin_elementwise_fma(sg, a[m], dpp2_xor4(a[m])), w4);
            }
            const v2f sg = (lane & 2) ? (v2f){-1.f, -1.f} : (v2f){1.f, 1.f}; // lane bit 1: the last stage
            xstage_last_pk<kDppQuadXor2, J>(a, sg);
        }
    }
#pragma unroll
    for (int m = 0; m < J; m++) a[m] = cmul2(a[m], T.tws[m * 64 + lane]); // W_sps^{k r} (+ fold)
    // reduce-scatter over r = the LD low lane bits: lanes with the bit clear keep the first half of the registers
    constexpr int R = J >> LD;
    v2f b1[R];
    if constexpr (LD == 3) { // (as wave_demod_symbol: lane bit 2 by row_shr:4 into banks 1,3 / row_shl:4 into banks 0,2, then bits 1, 0)
        v2f b4[J / 2], b2[J / 4];
#pragma unroll
        for (int i = 0; i < J / 2; i++) {
            const float lx = a[i].x, ly = a[i].y, hx = a[i + J / 2].x, hy = a[i + J / 2].y;
            v2f X, Y;
            X.x = dpp_rows_bank_f<0x114, 0xA>(lx, hx); X.y = dpp_rows_bank_f<0x114, 0xA>(ly, hy); // upper lanes: partner's second half
            Y.x = dpp_rows_bank_f<0x104, 0x5>(hx, lx); Y.y = dpp_rows_bank_f<0x104, 0x5>(hy, ly); // lower lanes: partner's first half
            b4[i] = X + Y;
        }
        const bool hi2 = (lane & 2) != 0, hi1 = (lane & 1) != 0;
#pragma unroll
        for (int i = 0; i < J / 4; i++) {
            const v2f t0 = b4[i] + dpp2<kDppQuadXor2>(b4[i]);
            const v2f t1 = b4[i + J / 4] + dpp2<kDppQuadXor2>(b4[i + J / 4]);
            b2[i] = hi2 ? t1 : t0;
        }
#pragma unroll
        for (int i = 0; i < J / 8; i++) {
            const v2f t0 = b2[i] + dpp2<kDppQuadXor1>(b2[i]);
            const v2f t1 = b2[i + J / 8] + dpp2<kDppQuadXor1>(b2[i + J / 8]);
            b1[i] = hi1 ? t1 : t0;
        }
    } else if constexpr (LD == 2) {
        v2f b2[J / 2];
        const bool hi2 = (lane & 2) != 0, hi1 = (lane & 1) != 0;
#pragma unroll
        for (int i = 0; i < J / 2; i++) {
            const v2f t0 = a[i] + dpp2<kDppQuadXor2>(a[i]);
            const v2f t1 = a[i + J / 2] + dpp2<kDppQuadXor2>(a[i + J / 2]);
            b2[i] = hi2 ? t1 : t0;
        }
#pragma unroll
        for (int i = 0; i < J / 4; i++) {
            const v2f t0 = b2[i] + dpp2<kDppQuadXor1>(b2[i]);
            const v2f t1 = b2[i + J / 4] + dpp2<kDppQuadXor1>(b2[i + J / 4]);
            b1[i] = hi1 ? t1 : t0;
        }
    } else {
        const bool hi1 = (lane & 1) != 0;
#pragma unroll
        for (int i = 0; i < J / 2; i++) {
            const v2f t0 = a[i] + dpp2<kDppQuadXor1>(a[i]);
            const v2f t1 = a[i + J / 2] + dpp2<kDppQuadXor1>(a[i + J / 2]);
            b1[i] = hi1 ? t1 : t0;
        }
    }
    // arg-max on |X|^2 (monotone in the reference's std::abs, :454), first maximum in bin order wins (:463)
    const int gbase = LD == 3 ? ((lane & 4) ? J / 2 : 0) + ((lane & 2) ? J / 4 : 0) + ((lane & 1) ? J / 8 : 0)
                    : LD == 2 ? ((lane & 2) ? J / 2 : 0) + ((lane & 1) ? J / 4 : 0) : ((lane & 1) ? J / 2 : 0);
    float bv = -1.0f;
    int bi = 0x7fffffff;
#pragma unroll
    for (int i = 0; i < R; i++) {
        const int jb = wave_layout_bin_d(J, LOGJ, LD, gbase + i, lane);
        const float mag = b1[i].x * b1[i].x + b1[i].y * b1[i].y;
        if (mag > bv || (mag == bv && jb < bi)) { bv = mag; bi = jb; }
    }
    const float best = wave_max_nonneg_u(bv);
    const uint32_t s = (uint32_t)wave_min_u(bv == best ? bi : 0x7fffffff);
    s_out = s;
    fine_out = 0;
    if (P.enable_fine_sync == 0u) return;
    const uint32_t bin_idx = (s == 0u && P.demod_mode == 2u) ? 0u : (s + (uint32_t)N - 1u) % (uint32_t)N;
    if constexpr (EARLY_F) {
        if (!wave_fine_sync_regs<J, 1>(f, T.v + ((int)(bin_idx + 1u) * D + SPS), lane, fine_out)) s_out = kPoisonBin;
    } else {
        int zero = 0; // (the second read stays behind the reduce-scatter: hoisted above the FFT it would hold 2 J more registers)
        asm volatile("; fine-sync reload after the reduce-scatter" : "+v"(zero) : "v"(bv));
        if (!wave_fine_sync_reload<SPS, ZM>(x + zero, T.v + ((int)(bin_idx + 1u) * D + SPS), lane, fine_out)) s_out = kPoisonBin;
    }
}

// max_frequency_gradient_idx (:466-491) + fine_sync on one wavefront, as wave_demod_symbol_grad: f[j] = ifreq[n], n = 64 j + lane; the D samples of
// bin i = LQ j + (lane >> LD) sit in one aligned group of D lanes: the bin average is LD adds on a DPP operand, its left neighbour one lane permute.
template <int SF, int LD, bool ZM = false>
__device__ __forceinline__ void wave_demod_symbol_grad_d(const DevParams &P, const float *__restrict__ Tv, const float2 *__restrict__ x, bool want_energy,
                                                         uint32_t &bin_out, int32_t &fine_out, float &en_out)
{
    using G = WaveGeomD<SF, LD>;
    constexpr int N = G::N, D = G::D, J = G::J, SPS = G::SPS, LQ = G::LQ;
    int lane = threadIdx.x & 63;
    asm volatile("" : "+v"(lane));
    const auto xv = (const __attribute__((address_space(1))) v2f *)x;
    const int m = lane >> LD;
    const int perm_addr = ((lane - D) & 63) << 2;
    float bv = 0.1f; // max_gradient = 0.1f (:479)
    int bi = 0x7fffffff;
    float gs = 0.0f; // (carries the poison of a zero sample when there is no fine_sync sum to carry it)
    float prev_perm = 0.0f;
    float f[ZM ? 1 : J]; // this lane's ifreq[64 j + lane]: the bin averages and fine_sync's sums both come from these registers
    en_out = 0.0f;
    auto bin_step = [&](int j, float A) { // A: this lane's ifreq[64 j + lane]
        // the bin's D values added in the reference's order (volk_32f_accumulator_s32f's plain loop, :475: ((f0 + f1) + f2) + f3) - every lane of the group
        // reads them by quad broadcast: a window cut half a bin off its symbol (two samples at D = 4) shares the drop between two neighbouring
        // differences that tie up to this rounding
        if constexpr (LD == 3) { A += dpp_f<kDppQuadXor1>(A); A += dpp_f<kDppQuadXor2>(A); A += dpp_f<kDppRowHalfMirror>(A); } // (decimation 8: the tree of wave_demod_symbol_grad)
        else if constexpr (LD == 2) A = ((dpp_f<0x00>(A) + dpp_f<0x55>(A)) + dpp_f<0xAA>(A)) + dpp_f<0xFF>(A);
        else A += dpp_f<kDppQuadXor1>(A); // (two values: the order does not matter)
        A *= 1.0f / (float)D; // / d_decim_factor (a power of two)
        const float perm = __builtin_bit_cast(float, __builtin_amdgcn_ds_bpermute(perm_addr, __builtin_bit_cast(int, A))); // bin (m - 1) mod LQ of this register
        const float left = (m == 0) ? prev_perm : perm; // bin i - 1: for m = 0 the last bin of the previous register
        prev_perm = perm;
        const float g = left - A; // samples_ifreq_avg[i - 1] - samples_ifreq_avg[i]
        gs += A;
        const int i = LQ * j + m;
        if ((j > 0 || m > 0) && g > bv) { bv = g; bi = i; } // i runs from 1; strict '>' keeps the first maximum
    };
    if constexpr (ZM) {
        if (want_energy) {
            v2f e2 = (v2f){0.0f, 0.0f};
#pragma unroll 1
            for (int j = 0; j < J; j++) { const v2f aj = xv[j * 64 + lane]; e2 = __builtin_elementwise_fma(aj, aj, e2); }
            en_out = wave_sum_u(e2.x + e2.y);
        }
#pragma unroll 1
        for (int j = 0; j < J; j++) {
            int n = 64 * j + lane;
            n = n == SPS - 1 ? SPS - 2 : n; // ifreq[sps-1] = ifreq[sps-2] (:243)
            bin_step(j, ifreq_prod_z(x[n], x[n + 1]));
        }
    } else {
        v2f a[J];
#pragma unroll
        for (int j = 0; j < J; j++) a[j] = xv[j * 64 + lane];
        __builtin_amdgcn_sched_barrier(0); // all loads issued before the first use
        if (want_energy) {
            v2f e2 = (v2f){0.0f, 0.0f};
#pragma unroll
            for (int j = 0; j < J; j++) e2 = __builtin_elementwise_fma(a[j], a[j], e2);
            en_out = wave_sum_u(e2.x + e2.y);
        }
        v2f cn = dpp2<kDppWaveRol1>(a[0]); // a[j] of lane + 1 (lane 63: of lane 0)
#pragma unroll
        for (int j = 0; j < J; j += 2) {
            const v2f c0 = cn, c1 = dpp2<kDppWaveRol1>(a[j + 1]);
            const v2f c2 = (j + 2 < J) ? dpp2<kDppWaveRol1>(a[j + 2]) : c1;
            cn = c2;
            // x[n + 1]: the neighbouring lane's sample of the same register; for lane 63 lane 0's sample of the NEXT register
            const v2f s0 = (lane == 63) ? c1 : c0, s1 = (lane == 63) ? c2 : c1;
            const v2f fp = ifreq_prod_pk(a[j], s0, a[j + 1], s1);
            f[j] = fp.x; f[j + 1] = fp.y;
        }
        const float dup = dpp_f<kDppWaveRor1>(f[J - 1]); // ifreq[sps-1] = ifreq[sps-2] (:243)
        f[J - 1] = (lane == 63) ? dup : f[J - 1];
#pragma unroll
        for (int j = 0; j < J; j++) bin_step(j, f[j]);
    }
    const float best = wave_max_nonneg_u(bv);
    const int first = wave_min_u((bv == best) ? bi : 0x7fffffff);
    const uint32_t max_index = (first == 0x7fffffff) ? 0u : (uint32_t)first + 1u; // :486
    const uint32_t bin_idx = ((uint32_t)N - max_index) % (uint32_t)N;              // :490
    bin_out = bin_idx;
    fine_out = 0;
    if (P.enable_fine_sync == 0u) {
        if (!ZM && poisoned(wave_sum_u(gs))) bin_out = kPoisonBin; // (a sample of exactly zero in the window)
        return;
    }
    if constexpr (ZM) (void)wave_fine_sync_reload<SPS, true>(x, Tv + ((int)(bin_idx + 1u) * D + SPS), lane, fine_out);
    else if (!wave_fine_sync_regs<J, 0>(f, Tv + ((int)(bin_idx + 1u) * D + SPS), lane, fine_out)) bin_out = kPoisonBin;
}

// host side: the table block in the layout above, from the handle's downchirp
static void build_wave_tables_d_host(int sf, int ld, const float2 *down, float *out)
{
    const int N = 1 << sf, D = 1 << ld, SPS = N * D, J = SPS / 64, LQ = 64 / D;
    const int logj = ilog2(J);
    size_t o = 0;
    auto put = [&](double re, double im) { out[2 * o + 0] = (float)re; out[2 * o + 1] = (float)im; o++; };
    auto putw = [&](long long num, long long den) { // W_den^num
        const double a = -2.0 * M_PI * (double)(((num % den) + den) % den) / (double)den;
        put(std::cos(a), std::sin(a));
    };
    for (int n = 0; n < SPS; n++) put(down[n].x, down[n].y);
    for (int m = 0; m < J; m++)
        for (int lq = 0; lq < LQ; lq++) putw((long long)lq * brev_bits(m, logj), N);
    for (int g = 0; g < J; g++)
        for (int lane = 0; lane < 64; lane++) {
            const int r = lane & (D - 1);
            const int jb = wave_layout_bin_d(J, logj, ld, g, lane);
            const int k = (jb < N / 2) ? jb : jb - N;
            const int e = ((k * r) % SPS + SPS) % SPS;
            const double ang = -2.0 * M_PI * (double)e / (double)SPS;
            double re = std::cos(ang), im = std::sin(ang);
            if (jb == N / 2) { // tmp[N/2] += F[N/2] (:450)
                const double a2 = -2.0 * M_PI * (double)(((N / 2) * r) % SPS) / (double)SPS;
                re += std::cos(a2); im += std::sin(a2);
            }
            put(re, im);
        }
    for (int lane = 0; lane < 64; lane++) putw((lane >> ld) % (LQ / 2), LQ);     // stage 1 (lane bit 5)
    for (int lane = 0; lane < 64; lane++) putw((lane >> ld) % (LQ / 4), LQ / 2); // stage 2 (lane bit 4)
    for (int bit = 3; bit > ld; bit--) {                                            // the stages with a twiddle behind them: the upper lane of a pair
        const int span = 1 << (bit - ld);
        for (int lane = 0; lane < 64; lane++) {
            if ((lane >> bit) & 1) putw((lane >> ld) % span, 2 * span);
            else put(1.0, 0.0);
        }
    }
}

// SF7-SF9 at decimation 2 / 4; SF6 (implicit header: an SF6 header block holds 4 of the header's 5 codewords) at 4 / 8
static bool wave_decim_covers(uint32_t sf, uint32_t decim) { return ((decim == 2u || decim == 4u) && sf >= 7u && sf <= 9u) || (sf == 6u && (decim == 4u || decim == 8u)); }
uint32_t wave_tables_floats_d(uint32_t sf, uint32_t decim)
{
    if (!wave_decim_covers(sf, decim)) return 0u;
    const uint32_t sps = decim << sf, J = sps / 64u, LQ = 64u / decim, ld = decim == 2u ? 1u : decim == 4u ? 2u : 3u;
    return 2u * (sps + J * LQ + J * 64u + (5u - ld) * 64u);
}
void build_wave_tables_d(uint32_t sf, uint32_t decim, const float2 *down, float *out) { build_wave_tables_d_host((int)sf, decim == 2u ? 1 : decim == 4u ? 2 : 3, down, out); }

// ---- symbol-level kernels: one wavefront per symbol (lora_hip_demod_symbols_device) -----------------------------------------------------------------------------
template <int SF, int LD>
__global__ __launch_bounds__(512, (WaveGeomD<SF, LD>::J <= 16 ? 4 : 2)) void demod_symbols_wave_d_kernel(DevParams P, const float2 *iq, const int64_t *offsets, uint32_t n, uint32_t *bins, int32_t *fine)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const WaveTabs T = wave_tabs_to_lds_d<SF, LD>(P, smem, 512u);
    __syncthreads();
    const uint32_t wave = threadIdx.x >> 6, lane = threadIdx.x & 63u;
    for (uint32_t s = blockIdx.x * 8u + wave; s < n; s += gridDim.x * 8u) {
        const int64_t o0 = offsets[s];
        uint32_t b;
        int32_t fs;
        wave_demod_symbol_d<SF, LD>(P, T, iq + o0, b, fs);
        if (b == kPoisonBin) wave_demod_symbol_d<SF, LD, true>(P, T, iq + o0, b, fs); // (uniform) a window with a sample of exactly zero
        if (lane == 0u) { bins[s] = b; if (fine) fine[s] = fs; }
    }
}

template <int SF, int LD>
__global__ __launch_bounds__(256) void demod_symbols_wave_grad_d_kernel(DevParams P, const float2 *iq, const int64_t *offsets, uint32_t n, uint32_t *bins, int32_t *fine)
{
    using G = WaveGeomD<SF, LD>;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    float *lds_v = reinterpret_cast<float *>(smem);
    for (uint32_t i = threadIdx.x; i < 3u * G::SPS + 40u; i += 256u) lds_v[i] = P.up_ifreq_v[i];
    __syncthreads();
    const uint32_t wave = threadIdx.x >> 6;
    for (uint32_t s = blockIdx.x * 4u + wave; s < n; s += gridDim.x * 4u) {
        uint32_t b;
        int32_t fs;
        float en;
        wave_demod_symbol_grad_d<SF, LD>(P, lds_v, iq + offsets[s], false, b, fs, en);
        if (b == kPoisonBin) wave_demod_symbol_grad_d<SF, LD, true>(P, lds_v, iq + offsets[s], false, b, fs, en);
        if ((threadIdx.x & 63u) == 0u) { bins[s] = b; if (fine) fine[s] = fs; }
    }
}

// the (SF, D) pairs as one switch: F<SF, LD>::go(...)
template <template <int, int> class F, typename... A>
static int wave_decim_dispatch(uint32_t sf, uint32_t decim, A... args)
{
    const uint32_t key = sf * 8u + decim;
    switch (key) {
    case 7u * 8u + 2u: return F<7, 1>::go(args...);
    case 8u * 8u + 2u: return F<8, 1>::go(args...);
    case 9u * 8u + 2u: return F<9, 1>::go(args...);
    case 7u * 8u + 4u: return F<7, 2>::go(args...);
    case 8u * 8u + 4u: return F<8, 2>::go(args...);
    case 9u * 8u + 4u: return F<9, 2>::go(args...);
    case 6u * 8u + 4u: return F<6, 2>::go(args...);
    case 6u * 8u + 8u: return F<6, 3>::go(args...);
    default: return -1;
    }
}
