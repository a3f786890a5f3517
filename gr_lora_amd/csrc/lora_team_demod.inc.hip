// lora_team_demod.inc.hip -- get_shift_fft (lib/decoder_impl.cc:430-464) + fine_sync (:300-338) for SF10 .. SF12 at decimation 8 by TEAMS of wavefronts
// (round 6).  Included by lora_kernels.hip behind lora_walker3.inc.hip.
//
// A window is sps = 4096 G samples, G = 2 / 4 / 8 (SF10 / 11 / 12): G chunks of 4096.  The sps-point DFT splits over the chunk index c first (decimation in
// frequency): with n = 4096 c + n' and k = kappa + G k',
//     X[kappa + G k'] = DFT_4096( z_kappa )[k'],      z_kappa[n'] = W_sps^{kappa n'} sum_c W_G^{kappa c} y[4096 c + n'],      y = x * downchirp,
// i.e. G independent problems of exactly the size wave_demod_symbol<9> solves in one wavefront's registers (64 values per lane, 8-point polyphase combine,
// signed sub-bin |k'| <= 256: its network and its tables are used as they are - the reference's fold tmp[N/2] += F[N/2] (:450) belongs to kappa = 0 only, the
// other wavefronts take it out of the one table entry again).  A team is G wavefronts, wavefront w of a team ends up with kappa = w:
//   1. it loads slice w (rows 64 / G w .. of 64) of EVERY chunk - 64 loads of 512 contiguous bytes - and takes fine_sync's sign tests on them (ffs_row:
//      lora_wave_demod.inc.hip explains the closed form; the team's winding counts and end-point angles are summed through LDS with the arg-max);
//   2. dechirp: D[4096 c + n'] = D[n'] g_c[lane] - the first 4096 entries of the reference's table (LDS, as at SF9) times a factor that depends on the chunk and
//      on n' mod 8 G only (the chirp's phase is quadratic); exact up to the table's own rounding of its float phase argument (~1e-4 rad at SF12 - the size of the
//      noise the FFT's own rounding adds to a bin);
//   3. one radix-G butterfly over c in registers, the outputs in the order kappa = w, w + 1, ... (the factor W_G^{w c} is folded into g_c), the twiddle
//      W_sps^{kappa n'} = W_{64 G}^{kappa row} W_sps^{kappa lane} (a uniform LDS entry times a per-lane one);
//   4. ONE exchange: the 64 / G values for kappa = w stay, the others go to their wavefronts through LDS mailboxes, eight values per wavefront and step,
//      double-buffered, one barrier per step (4 / 6 / 7 steps).  Wavefront kappa receives the slices in the order kappa, kappa + 1, ... - the window cyclically
//      rotated by a whole number of slices, which multiplies bin k1 of the 64-point in-lane FFT by a phase and leaves |X|^2 alone;
//   5. the SF9 network on the 64 registers; first maximum over the team in bin order (bin = kappa + G * sub-bin);
//   6. fine_sync: the closed form for the team's window; the three sums - every wavefront over the rows it loaded, from a second read - only when some team of the
//      workgroup needs them (the decision is uniform over the workgroup: the barriers are).  A sample of exactly zero: the NaN of the fast arctangent is
//      patched with the reference's form (ifreq_prod_z) inside the sums, no second evaluation.
// The barriers are among the wavefronts of a team only (team_barrier: arrivals counted in LDS; SF12's team is the workgroup): the 4 / 2 teams of an SF10 / SF11
// workgroup drift apart, and one's memory round trips are the other's arithmetic.

template <int SF> struct TeamGeom {
    static_assert(SF >= 10 && SF <= 12, "teams of 2 / 4 / 8 wavefronts");
    static constexpr int N = 1 << SF, SPS = 8 * N, GW = 1 << (SF - 9), V = 64 / GW, TPW = 8 / GW, M = 64 * GW;
    static constexpr int NS = 8 - V / 8; // exchange steps (eight values per wavefront and step)
    using G9 = WaveGeom<9>;
    static constexpr uint32_t n_wave = G9::n_ent, n_g = GW * 64, n_m = M, n_h = GW * 64, n_q = 64; // 8-byte entries: wave tables | g_c[lane] | W_M^t | W_sps^{kappa lane} | fold
    static constexpr uint32_t n_ent = n_wave + n_g + n_m + n_h + n_q;
    static constexpr uint32_t mail_entries = 2u * 8u * 8u * 64u; // [buffer][wavefront][value][lane]
    static constexpr uint32_t red_floats = 8u * 16u;
    static constexpr uint32_t lds_bytes = (n_ent + mail_entries) * 8u + red_floats * 4u + 16u * 4u; // (+ one barrier counter per team)
};

struct TeamLds {
    WaveTabs T;       // the SF9-sized network's tables (T.down: the first 4096 entries of this SF's downchirp; T.v unused)
    const v2f *g, *m, *h, *q;
    v2f *mail;
    float *red;
    uint32_t *ctr;    // [team] arrivals at the team's barriers, counted up for the kernel's lifetime
};

template <int SF>
__device__ __forceinline__ TeamLds team_carve(unsigned char *smem)
{
    using G = TeamGeom<SF>;
    using G9 = WaveGeom<9>;
    v2f *l2 = reinterpret_cast<v2f *>(smem);
    TeamLds L{};
    L.T.down = l2; L.T.twn = l2 + G9::n_down; L.T.tws = L.T.twn + G9::n_twn; L.T.xst = L.T.tws + G9::n_tws; L.T.v = nullptr;
    L.g = l2 + G::n_wave; L.m = L.g + G::n_g; L.h = L.m + G::n_m; L.q = L.h + G::n_h;
    L.mail = l2 + G::n_ent;
    L.red = reinterpret_cast<float *>(L.mail + G::mail_entries);
    L.ctr = reinterpret_cast<uint32_t *>(L.red + G::red_floats);
    return L;
}

// A barrier among the G wavefronts of ONE team (SF10 / SF11: 4 / 2 teams share a workgroup; with __syncthreads all eight wavefronts would move in lock step
// and every team would wait out every other team's memory round trips).  Arrivals are counted up in LDS for the kernel's lifetime; `seen` is what this
// wavefront has waited for so far.  The LDS unit serves a wavefront's operations in order, so the mailbox writes in front of the arrival are visible to whoever
// sees the count.  Every wavefront of a team calls it the same number of times (team-uniform control flow only).
template <int GW>
__device__ __forceinline__ void team_barrier(uint32_t *ctr, uint32_t &seen, int lane)
{
    if constexpr (GW == 8) { __syncthreads(); return; } // (the team is the workgroup)
    seen += (uint32_t)GW;
    if (lane == 0) (void)__hip_atomic_fetch_add(ctr, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
    uint32_t spins = 0u;
    while ((int32_t)(__hip_atomic_load(ctr, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP) - seen) < 0) {
        __builtin_amdgcn_s_sleep(1);
        if (++spins > (1u << 22)) break; // (a lost team mate must not hang the device: the results are then wrong, the kernel ends)
    }
}

// the G-point DFT over the chunk index, in place on z[c * V + jj] (outputs in natural order i = 0 .. G-1 at z[i * V + jj]); forward: W = e^{-2 pi i / G}
template <int GW, int V>
__device__ __forceinline__ void team_butterfly(v2f (&z)[64])
{
    constexpr float kR = 0.70710678118654752440f;
#pragma unroll
    for (int jj = 0; jj < V; jj++) {
        if constexpr (GW == 2) {
            const v2f a = z[jj], b = z[V + jj];
            z[jj] = a + b; z[V + jj] = a - b;
        } else if constexpr (GW == 4) {
            const v2f a = z[jj], b = z[V + jj], c = z[2 * V + jj], d = z[3 * V + jj];
            const v2f s0 = a + c, d0 = a - c, s1 = b + d, d1 = b - d;
            const v2f d1r = d1.yx * (v2f){1.0f, -1.0f}; // d1 * (-i)
            z[jj] = s0 + s1; z[2 * V + jj] = s0 - s1;
            z[V + jj] = d0 + d1r; z[3 * V + jj] = d0 - d1r;
        } else {
            v2f t[8];
#pragma unroll
            for (int c = 0; c < 8; c++) t[c] = z[c * V + jj];
            // radix-2 DIF, three stages, then the bit reversal by register naming
            v2f u[8];
#pragma unroll
            for (int c = 0; c < 4; c++) { u[c] = t[c] + t[c + 4]; u[c + 4] = t[c] - t[c + 4]; }
            u[5] = (v2f){(u[5].x + u[5].y) * kR, (u[5].y - u[5].x) * kR};  // * W_8^1 = (1 - i) / sqrt 2
            u[6] = u[6].yx * (v2f){1.0f, -1.0f};                             // * W_8^2 = -i
            u[7] = (v2f){(u[7].y - u[7].x) * kR, -(u[7].x + u[7].y) * kR}; // * W_8^3 = (-1 - i) / sqrt 2
            v2f w[8];
#pragma unroll
            for (int h = 0; h < 2; h++) {
                w[4 * h + 0] = u[4 * h + 0] + u[4 * h + 2]; w[4 * h + 2] = u[4 * h + 0] - u[4 * h + 2];
                w[4 * h + 1] = u[4 * h + 1] + u[4 * h + 3];
                const v2f dd = u[4 * h + 1] - u[4 * h + 3];
                w[4 * h + 3] = dd.yx * (v2f){1.0f, -1.0f};
            }
            // last stage; output index = bit reversal of the position: positions (0,1) -> X0, X4; (2,3) -> X2, X6; (4,5) -> X1, X5; (6,7) -> X3, X7
            z[0 * V + jj] = w[0] + w[1]; z[4 * V + jj] = w[0] - w[1];
            z[2 * V + jj] = w[2] + w[3]; z[6 * V + jj] = w[2] - w[3];
            z[1 * V + jj] = w[4] + w[5]; z[5 * V + jj] = w[4] - w[5];
            z[3 * V + jj] = w[6] + w[7]; z[7 * V + jj] = w[6] - w[7];
        }
    }
}

// the SF9-sized network of wave_demod_symbol<9> (lora_wave_demod.inc.hip steps 1b-4: in-lane 64-point DIF, 8-point DIF over lq, polyphase combine, reduce-scatter,
// arg-max) on values that are dechirped and combined already.  nofold: take the reference's fold out of its one table entry (kappa != 0).  Returns |X|^2 of the
// best sub-bin and the sub-bin (first maximum in bin order).
__device__ __forceinline__ void team_fft_core(v2f (&a)[64], const TeamLds &L, int lane, bool nofold, float &best_out, uint32_t &s_out)
{
    constexpr int J = 64, LOGJ = 6;
    const WaveTabs &T = L.T;
    const int lq = lane >> 3;
    fft_inlane_dif_pk<J>(a);
#pragma unroll
    for (int m = 1; m < J; m++) a[m] = cmul2(a[m], T.twn[m * 8 + lq]); // W_N^{lq k1}
    { // 8-point DIF over lq
        const v2f w1 = T.xst[lane], w2 = T.xst[64 + lane];
#pragma unroll
        for (int i = 0; i < J / 2; i++) { // lq bit 2 = lane bit 5
            const float dx = a[i].x, dy = a[i].y, sx = a[i + J / 2].x, sy = a[i + J / 2].y;
            const auto px = __builtin_amdgcn_permlane32_swap(__builtin_bit_cast(int, dx), __builtin_bit_cast(int, sx), false, false);
            const auto py = __builtin_amdgcn_permlane32_swap(__builtin_bit_cast(int, dy), __builtin_bit_cast(int, sy), false, false);
            const int x0 = px[0], x1 = px[1], y0 = py[0], y1 = py[1];
            const v2f lo = (v2f){__builtin_bit_cast(float, x0), __builtin_bit_cast(float, y0)};
            const v2f hi = (v2f){__builtin_bit_cast(float, x1), __builtin_bit_cast(float, y1)};
            a[i] = lo + hi;
            a[i + J / 2] = cmul2(lo - hi, w1);
        }
#pragma unroll
        for (int h = 0; h < 2; h++)
#pragma unroll
            for (int i = 0; i < J / 4; i++) { // lq bit 1 = lane bit 4
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
        const v2f sg3 = (lane & 8) ? (v2f){-1.f, -1.f} : (v2f){1.f, 1.f}; // lq bit 0 = lane bit 3
        xstage_last_pk<kDppRor8, J>(a, sg3);
    }
    {
        v2f w0 = T.tws[lane];
        if (nofold) w0 = w0 - L.q[lane]; // (uniform) tmp[N/2] += F[N/2] is bin N/2 of the whole window: kappa = 0, sub-bin 256
        a[0] = cmul2(a[0], w0);
    }
#pragma unroll
    for (int m = 1; m < J; m++) a[m] = cmul2(a[m], T.tws[m * 64 + lane]); // W_4096^{k' r} (+ fold)
    v2f b4[J / 2], b2[J / 4], b1[J / 8];
#pragma unroll
    for (int i = 0; i < J / 2; i++) {
        const float lx = a[i].x, ly = a[i].y, hx = a[i + J / 2].x, hy = a[i + J / 2].y;
        v2f X, Y;
        X.x = dpp_rows_bank_f<0x114, 0xA>(lx, hx); X.y = dpp_rows_bank_f<0x114, 0xA>(ly, hy);
        Y.x = dpp_rows_bank_f<0x104, 0x5>(hx, lx); Y.y = dpp_rows_bank_f<0x104, 0x5>(hy, ly);
        b4[i] = X + Y;
    }
    {
        const bool hi = (lane & 2) != 0;
#pragma unroll
        for (int i = 0; i < J / 4; i++) {
            const v2f t0 = b4[i] + dpp2<kDppQuadXor2>(b4[i]);
            const v2f t1 = b4[i + J / 4] + dpp2<kDppQuadXor2>(b4[i + J / 4]);
            b2[i] = hi ? t1 : t0;
        }
    }
    {
        const bool hi = (lane & 1) != 0;
#pragma unroll
        for (int i = 0; i < J / 8; i++) {
            const v2f t0 = b2[i] + dpp2<kDppQuadXor1>(b2[i]);
            const v2f t1 = b2[i + J / 8] + dpp2<kDppQuadXor1>(b2[i + J / 8]);
            b1[i] = hi ? t1 : t0;
        }
    }
    const int gbase = ((lane & 4) ? J / 2 : 0) + ((lane & 2) ? J / 4 : 0) + ((lane & 1) ? J / 8 : 0);
    float bv = -1.0f;
    int bi = 0x7fffffff;
#pragma unroll
    for (int i = 0; i < J / 8; i++) {
        const int jb = wave_layout_bin(J, LOGJ, gbase + i, lane);
        const float mag = b1[i].x * b1[i].x + b1[i].y * b1[i].y;
        if (mag > bv || (mag == bv && jb < bi)) { bv = mag; bi = jb; }
    }
    const float best = wave_max_nonneg_u(bv);
    s_out = (uint32_t)wave_min_u(bv == best ? bi : 0x7fffffff);
    best_out = best;
}

// One window per team: the team this wavefront belongs to demodulates the window at x (valid: team-uniform).  Called by all the wavefronts of the team the same
// number of times (team barriers inside); s_out = get_shift_fft's value, fine_out = d_fine_sync after fine_sync(bin_idx, 2), both uniform over the team.
template <int SF>
__device__ __forceinline__ void team_demod_window(const W3DemodArgs &P, const TeamLds &L, const float2 *__restrict__ x, bool valid, uint32_t &s_out, int32_t &fine_out,
                                                  uint32_t &seen /* team_barrier's count */)
{
    using G = TeamGeom<SF>;
    constexpr int N = G::N, SPS = G::SPS, GW = G::GW, V = G::V, M = G::M, CLS = kFfsClass<SF>;
    int tt = threadIdx.x;
    asm volatile("" : "+v"(tt)); // keeps per-thread table addresses out of the caller's loop-invariant set
    const int lane = tt & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tt >> 6), team = wave / GW, tw = wave % GW;
    const bool want_fine = P.enable_fine_sync != 0u;
    const bool ffs = want_fine && P.ffs_on != 0u;
    const auto xv = (const __attribute__((address_space(1))) v2f *)x;
    const int row0 = V * tw; // this wavefront's first row inside a chunk

    v2f z[64];
    float ffs_W = 0.0f, ffs_th = 0.0f;
    int ffs_zb = 0;
    if (valid) {
#pragma unroll
        for (int c = 0; c < GW; c++)
#pragma unroll
            for (int jj = 0; jj < V; jj++) z[c * V + jj] = xv[4096 * c + 64 * (row0 + jj) + lane];
        if (ffs) { // fine_sync's sign tests on this wavefront's 64 rows (row r = c V + jj <-> n = 4096 c + 64 (row0 + jj) + lane)
            uint32_t mA[2] = {0u, 0u}, mC[2] = {0u, 0u};
            float zmin = 3.0e38f;
#pragma unroll
            for (int r = 0; r < 64; r++) {
                float tq, re;
                const bool ends = CLS != 0 && (r == 0 || r == 63); // (the window's first and last four products: not held to the class bound, wave_demod_symbol)
                if (ends) {
                    ffs_row<CLS, false>(z[r].x, z[r].y, mA[r >> 5], mC[r >> 5], zmin, tq, re);
                    float u = CLS == 1 ? re : __builtin_fmaf(-2.0f, fabsf(tq), re);
                    u = (r == 0 ? (tw == 0 && lane < 4) : (tw == GW - 1 && lane >= 60)) ? 1.0f : u;
                    asm("v_min3_f32 %0, %1, |%2|, %3" : "=v"(zmin) : "v"(u), "v"(tq), "v"(zmin));
                } else {
                    ffs_row<CLS, true>(z[r].x, z[r].y, mA[r >> 5], mC[r >> 5], zmin, tq, re);
                }
            }
            zmin = lane == 0 ? 3.0e38f : zmin; // (lane 0's products were taken with lane 63's sample of its own row)
            int cnt = 0;
#pragma unroll
            for (int g = 0; g < 2; g++) {
                const uint32_t A = mA[g], Cm = mC[g];
                const uint32_t B = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)A, kDppWaveRor1, 0xf, 0xf, true);
                uint32_t q = ((Cm & B) | (~Cm & A)) & (A ^ B);
                q = lane == 0 ? 0u : q;
                cnt += __builtin_popcount(q) - 2 * __builtin_popcount(q & Cm);
            }
            { // the rows' first samples: lane l holds (x[n - 1], x[n]) of row l, n = 4096 (l / V) + 64 (row0 + l % V) (the window's n = 0 has no predecessor)
                const int nl = 4096 * (lane / V) + 64 * (row0 + lane % V);
                const bool mine = nl > 0;
                typedef float f4u __attribute__((ext_vector_type(4), aligned(8)));
                const f4u pb = *reinterpret_cast<const __attribute__((address_space(1))) f4u *>((const __attribute__((address_space(1))) float *)x + (mine ? 2 * nl - 2 : 0));
                const float tq = pb.w * pb.x - pb.z * pb.y, re = pb.z * pb.x + pb.w * pb.y;
                const uint32_t A = __builtin_bit_cast(uint32_t, pb.w), B = __builtin_bit_cast(uint32_t, pb.y), Cm = __builtin_bit_cast(uint32_t, tq);
                const uint32_t q = mine ? (((Cm & B) | (~Cm & A)) & (A ^ B)) : 0u;
                cnt += (int)(q >> 31) - 2 * (int)((q & Cm) >> 31);
                float u = fabsf(tq);
                if constexpr (CLS != 0) u = fminf(u, CLS == 1 ? re : __builtin_fmaf(-2.0f, u, re));
                zmin = mine ? fminf(zmin, u) : zmin;
            }
            ffs_W = wave_sum_u((float)cnt);
            ffs_zb = wave_min_u(__builtin_bit_cast(int, zmin));
            const v2f sel = (tw == 0 && lane == 0) ? z[0] : z[63]; // x[0] (first wavefront, lane 0) / x[sps-2], x[sps-1] (last wavefront, lanes 62, 63)
            ffs_th = lean_atan2_pk((v2f){sel.y, sel.y}, (v2f){sel.x, sel.x}).x;
        }
        // dechirp (:437): D[4096 c + n'] = D[n'] g_c[lane]; W_G^{tw c} rides on g_c (the butterfly then delivers kappa = tw, tw + 1, ...)
#pragma unroll
        for (int jj = 0; jj < V; jj++) {
            const v2f d = L.T.down[64 * (row0 + jj) + lane];
#pragma unroll
            for (int c = 0; c < GW; c++) z[c * V + jj] = cmul2(z[c * V + jj], d);
        }
#pragma unroll
        for (int c = 1; c < GW; c++) {
            const v2f gc = cmul2(L.g[c * 64 + lane], L.m[((tw * c) % GW) * 64]); // W_G^{tw c} = W_M^{64 (tw c mod G)}
#pragma unroll
            for (int jj = 0; jj < V; jj++) z[c * V + jj] = cmul2(z[c * V + jj], gc);
        }
        team_butterfly<GW, V>(z);
        // z[i V + jj] is now destined for kappa = (tw + i) mod G: the twiddle W_sps^{kappa n'}, n' = 64 (row0 + jj) + lane
#pragma unroll
        for (int i = 0; i < GW; i++) {
            const int kappa = (tw + i) % GW;
            if (kappa != 0) { // (uniform)
                const v2f hk = L.h[kappa * 64 + lane];
#pragma unroll
                for (int jj = 0; jj < V; jj++) {
                    const v2f w = cmul2(hk, L.m[(kappa * (row0 + jj)) % M]);
                    z[i * V + jj] = cmul2(z[i * V + jj], w);
                }
            }
        }
    } else {
#pragma unroll
        for (int r = 0; r < 64; r++) z[r] = (v2f){0.0f, 0.0f};
    }

    // ---- the exchange: eight values per wavefront and step; slot i of a wavefront goes to team mate (tw + i) mod G and arrives there as slot G - i ----
    {
        int step = 0;
        auto xchg = [&](int sb /* z[sb .. sb + 8): what this wavefront sends */, v2f (&rb)[8], int d) {
            v2f *mb = L.mail + (size_t)(step & 1) * (8 * 8 * 64);
            v2f *mine = mb + (size_t)wave * (8 * 64) + lane;
#pragma unroll
            for (int q = 0; q < 8; q++) mine[q * 64] = z[sb + q];
            team_barrier<GW>(L.ctr + team, seen, lane);
            const int from = team * GW + (tw + GW - d) % GW; // the team mate whose slot d is meant for me
            const v2f *theirs = mb + (size_t)from * (8 * 64) + lane;
#pragma unroll
            for (int q = 0; q < 8; q++) rb[q] = theirs[q * 64];
            step++;
        };
        // self-paired distance G / 2: what is sent from slot G / 2 comes back into slot G / 2
#pragma unroll
        for (int part = 0; part < V / 8; part++) {
            v2f rb[8];
            xchg((GW / 2) * V + 8 * part, rb, GW / 2);
#pragma unroll
            for (int q = 0; q < 8; q++) z[(GW / 2) * V + 8 * part + q] = rb[q];
        }
        // pairs (d, G - d): slot d goes out (what comes in is my slot G - d, still to be sent: parked), slot G - d goes out (what comes in is my slot d)
#pragma unroll
        for (int d = 1; d < GW / 2; d++)
#pragma unroll
            for (int part = 0; part < V / 8; part++) {
                v2f park[8], rb[8];
                xchg(d * V + 8 * part, park, d);
                xchg((GW - d) * V + 8 * part, rb, GW - d);
#pragma unroll
                for (int q = 0; q < 8; q++) { z[d * V + 8 * part + q] = rb[q]; z[(GW - d) * V + 8 * part + q] = park[q]; }
            }
    }

    // ---- the SF9-sized network, the team's arg-max and the closed form's sums ----
    float best;
    uint32_t sub;
    team_fft_core(z, L, lane, tw != 0, best, sub);
    float *red = L.red + wave * 16;
    if (lane == 0) {
        red[0] = valid ? best : -1.0f; ((int *)red)[1] = (int)(tw + GW * (int)sub);
        red[2] = ffs_W; ((int *)red)[3] = (ffs && valid) ? ffs_zb : 0;
    }
    if (ffs && valid) {
        if (tw == 0 && lane == 0) red[4] = ffs_th;
        if (tw == GW - 1 && lane >= 62) red[5 + (lane - 62)] = ffs_th;
    }
    team_barrier<GW>(L.ctr + team, seen, lane);
    const float *rg = L.red + team * GW * 16;
    float gv = rg[0], Wg = rg[2];
    int gi = ((const int *)rg)[1], zb = ((const int *)rg)[3];
#pragma unroll
    for (int w = 1; w < GW; w++) {
        const float ov = rg[w * 16];
        const int oi = ((const int *)rg)[w * 16 + 1];
        if (ov > gv || (ov == gv && oi < gi)) { gv = ov; gi = oi; }
        Wg += rg[w * 16 + 2];
        zb = min(zb, ((const int *)rg)[w * 16 + 3]);
    }
    s_out = (uint32_t)__builtin_amdgcn_readfirstlane(gi);
    fine_out = 0;
    if (!want_fine) return;
    // fine_sync (:300-338), lags -1, 0, +1: the closed form first (wave_demod_symbol FMODE 2 explains the rule)
    const uint32_t bin_idx = (s_out == 0u && P.demod_mode == 2u) ? 0u : (s_out + (uint32_t)N - 1u) % (uint32_t)N;
    if (!valid) return;
    if (ffs && zb > 0 && bin_idx != (uint32_t)N - 1u) { // (team-uniform)
        const float th0 = rg[4], th2 = rg[(GW - 1) * 16 + 5], the = rg[(GW - 1) * 16 + 6];
        float last = the - th2; // ifreq[sps-1] = ifreq[sps-2] (:243)
        last = last > 3.14159265358979324f ? last - 6.28318530717958648f : (last < -3.14159265358979324f ? last + 6.28318530717958648f : last);
        const float F = w3_uni((the - th0) + 6.28318530717958648f * Wg + last);
        const int ka = SPS - 8 * ((int)bin_idx + 1);
        const v2f xs = xv[ka - 1 + (lane < 2 ? lane : 2)];
        const float th = lean_atan2_pk((v2f){xs.y, xs.y}, (v2f){xs.x, xs.x}).x;
        float d = th - dpp_f<kDppWaveRor1>(th);
        d = d > 3.14159265358979324f ? d - 6.28318530717958648f : (d < -3.14159265358979324f ? d + 6.28318530717958648f : d);
        const float fb = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, d), 1)); // ifreq[ka - 1]
        const float fa = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, d), 2)); // ifreq[ka]
        const float D0 = P.ffs_alpha * F + P.ffs_jump * fa, D1 = P.ffs_alpha * F + P.ffs_jump * fb;
        if (w3_ub(D0 > P.ffs_tol && D1 < -P.ffs_tol)) return; // lag 0 whatever the signs of the sums (the same decision in every wavefront of the team: same inputs)
    }
    // the three sums themselves: every wavefront over the rows it loaded (a second, cache-hot read), f = ifreq[n - 1] against v[k - 1], v[k], v[k + 1], k = n - 1
    float c0 = 0.0f, c1 = 0.0f, c2 = 0.0f;
    {
        const auto vp = (const __attribute__((address_space(1))) float *)(P.up_ifreq_v + ((int)(bin_idx + 1u) * 8 + SPS)) - 2; // vp[n] = v[k - 1], k = n - 1
#pragma unroll 1
        for (int c = 0; c < GW; c++) {
#pragma unroll 4
            for (int jj = 0; jj < V; jj += 2) {
                const int n0 = 4096 * c + 64 * (row0 + jj) + lane, n1 = n0 + 64;
                const v2f p0 = xv[n0 >= 1 ? n0 - 1 : 0], q0 = xv[n0], p1 = xv[n1 - 1], q1 = xv[n1];
                v2f fp = ifreq_prod_pk(p0, q0, p1, q1);
                if (__builtin_amdgcn_ballot_w64(poisoned(fp.x) || poisoned(fp.y)) != 0ull) { // a sample of exactly zero: the reference's form for those values (rare)
                    if (poisoned(fp.x)) fp.x = ifreq_prod_z(make_float2(p0.x, p0.y), make_float2(q0.x, q0.y));
                    if (poisoned(fp.y)) fp.y = ifreq_prod_z(make_float2(p1.x, p1.y), make_float2(q1.x, q1.y));
                }
                const float f0 = n0 >= 1 ? fp.x : 0.0f, f1 = fp.y; // (n = 0 has no k)
                const int m0 = n0 >= 1 ? n0 : 2;
                c0 += f0 * vp[m0] + f1 * vp[n1]; c1 += f0 * vp[m0 + 1] + f1 * vp[n1 + 1]; c2 += f0 * vp[m0 + 2] + f1 * vp[n1 + 2];
                if (n1 == SPS - 1) { c0 += f1 * vp[n1 + 1]; c1 += f1 * vp[n1 + 2]; c2 += f1 * vp[n1 + 3]; } // ifreq[sps-1] = ifreq[sps-2] (:243): the duplicated tap
            }
        }
    }
    c0 = wave_sum_u(c0); c1 = wave_sum_u(c1); c2 = wave_sum_u(c2);
    if (lane == 0) { red[8] = c0; red[9] = c1; red[10] = c2; }
    team_barrier<GW>(L.ctr + team, seen, lane);
    float a0 = 0.0f, a1 = 0.0f, a2 = 0.0f;
#pragma unroll
    for (int w = 0; w < GW; w++) { a0 += rg[w * 16 + 8]; a1 += rg[w * 16 + 9]; a2 += rg[w * 16 + 10]; }
    float mx = 0.0f;
    int32_t lag = 0;
    if (a0 > mx) { mx = a0; lag = -1; }
    if (a1 > mx) { mx = a1; lag = 0; }
    if (a2 > mx) { mx = a2; lag = 1; }
    fine_out = __builtin_amdgcn_readfirstlane(-lag);
}

// ---- the symbol-level kernel: lora_hip_demod_symbols_device and the payload pass of a decoupled pass, SF10-SF12 FFT ------------------------------------
template <int SF>
__global__ __launch_bounds__(512, 2) void demod_symbols_team_kernel(DevParams P, const float2 *iq, const int64_t *offsets, uint32_t n, uint32_t *bins, int32_t *fine, DemodAlt alt)
{
    using G = TeamGeom<SF>;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const TeamLds L = team_carve<SF>(smem);
    {
        const v2f *__restrict__ src = reinterpret_cast<const v2f *>(P.team_tabs);
        v2f *dst = reinterpret_cast<v2f *>(smem);
        for (uint32_t i = threadIdx.x; i < G::n_ent; i += 512u) dst[i] = src[i];
        if (threadIdx.x < 16u) L.ctr[threadIdx.x] = 0u;
    }
    __syncthreads();
    const W3DemodArgs DA{P.down, P.w3_ctab, P.up_ifreq_v, P.enable_fine_sync, P.demod_mode, P.ffs_on, P.ffs_alpha, P.ffs_jump, P.ffs_tol};
    const int wave = __builtin_amdgcn_readfirstlane((int)threadIdx.x >> 6), team = wave / G::GW, tw = wave % G::GW;
    const bool writer = tw == 0 && (threadIdx.x & 63u) == 0u; // the team's results are uniform over the team: its first lane stores them
    uint32_t seen = 0u;
    for (uint32_t s0 = blockIdx.x * G::TPW; s0 < n; s0 += gridDim.x * G::TPW) { // (every team of the workgroup makes the same number of turns)
        const uint32_t s = s0 + (uint32_t)team;
        const bool valid = s < n;
        const int64_t o0 = offsets[valid ? s : s0];
        uint32_t b;
        int32_t fs;
        team_demod_window<SF>(DA, L, iq + o0, valid, b, fs, seen);
        if (writer && valid) { bins[s] = b; if (fine) fine[s] = fs; }
        if (alt.shift) { // second reads (DemodAlt): the successor of a symbol that moved the symbol clock, that far further on - the team that met the move reads it
            int32_t sh = 0;
            if (valid && fs != 0 && s + 1u < n) { // (team-uniform)
                const int64_t o1 = offsets[s + 1u], a = o1 + (int64_t)fs;
                if (o1 == o0 + (int64_t)G::SPS && a >= 0 && a <= alt.max_start) {
                    uint32_t b2;
                    int32_t f2;
                    team_demod_window<SF>(DA, L, iq + a, true, b2, f2, seen);
                    if (writer) { alt.bins[s + 1u] = b2; alt.fine[s + 1u] = f2; }
                    sh = fs;
                }
            }
            // DemodAlt.shift[s + 1] is written by the team that demodulated symbol s, taken or not (and shift[0] by the first one): the caller clears nothing
            if (writer && valid) { if (s + 1u < n) alt.shift[s + 1u] = sh; if (s == 0u) alt.shift[0] = 0; }
        }
    }
}

// host side: the table block [wave tables of the 4096-point sub-problem | g_c[lane] | W_M^t | W_sps^{kappa lane} | the fold's addend], 8-byte entries
template <int SF>
static void build_team_tables_sf(const float2 *down, double dt, double bandwidth, float2 *out)
{
    using G = TeamGeom<SF>;
    build_wave_tables_host(9u, down, reinterpret_cast<float *>(out)); // (its downchirp part: the first 4096 entries of THIS spreading factor's table)
    float2 *g = out + G::n_wave, *m = g + G::n_g, *h = m + G::n_m, *q = h + G::n_h;
    // the ideal chirp's phase (build_ideal_chirps :141-160, without the float rounding of its argument): pd(i) = 2 pi t (f0 + T t), t = dt i
    const double sym_rate = bandwidth / (double)G::N, T = -0.5 * bandwidth * sym_rate, f0 = bandwidth / 2.0;
    auto pd = [&](double i) { const double t = dt * i; return 2.0 * M_PI * t * (f0 + T * t); };
    for (int c = 0; c < G::GW; c++)
        for (int lane = 0; lane < 64; lane++) { // g_c[lane] = D[4096 c + lane] / D[lane]
            const double a = pd(4096.0 * c + lane) - pd((double)lane);
            g[c * 64 + lane] = make_float2((float)std::cos(a), (float)std::sin(a));
        }
    for (int t = 0; t < G::M; t++) { const double a = -2.0 * M_PI * (double)t / (double)G::M; m[t] = make_float2((float)std::cos(a), (float)std::sin(a)); }
    for (int k = 0; k < G::GW; k++)
        for (int lane = 0; lane < 64; lane++) { const double a = -2.0 * M_PI * (double)(k * lane) / (double)G::SPS; h[k * 64 + lane] = make_float2((float)std::cos(a), (float)std::sin(a)); }
    for (int lane = 0; lane < 64; lane++) { // what tmp[N/2] += F[N/2] (:450) added to the polyphase coefficient of sub-bin 256: register 0, lanes 8 .. 15
        q[lane] = make_float2(0.0f, 0.0f);
        if ((lane >> 3) == 1) {
            const int r = lane & 7;
            const double a2 = -2.0 * M_PI * (double)((256 * r) % 4096) / 4096.0;
            q[lane] = make_float2((float)std::cos(a2), (float)std::sin(a2));
        }
    }
}

uint32_t team_tables_entries(uint32_t sf) { return sf == 10u ? TeamGeom<10>::n_ent : sf == 11u ? TeamGeom<11>::n_ent : sf == 12u ? TeamGeom<12>::n_ent : 0u; }
void build_team_tables(uint32_t sf, const float2 *down, double dt, double bandwidth, float2 *out)
{
    if (sf == 10u) build_team_tables_sf<10>(down, dt, bandwidth, out);
    else if (sf == 11u) build_team_tables_sf<11>(down, dt, bandwidth, out);
    else if (sf == 12u) build_team_tables_sf<12>(down, dt, bandwidth, out);
}
static uint32_t team_lds_bytes(uint32_t sf) { return sf == 10u ? TeamGeom<10>::lds_bytes : sf == 11u ? TeamGeom<11>::lds_bytes : TeamGeom<12>::lds_bytes; }
