// lora_wave_demod.inc.hip -- one wavefront demodulates one symbol (decimation 8: SF7 / SF8 / SF9 at 1 Msps / 125 kHz).
// Included by lora_kernels.hip.
//
// get_shift_fft (lib/decoder_impl.cc:430-464) + the per-symbol fine_sync (:300-338, :514-518) for the
// decode rounds of walker2 and for lora_hip_demod_symbols_device.  Written for the VALU, which is what
// bounds the walker: complex arithmetic in packed fp32 (v_pk_fma_f32 / v_pk_mul_f32 / v_pk_add_f32, two
// flops per lane per instruction), lane exchanges as single DPP moves (bound_ctrl: no `old` operand to
// materialise) or v_permlane{16,32}_swap, every twiddle an 8-byte LDS entry (w.x, w.y) and a complex multiply
// two packed instructions: a.yy * (-w.y, w.x) through op_sel / neg_lo, then a.xx * (w.x, w.y) + that (cmul2).
// (Until round 5 the entries were 16 bytes - (w.x, w.y, -w.y, w.x) - for the same two instructions from vector
// code; at half the size the SF8 walker's workgroup fits a CU twice and SF9 fits at all: profiles/r05_ab_sf9_wave_fft.txt.)
//
// Work split.  sps = 8 N samples, lane = 8 lq + r.  A lane owns the samples n = 64 j + lane, j < J = N / 8:
// every load instruction covers 512 contiguous bytes (the earlier layout with the 8 lanes of a polyphase
// branch adjacent cost one L1 access per LANE and made the texture path, not the VALU, the bound).  In terms
// of the polyphase split n = 8 q + r, q = 8 j + lq: lane bits 0-2 are the branch r, bits 3-5 are lq.
//   1. dechirp, J-point DIF in registers (output k1 = bitrev(m))
//   2. twiddle W_N^{lq k1}, 8-point DIF over lq = lane bits 5, 4, 3.  Bits 5 and 4 use v_permlane{32,16}_swap
//      as a TRANSPOSE: swapping registers (i, i + J/2) leaves each lane with both inputs of one butterfly
//      (in_low, in_high), so sum and twiddled difference are computed in the lane, without sign selects;
//      afterwards the lane bit says which element the lane holds and the register slot carries the lq bit.
//      Bit 3 is a DPP row_ror:8 butterfly.  Result: Y_r[k1 + J k2] for the (register, lane) layout that
//      wave_layout_bin() below describes; the host builds the polyphase table from the same function.
//   3. polyphase combine with W_sps^{k r} (k = signed bin; the reference's fold tmp[N/2] += F[N/2], :450, is
//      part of the table) and a reduce-scatter over r = lane bits 2 (row_shl/shr:4 with bank masks), 1, 0
//   4. |X|^2 arg-max, first maximum in bin order (:454-463)
//   5. fine_sync over lags -1, 0, +1 against the ifreq template; the window's instantaneous frequency is
//      computed from the registers that were loaded for the dechirp (EARLY_F: every shipped instantiation - SF7, SF8, SF9) or
//      from a second, cache-hot read after the FFT (an SF8 build held to 128 registers, where 32 more live ones would spill).
//
// Instruction costs this is written against (tools/ubench_valu.hip, cycles per wave64 instruction per SIMD):
// v_fma/mul/add_f32 2.4-3.0, v_pk_{fma,mul,add}_f32 4.3, DPP moves and v_*_dpp 4.3, v_cndmask/v_cmp/v_max 4.3,
// v_rcp_f32 and v_permlane*_swap 8.2.

#ifndef LORA_W2_FFS
#define LORA_W2_FFS 7 // bit (SF - 7): fine_sync in closed form (wave_demod_symbol FMODE 2) - SF7, SF8, SF9
                      // oracle on the GPU (tests/test_gpu_ffs.py with the switch on: 221 passed) and measured: 16 % fewer VALU instructions per symbol, but the
                      // lane-mask bookkeeping is scalar work of the same wavefront - SF7 -4 %, SF8 -1 % (profiles/r03_ab_closed_form_fine_sync.txt): off
#endif
#ifndef LORA_W2_EARLY_F_SF8
#define LORA_W2_EARLY_F_SF8 1 // SF8: fine_sync's ifreq from the registers loaded for the dechirp (as SF7) instead of a second, cache-hot read:
                              // at a 256-register budget it fits without a spill (227 VGPRs) and measured +6.5 %; at the 128 registers of the two-per-CU walker (round 5) it spills and still wins
#endif

typedef float v2f __attribute__((ext_vector_type(2)));

constexpr int kDppQuadXor1 = 0xB1, kDppQuadXor2 = 0x4E, kDppHalfMirror = 0x141, kDppMirror = 0x140, kDppRor8 = 0x128,
              kDppBcast15 = 0x142, kDppBcast31 = 0x143,
              kDppWaveRor1 = 0x13C, // lane i reads lane i - 1, lane 0 reads lane 63 (GFX9 wavefront rotate)
              kDppWaveRol1 = 0x134, // lane i reads lane i + 1, lane 63 reads lane 0
              kDppRowHalfMirror = 0x141;

template <int CTRL>
__device__ __forceinline__ float dpp_f(float v)
{ // every lane reads a valid lane: bound_ctrl = true lets the compiler fold the move into the consumer
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xf, 0xf, true));
}
template <int CTRL>
__device__ __forceinline__ v2f dpp2(v2f v)
{
    const float x = v.x, y = v.y; // (bit_cast on a vector element reads element 0 with this compiler: go through scalars)
    v2f r;
    r.x = dpp_f<CTRL>(x);
    r.y = dpp_f<CTRL>(y);
    return r;
}
template <int CTRL, int ROWMASK>
__device__ __forceinline__ float dpp_rows_f(float old, float v)
{
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(__builtin_bit_cast(int, old), __builtin_bit_cast(int, v), CTRL, ROWMASK, 0xf, false));
}

template <int CTRL, int BANKMASK>
__device__ __forceinline__ float dpp_rows_bank_f(float old, float v)
{ // lanes of the enabled banks read `v` through the DPP pattern, the others keep `old`
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(__builtin_bit_cast(int, old), __builtin_bit_cast(int, v), CTRL, 0xf, BANKMASK, false));
}

// sums / maxima over the wavefront, result uniform (SGPR): 4 DPP steps inside the 16-lane rows, 2 row broadcasts
__device__ __forceinline__ float wave_sum_u(float v)
{
    v += dpp_f<kDppQuadXor1>(v); v += dpp_f<kDppQuadXor2>(v); v += dpp_f<kDppHalfMirror>(v); v += dpp_f<kDppMirror>(v);
    v += dpp_rows_f<kDppBcast15, 0xa>(0.0f, v);
    v += dpp_rows_f<kDppBcast31, 0xc>(0.0f, v);
    return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 63));
}
__device__ __forceinline__ float wave_max_nonneg_u(float v)
{
    v = fmaxf(v, dpp_f<kDppQuadXor1>(v)); v = fmaxf(v, dpp_f<kDppQuadXor2>(v));
    v = fmaxf(v, dpp_f<kDppHalfMirror>(v)); v = fmaxf(v, dpp_f<kDppMirror>(v));
    v = fmaxf(v, dpp_rows_f<kDppBcast15, 0xa>(0.0f, v));
    v = fmaxf(v, dpp_rows_f<kDppBcast31, 0xc>(0.0f, v));
    return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 63));
}
__device__ __forceinline__ int wave_min_u(int v)
{
#define LORA_MIN_STEP(CTRL) v = min(v, __builtin_amdgcn_update_dpp(0, v, CTRL, 0xf, 0xf, true))
    LORA_MIN_STEP(kDppQuadXor1); LORA_MIN_STEP(kDppQuadXor2); LORA_MIN_STEP(kDppHalfMirror); LORA_MIN_STEP(kDppMirror);
#undef LORA_MIN_STEP
    v = min(v, __builtin_amdgcn_update_dpp(0x7fffffff, v, kDppBcast15, 0xa, 0xf, false));
    v = min(v, __builtin_amdgcn_update_dpp(0x7fffffff, v, kDppBcast31, 0xc, 0xf, false));
    return __builtin_amdgcn_readlane(v, 63);
}

// a * w with the twiddle given as (w, wr = (-w.y, w.x)): two packed instructions
__device__ __forceinline__ v2f cmulw(v2f a, v2f w, v2f wr) { return __builtin_elementwise_fma(a.xx, w, a.yy * wr); }

// a * w in two packed instructions: t = a.yy * (-w.y, w.x) through op_sel / neg_lo, then a.xx * w + t (written out: from
// vector code the compiler builds (-w.y, w.x) with a v_xor and a v_mov first)
__device__ __forceinline__ v2f cmul2(v2f a, v2f w)
{
    v2f t, r;
    asm("v_pk_mul_f32 %0, %1, %2 op_sel:[1,1] op_sel_hi:[1,0] neg_lo:[0,1]" : "=v"(t) : "v"(a), "v"(w));
    asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel_hi:[0,1,1]" : "=v"(r) : "v"(a), "v"(w), "v"(t));
    return r;
}
// in-register radix-2 DIF, natural input order, bit-reversed output
template <int J>
__device__ __forceinline__ void fft_inlane_dif_pk(v2f (&a)[J])
{
#pragma unroll
    for (int h = J / 2; h >= 1; h >>= 1) {
#pragma unroll
        for (int b = 0; b < J / 2; b++) {
            const int off = b % h, blk = b / h;
            const int i0 = blk * 2 * h + off, i1 = i0 + h;
            const int tw = off * (J / 2 / h); // W_J^tw
            const v2f u = a[i0], v = a[i1];
            a[i0] = u + v;
            const v2f d = u - v;
            if (tw == 0) a[i1] = d;
            else if (tw == J / 4) a[i1] = d.yx * (v2f){1.0f, -1.0f}; // * (-i)
            else {
                const float c = kW64c[tw * (64 / J)], s = kW64s[tw * (64 / J)];
                a[i1] = cmulw(d, (v2f){c, s}, (v2f){-s, c});
            }
        }
    }
}

// the last DIF butterfly stage across lanes: a' = sg * a + partner   (upper lane: sg = +1; lower: sg = -1)
template <int CTRL, int J>
__device__ __forceinline__ void xstage_last_pk(v2f (&a)[J], v2f sg)
{
#pragma unroll
    for (int m = 0; m < J; m++) a[m] = __builtin_elementwise_fma(sg, a[m], dpp2<CTRL>(a[m]));
}

// LDS-resident tables of the wave demodulator (built by build_wave_tables below, copied in by the kernels)
struct WaveTabs {
    const v2f   *down;  // [sps]      d_downchirp[n]
    const v2f   *twn;   // [J][8]     W_N^{lq * bitrev(m)}
    const v2f   *tws;   // [J][64]    polyphase twiddle of register m on each lane (incl. the N/2 fold)
    const v2f   *xst;   // [2][64]    lane twiddles of the first two cross-lane stages
    const float *v;     // d_upchirp_ifreq_v (+ guard)
};

template <int SF> struct WaveGeom {
    static constexpr int N = 1 << SF, J = N / 8, SPS = 8 * N, LOGJ = ilog2(J);
    static constexpr uint32_t n_down = SPS, n_twn = J * 8, n_tws = J * 64, n_xst = 2 * 64;
    static constexpr uint32_t n_ent = n_down + n_twn + n_tws + n_xst; // 8-byte entries of the packed table block
};

// Bin held by register g of `lane` after the cross-lane FFT (and, with g the surviving register, after the
// reduce-scatter): stage 1 paired registers (i, i + J/2) -> slot s, stage 2 (i, i + J/4) -> slot t.
__device__ __host__ constexpr int wave_layout_bin(int J, int logj, int g, int lane)
{
    const int b5 = (lane >> 5) & 1, b4 = (lane >> 4) & 1, b3 = (lane >> 3) & 1;
    const int s = g / (J / 2), t = (g / (J / 4)) & 1, i = g % (J / 4);
    const int e = i + (J / 4) * b4 + (J / 2) * b5; // in-lane FFT output register the value descends from
    const int k1 = brev_bits(e, logj);
    const int k2 = 4 * b3 + 2 * t + s;             // bitrev3 of the position (s, t, b3) in the 8-point DIF
    return k1 + J * k2;
}

// atan2 of two points at once where it pays (the polynomial); see lean_atan2 for the accuracy statement
__device__ __forceinline__ v2f lean_atan2_pk(v2f y, v2f x)
{
    const float ax0 = fabsf(x.x), ay0 = fabsf(y.x), ax1 = fabsf(x.y), ay1 = fabsf(y.y);
    // atan2(0, 0) = NaN (0 * rcp(0)): a zero PRODUCT x[n] conj(x[n-1]) means a SAMPLE of exactly zero, next to which the reference's value is not the
    // product's argument at all (std::arg(0) = 0: lora_kernels.hip, ifreq_prod_z).  The NaN poisons the sums the value feeds; every consumer checks its
    // (uniform) sum once per window and has a poisoned window evaluated again, sample by sample, in a round of its own (kPoisonBin / ZM below).
    v2f a;
    // (written out: behind inline-asm producers the compiler canonicalises both fminf / fmaxf operands with a v_max x, x each)
    float mn0, mn1, mx0, mx1;
    asm("v_max_f32_e64 %0, |%1|, |%2|" : "=v"(mx0) : "v"(x.x), "v"(y.x));
    asm("v_max_f32_e64 %0, |%1|, |%2|" : "=v"(mx1) : "v"(x.y), "v"(y.y));
    asm("v_min_f32_e64 %0, |%1|, |%2|" : "=v"(mn0) : "v"(x.x), "v"(y.x));
    asm("v_min_f32_e64 %0, |%1|, |%2|" : "=v"(mn1) : "v"(x.y), "v"(y.y));
    a.x = mn0 * __builtin_amdgcn_rcpf(mx0);
    a.y = mn1 * __builtin_amdgcn_rcpf(mx1);
    const v2f s = a * a;
    v2f p = (v2f){-0.0040545230731368065f, -0.0040545230731368065f};
    p = __builtin_elementwise_fma(p, s, (v2f){0.02186279185116291f, 0.02186279185116291f});
    p = __builtin_elementwise_fma(p, s, (v2f){-0.0559120774269104f, -0.0559120774269104f});
    p = __builtin_elementwise_fma(p, s, (v2f){0.09642177820205688f, 0.09642177820205688f});
    p = __builtin_elementwise_fma(p, s, (v2f){-0.13908621668815613f, -0.13908621668815613f});
    p = __builtin_elementwise_fma(p, s, (v2f){0.19946564733982086f, 0.19946564733982086f});
    p = __builtin_elementwise_fma(p, s, (v2f){-0.33329859375953674f, -0.33329859375953674f});
    p = __builtin_elementwise_fma(p, s, (v2f){0.9999993443489075f, 0.9999993443489075f});
    const v2f r = a * p;
    float r0 = r.x, r1 = r.y;
    r0 = (ay0 > ax0) ? 1.57079632679489662f - r0 : r0;
    r1 = (ay1 > ax1) ? 1.57079632679489662f - r1 : r1;
    r0 = (x.x < 0.0f) ? 3.14159265358979324f - r0 : r0;
    r1 = (x.y < 0.0f) ? 3.14159265358979324f - r1 : r1;
    return (v2f){copysignf(r0, y.x), copysignf(r1, y.y)};
}

// instantaneous frequency (:231-240) of two adjacent-sample pairs: arg(cur * conj(prev)).  The products are written out
// (cur * conj(prev) = cur.xx * (prev.x, -prev.y) + cur.yy * (prev.y, prev.x): two packed instructions on the registers
// the samples are in; from vector code the compiler first gathers (c0.y, c1.y), (p0.x, p1.x), ... with four v_mov per pair)
__device__ __forceinline__ v2f cmul_conj(v2f c, v2f p)
{
    v2f t, z;
    asm("v_pk_mul_f32 %0, %1, %2 op_sel:[1,1] op_sel_hi:[1,0]" : "=v"(t) : "v"(c), "v"(p));
    asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel_hi:[0,1,1] neg_hi:[0,1,0]" : "=v"(z) : "v"(c), "v"(p), "v"(t));
    return z; // (re, im)
}
__device__ __forceinline__ v2f ifreq_prod_pk(v2f p0, v2f c0, v2f p1, v2f c1)
{
    const v2f z0 = cmul_conj(c0, p0), z1 = cmul_conj(c1, p1);
    const float y0 = z0.y, y1 = z1.y, x0 = z0.x, x1 = z1.x;
    return lean_atan2_pk((v2f){y0, y1}, (v2f){x0, x1});
}

// ... and with the reference's convention next to a sample that is exactly zero (ZM paths; ifreq_prod_z is out of line)
__device__ __forceinline__ v2f ifreq_prod_pk_z(v2f p0, v2f c0, v2f p1, v2f c1)
{
    return (v2f){ifreq_prod_z(make_float2(p0.x, p0.y), make_float2(c0.x, c0.y)), ifreq_prod_z(make_float2(p1.x, p1.y), make_float2(c1.x, c1.y))};
}
__device__ __forceinline__ bool poisoned3(float a, float b, float c) { return poisoned(a + b + c); }
// what a demodulator returns in place of the bin for a POISONED window (a sample of exactly zero: its fine_sync sums are NaN): the caller has the
// window evaluated again by the ZM = true instantiation, which forms every ifreq value as the reference does (walkers: a round of its own)
constexpr uint32_t kPoisonBin = 0xfffffffeu;

// One row of the closed form's bookkeeping (wave_demod_symbol, FMODE 2) for the sample (ax, ay) = x[n] of this lane, n = 64 j + lane:
//   z = x[n] conj x[n-1] (x[n-1]: the neighbouring lane's sample as a DPP operand; lane 0 reads lane 63: the caller drops its bits),
//   mA <- sign Im x[n], mC <- sign Im z, zmin <- min(zmin, |Im z| [, Re z | Re z - 2 |Im z|]) (MIN; CLS 1 | 2: the class bound on |arg z|, kFfsClass).
// Written out: left to the compiler the scalar products are re-packed into v_pk_fma_f32 with a v_mov_b32_dpp per operand and register pairs to
// assemble, and every row's products are formed ahead of the serial mask chains (277 registers spilled at SF9).  The s_nop covers the
// VALU -> DPP read hazard should the compiler have moved (ax, ay) with a VALU instruction directly in front.
template <int CLS, bool MIN>
__device__ __forceinline__ void ffs_row(float ax, float ay, uint32_t &mA, uint32_t &mC, float &zmin, float &t, float &re)
{
#define LORA_FFS_DPP " wave_ror:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
    if constexpr (CLS == 0) { // Im z only
        asm("v_alignbit_b32 %0, %0, %5, 31\n\t"
            "s_nop 0\n\t"
            "v_mul_f32_dpp %3, %4, %5" LORA_FFS_DPP
            "v_fmac_f32_dpp %3, %5, -%4" LORA_FFS_DPP
            "v_alignbit_b32 %1, %1, %3, 31\n\t"
            "v_min_f32_e64 %2, |%3|, %2"
            : "+v"(mA), "+v"(mC), "+v"(zmin), "=&v"(t) : "v"(ax), "v"(ay));
        re = 0.0f;
    } else {
        float u;
#define LORA_FFS_HEAD                                 \
    "v_alignbit_b32 %0, %0, %7, 31\n\t"               \
    "s_nop 0\n\t"                                     \
    "v_mul_f32_dpp %3, %6, %7" LORA_FFS_DPP           \
    "v_mul_f32_dpp %4, %6, %6" LORA_FFS_DPP           \
    "v_fmac_f32_dpp %3, %7, -%6" LORA_FFS_DPP         \
    "v_fmac_f32_dpp %4, %7, %7" LORA_FFS_DPP          \
    "v_alignbit_b32 %1, %1, %3, 31"
#define LORA_FFS_OPS : "+v"(mA), "+v"(mC), "+v"(zmin), "=&v"(t), "=&v"(re), "=&v"(u) : "v"(ax), "v"(ay)
        if constexpr (!MIN) asm(LORA_FFS_HEAD LORA_FFS_OPS);
        else if constexpr (CLS == 1) asm(LORA_FFS_HEAD "\n\tv_min3_f32 %2, %4, |%3|, %2" LORA_FFS_OPS);
        else asm(LORA_FFS_HEAD "\n\tv_fma_f32 %5, -2.0, |%3|, %4\n\tv_min3_f32 %2, %5, |%3|, %2" LORA_FFS_OPS);
#undef LORA_FFS_HEAD
#undef LORA_FFS_OPS
    }
#undef LORA_FFS_DPP
}

// Demodulates the symbol window x[0 .. sps): s_out = get_shift_fft's return value, fine_out = d_fine_sync after
// fine_sync(bin_idx, 2) (0 when drift correction is disabled).  Must be called by a whole wavefront.
// FMODE: where fine_sync's instantaneous frequency comes from - 0: a second, cache-hot read of the window behind the FFT; 1: the registers
// loaded for the dechirp (16 / 32 / 64 values per lane kept through the FFT); 2: not computed at all for most windows - the CLOSED FORM (below),
// with mode 0 as the fall-back when the closed form cannot vouch for the reference's decision.
//
// The closed form (round 6: sign tests per lane; rounds 3-5 had built it on wavefront ballots and scalar popcounts, which cost more than it saved).
// fine_sync picks the first maximum > 0 of c(i) = sum_k ifreq[k] v[o + i + k], i = -1, 0, 1 (:306-313; o = shift_ref + sps, ifreq[sps-1] =
// ifreq[sps-2] :243).  v = d_upchirp_ifreq_v is a ramp of slope alpha with one step J at index 2 sps - 1 inside every stretch fine_sync can look
// at for bin_idx < N - 1 (DevParams::ffs_*, checked on the table itself at lora_hip_create), so
//     c(i+1) - c(i) = alpha F + J ifreq[2 sps - 1 - (o + i)] + (table noise, |.| <= ffs_tol),    F = sum_k ifreq[k],
// and F telescopes: ifreq[k] = theta[k+1] - theta[k] + 2 pi w[k] (:231-240), w[k] = -1 / +1 when the step crosses the negative real axis
// upwards / downwards - read off the signs of Im x[k], Im x[k+1] and Im(x[k+1] conj x[k]).  Per sample that is one product (two multiplies with
// the neighbour lane's sample as a DPP operand), two v_alignbit that collect the sign bits of a lane's rows into 32-bit words and one v_min; the
// crossings of 32 rows are then five bitwise operations and two population counts.  arg x[0], arg x[sps-2], arg x[sps-1] come from the registers
// the window was loaded into, the two ifreq values next to the template's step from three samples read again behind the arg-max.
// The decision taken here is the common one only: c(0) > c(-1) and c(0) >= c(1), both by more than the noise bound - the scan then ends on lag 0
// whatever the signs of the sums are (if c(-1) <= 0 the scan never leaves lag 0 before c(0); c(1) <= c(0) cannot displace it).  A symbol that moves
// the clock (~1 %), a window next to the table's tail (bin_idx = N - 1), a product of exactly zero (std::arg(0): see kPoisonBin; or two collinear
// samples, a tie of the sign tests), differences inside the noise bound and - SF9 and up, where the noise bound is computed for bounded ifreq - a window
// with an |ifreq[k]| above atan(1/2) away from its ends take the sums themselves (the exact path).  tools/ffs_model.py holds
// the same rule in numpy against the oracle's fine_sync (SF7 .. SF12: clean, noisy down to -15 dB, interferers, carrier offsets, partial
// windows); tests/test_gpu_ffs.py holds the kernel to the oracle.
// the bound on |ifreq[k]| a window has to keep (away from its ends) for the closed form to vouch for it - what DevParams::ffs_tol is computed for at
// lora_hip_create (ffs_class_bound): 0: none (pi), 1: pi / 2 (Re > 0), 2: atan(1/2) (Re > 2 |Im|)
template <int SF> constexpr int kFfsClass = SF <= 8 ? 0 : SF <= 10 ? 1 : 2;
template <int SF> constexpr int kWaveFmode = ((LORA_W2_FFS >> (SF - 7)) & 1) ? 2 : ((SF == 7) || LORA_W2_EARLY_F_SF8) ? 1 : 0;
template <int SF, int FMODE_, bool ZM = false>
__device__ __forceinline__ void wave_demod_symbol(const DevParams &P, const WaveTabs &T, const float2 *__restrict__ x, uint32_t &s_out, int32_t &fine_out,
                                                  float *en_out = nullptr /* implicit header: the window's energy (determine_energy, :368-375) */,
                                                  long long *stamps = nullptr /* tools/probe_phases.hip */)
{
    constexpr int FMODE = ZM ? 3 : FMODE_; // ZM: fine_sync's ifreq sample by sample with std::arg(0) = 0, in a rolled loop behind the arg-max (mode 3)
    constexpr bool EARLY_F = FMODE == 1;
#define LORA_WSTAMP(i) do { if (stamps) stamps[i] = clock64(); } while (0)
    LORA_WSTAMP(0);
    using G = WaveGeom<SF>;
    constexpr int N = G::N, J = G::J, SPS = G::SPS, LOGJ = G::LOGJ;
    int lane = threadIdx.x & 63;
    // opaque to the optimiser: keeps the per-lane table addresses from being hoisted out of the caller's
    // state-machine loop (that costs ~100 VGPRs of loop-invariant addresses)
    asm volatile("" : "+v"(lane));
    const int lq = lane >> 3;
    const int nl = lane;
    const bool want_fine = P.enable_fine_sync != 0u;
    const v2f *__restrict__ xv = reinterpret_cast<const v2f *>(x);

    v2f a[J];
    float f[J]; // ifreq[n - 1] of this lane's samples
#pragma unroll
    for (int j = 0; j < J; j++) a[j] = xv[j * 64 + nl];
    if (en_out) { // (a uniform branch: only implicit-header decoders ask)
        v2f e2 = (v2f){0.0f, 0.0f};
#pragma unroll
        for (int j = 0; j < J; j++) e2 = __builtin_elementwise_fma(a[j], a[j], e2);
        *en_out = wave_sum_u(e2.x + e2.y);
    }
    if (EARLY_F && want_fine) {
        // sample n - 1 of this lane's n = 64 j + lane sits in the neighbouring lane (lane 0: lane 63 of the previous
        // register): one wave rotate per register instead of a second, dependent round of loads.  (A second batch of
        // loads one item down saves 8 VALU slots per sample pair and was measured 8 % SLOWER in the walker: the round's
        // first-touch loads queue behind twice as many requests.)
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
    // FMODE 2: the window's winding number and end-point angles, from sign tests - no arctangent per sample (see "The closed form" above)
    float ffs_F = 0.0f;   // F = sum_k ifreq[k] (uniform)
    bool ffs_ok = false;  // (uniform) the closed form may vouch for this window: no product of exactly zero (a zero SAMPLE: std::arg(0), or two collinear
                          // samples: a tie of the sign tests) and - SF9 and up - every |ifreq[k]| inside the bound ffs_tol is computed for
    if constexpr (FMODE == 2) {
        if (want_fine && P.ffs_on != 0u) {
            constexpr int CLS = kFfsClass<SF>;
            constexpr int NM = (J + 31) / 32;
            uint32_t mA[NM], mC[NM]; // one bit per row j of this lane's samples n = 64 j + lane: Im x[n] < 0, Im(x[n] conj x[n-1]) < 0
#pragma unroll
            for (int g = 0; g < NM; g++) { mA[g] = 0u; mC[g] = 0u; }
            float zmin = 3.0e38f;
#pragma unroll
            for (int j = 0; j < J; j++) {
                // a window one or two samples off its symbol holds the phase step between two symbols next to one of its ends: the first and the last four
                // products are not held to the class (ffs_tol allows for eight values up to pi), only to being non-zero
                const bool ends = CLS != 0 && (j == 0 || j == J - 1);
                float t, re;
                if (ends) {
                    ffs_row<CLS, false>(a[j].x, a[j].y, mA[j >> 5], mC[j >> 5], zmin, t, re);
                    float u = CLS == 1 ? re : __builtin_fmaf(-2.0f, fabsf(t), re);
                    u = (j == 0 ? lane < 4 : lane >= 60) ? 1.0f : u;
                    asm("v_min3_f32 %0, %1, |%2|, %3" : "=v"(zmin) : "v"(u), "v"(t), "v"(zmin));
                } else {
                    ffs_row<CLS, true>(a[j].x, a[j].y, mA[j >> 5], mC[j >> 5], zmin, t, re);
                }
            }
            zmin = lane == 0 ? 3.0e38f : zmin; // (lane 0's products were taken with the wrong neighbour)
            int cnt = 0;
#pragma unroll
            for (int g = 0; g < NM; g++) {
                const uint32_t A = mA[g], Cm = mC[g];
                const uint32_t B = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)A, kDppWaveRor1, 0xf, 0xf, true); // Im x[n-1] < 0: the neighbour's bit of the same row
                uint32_t q = ((Cm & B) | (~Cm & A)) & (A ^ B); // the step crosses the real axis ... the long way round: the negative half
                q = lane == 0 ? 0u : q;
                cnt += __builtin_popcount(q) - 2 * __builtin_popcount(q & Cm); // upwards (+1: Cm clear) / downwards (-1)
            }
            { // the row boundaries: lane l, 1 <= l < J, holds (x[64 l - 1], x[64 l])
                const bool mine = lane >= 1 && lane < J;
                typedef float f4u __attribute__((ext_vector_type(4), aligned(8))); // (8-byte aligned: x[64 l - 1] is the last item of a 16-byte pair)
                const f4u pb = *reinterpret_cast<const __attribute__((address_space(1))) f4u *>((const __attribute__((address_space(1))) float *)x + (mine ? 128 * lane - 2 : 0));
                const float t = pb.w * pb.x - pb.z * pb.y, re = pb.z * pb.x + pb.w * pb.y; // x[n] conj x[n-1]
                const uint32_t A = __builtin_bit_cast(uint32_t, pb.w), B = __builtin_bit_cast(uint32_t, pb.y), Cm = __builtin_bit_cast(uint32_t, t);
                const uint32_t q = mine ? (((Cm & B) | (~Cm & A)) & (A ^ B)) : 0u;
                cnt += (int)(q >> 31) - 2 * (int)((q & Cm) >> 31);
                float u = fabsf(t);
                if constexpr (CLS != 0) u = fminf(u, CLS == 1 ? re : __builtin_fmaf(-2.0f, u, re));
                zmin = mine ? fminf(zmin, u) : zmin;
            }
            const float W = wave_sum_u((float)cnt); // (exact: |W| <= sps)
            const int zbits = wave_min_u(__builtin_bit_cast(int, zmin)); // (as integers: a negative value - out of the class - is smaller than any positive one)
            // arg x[0] (lane 0), arg x[sps-2], arg x[sps-1] (lanes 62, 63): one evaluation; ifreq[sps-1] = ifreq[sps-2] (:243) is the sum's last term
            const v2f sel = (lane == 0) ? a[0] : a[J - 1];
            const float th = lean_atan2_pk((v2f){sel.y, sel.y}, (v2f){sel.x, sel.x}).x;
            const float th0 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, th), 0));
            const float th2 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, th), 62));
            const float the = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, th), 63));
            float last = the - th2;
            last = last > 3.14159265358979324f ? last - 6.28318530717958648f : (last < -3.14159265358979324f ? last + 6.28318530717958648f : last);
            ffs_F = (the - th0) + 6.28318530717958648f * W + last;
            ffs_ok = zbits > 0 && ffs_F == ffs_F;
        }
    }
    LORA_WSTAMP(1);
#pragma unroll
    for (int j = 0; j < J; j++) a[j] = cmul2(a[j], T.down[j * 64 + nl]); // dechirp (:437)
    fft_inlane_dif_pk<J>(a);
    LORA_WSTAMP(2);
#pragma unroll
    for (int m = 1; m < J; m++) a[m] = cmul2(a[m], T.twn[m * 8 + lq]); // W_N^{lq k1}
    LORA_WSTAMP(3);
    { // 8-point DIF over lq
        const v2f w1 = T.xst[lane], w2 = T.xst[64 + lane]; // W_8^{lq & 3}, W_4^{lq & 1}: the same on both lanes of a pair
#pragma unroll
        for (int i = 0; i < J / 2; i++) { // lq bit 2 = lane bit 5
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
    LORA_WSTAMP(4);
#pragma unroll
    for (int m = 0; m < J; m++) a[m] = cmul2(a[m], T.tws[m * 64 + lane]); // W_sps^{k r} (+ fold)
    // reduce-scatter over r = lane bits 2, 1, 0: lanes with the bit clear keep the first half of the registers
    // (Measured and dropped: the same steps as v_add_f32 with a DPP operand written out in asm - fewer issue slots on paper,
    // 13 % slower in the walker: the volatile sequence no longer interleaves with the table loads around it.)
    v2f b4[J / 2], b2[J / 4], b1[J / 8];
#pragma unroll
    for (int i = 0; i < J / 2; i++) { // lane bit 2: row_shr:4 into banks 1,3 / row_shl:4 into banks 0,2
        const float lx = a[i].x, ly = a[i].y, hx = a[i + J / 2].x, hy = a[i + J / 2].y;
        v2f X, Y;
        X.x = dpp_rows_bank_f<0x114, 0xA>(lx, hx); X.y = dpp_rows_bank_f<0x114, 0xA>(ly, hy); // upper lanes: partner's second half
        Y.x = dpp_rows_bank_f<0x104, 0x5>(hx, lx); Y.y = dpp_rows_bank_f<0x104, 0x5>(hy, ly); // lower lanes: partner's first half
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
    // late ifreq: second read of the window, pinned behind the reduce-scatter so that the loads are not
    // hoisted above the FFT (where they would hold 4 J more registers)
    if (FMODE == 0 && want_fine) {
        int zero = 0;
        asm volatile("; fine-sync reload after the reduce-scatter" : "+v"(zero) : "v"(b1[0].x));
        const int nl2 = nl + zero;
#pragma unroll
        for (int j = 0; j < J; j += 2) {
            const int n0 = j * 64 + nl2, n1 = n0 + 64;
            const v2f fp = ifreq_prod_pk(xv[n0 >= 1 ? n0 - 1 : 0], xv[n0], xv[n1 - 1], xv[n1]);
            f[j] = (n0 >= 1) ? fp.x : 0.0f;
            f[j + 1] = fp.y;
        }
    }
    LORA_WSTAMP(5);
    // arg-max on |X|^2 (monotone in the reference's std::abs, :454), first maximum in bin order wins (:463)
    const int gbase = ((lane & 4) ? J / 2 : 0) + ((lane & 2) ? J / 4 : 0) + ((lane & 1) ? J / 8 : 0); // surviving registers
    float bv = -1.0f;
    int bi = 0x7fffffff;
#pragma unroll
    for (int i = 0; i < J / 8; i++) {
        const int jb = wave_layout_bin(J, LOGJ, gbase + i, lane);
        const float mag = b1[i].x * b1[i].x + b1[i].y * b1[i].y;
        if (mag > bv || (mag == bv && jb < bi)) { bv = mag; bi = jb; }
    }
    const float best = wave_max_nonneg_u(bv);
    const uint32_t s = (uint32_t)wave_min_u(bv == best ? bi : 0x7fffffff);
    s_out = s;
    fine_out = 0;
    LORA_WSTAMP(6);
    if (!want_fine) return;
    // fine_sync (:300-338) with search = max(D/4, 2) = 2 -> lags -1, 0, +1
    const uint32_t bin_idx = (s == 0u && P.demod_mode == 2u) ? 0u : (s + (uint32_t)N - 1u) % (uint32_t)N;
    const float *__restrict__ v = T.v + ((int)(bin_idx + 1u) * 8 + SPS);
    if constexpr (FMODE == 2) {
        if (ffs_ok && bin_idx != (uint32_t)N - 1u) { // (uniform)
            // ifreq[ka - 1], ifreq[ka] next to the template's step, ka = sps - 8 (bin_idx + 1) >= 8: three samples read again (cache-hot), formed as the
            // reference forms them - the difference of two sample arguments, unwrapped (:231-240).  (Taking the two decisions as flags while the window is
            // in registers - nothing read again - costs four more instructions per sample and measured the same in the walkers, 14 % slower standalone at SF8.)
            const int ka = SPS - 8 * ((int)bin_idx + 1);
            const v2f xs = xv[ka - 1 + (lane < 2 ? lane : 2)];
            const float th = lean_atan2_pk((v2f){xs.y, xs.y}, (v2f){xs.x, xs.x}).x;
            float d = th - dpp_f<kDppWaveRor1>(th);
            d = d > 3.14159265358979324f ? d - 6.28318530717958648f : (d < -3.14159265358979324f ? d + 6.28318530717958648f : d);
            const float fb = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, d), 1)); // ifreq[ka - 1]
            const float fa = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, d), 2)); // ifreq[ka]
            const float D0 = P.ffs_alpha * ffs_F + P.ffs_jump * fa, D1 = P.ffs_alpha * ffs_F + P.ffs_jump * fb; // c(0) - c(-1), c(1) - c(0)
            // c(0) above c(-1) and c(1) below c(0), both by more than the table's noise: the scan (:306-313) ends on lag 0 whatever the signs of the sums
            // are.  Everything else - the ~1 % of symbols that move the clock, and the windows whose differences lie inside the noise - is decided by the sums
            if (D0 > P.ffs_tol && D1 < -P.ffs_tol) { // (uniform; NaN: not taken)
                LORA_WSTAMP(7);
                return;
            }
        }
        // the exact path: the window's ifreq from a second, cache-hot read
#pragma unroll
        for (int j = 0; j < J; j += 2) {
            const int n0 = j * 64 + nl, n1 = n0 + 64;
            const v2f fp = ifreq_prod_pk(xv[n0 >= 1 ? n0 - 1 : 0], xv[n0], xv[n1 - 1], xv[n1]);
            f[j] = (n0 >= 1) ? fp.x : 0.0f;
            f[j + 1] = fp.y;
        }
    }
    // sample n = 64 j + lane carries ifreq[k], k = n - 1, against v[k - 1], v[k], v[k + 1]: one lane-dependent base and
    // immediate offsets.  (Lane 0, j = 0 has no k: its f is 0 and it reads the finite table entries in front of v.)
    const float *__restrict__ vp = v + (nl - 2);
    float c0 = 0.f, c1 = 0.f, c2 = 0.f;
    if constexpr (ZM) { // the window again from memory, every ifreq value as the reference forms it (ifreq_prod_z); rolled: this path is rare
#pragma unroll 1
        for (int j = 0; j < J; j++) {
            const int n = 64 * j + nl;
            const float fj = n >= 1 ? ifreq_prod_z(x[n - 1], x[n]) : 0.0f; // ifreq[n - 1]
            c0 += fj * vp[64 * j]; c1 += fj * vp[64 * j + 1]; c2 += fj * vp[64 * j + 2];
            if (j == J - 1 && nl == 63) { c0 += fj * vp[64 * j + 1]; c1 += fj * vp[64 * j + 2]; c2 += fj * vp[64 * j + 3]; } // ifreq[sps-1] = ifreq[sps-2] (:243)
        }
    } else {
#pragma unroll
    for (int j = 0; j < J; j++) {
        const float fj = f[j];
        c0 += fj * vp[64 * j]; c1 += fj * vp[64 * j + 1]; c2 += fj * vp[64 * j + 2];
        if (j == J - 1) { // ifreq[sps-1] = ifreq[sps-2] (:243): the lane that owns n = sps-1 adds the duplicated tap
            const float fl = (nl == 63) ? fj : 0.0f;
            c0 += fl * vp[64 * j + 1]; c1 += fl * vp[64 * j + 2]; c2 += fl * vp[64 * j + 3];
        }
    }
    }
    c0 = wave_sum_u(c0); c1 = wave_sum_u(c1); c2 = wave_sum_u(c2);
    if (!ZM && poisoned3(c0, c1, c2)) { s_out = kPoisonBin; return; } // (uniform) a sample of the window is exactly zero
    float mx = 0.0f;
    int32_t lag = 0;
    if (c0 > mx) { mx = c0; lag = -1; }
    if (c1 > mx) { mx = c1; lag = 0; }
    if (c2 > mx) { mx = c2; lag = 1; }
    fine_out = -lag;
    LORA_WSTAMP(7);
#undef LORA_WSTAMP
}

// ---- the reference's SHIPPED demodulator on one wavefront: max_frequency_gradient_idx (:466-491) + fine_sync (:300-338) ----
// No FFT at all: the window's instantaneous frequency (which fine_sync needs anyway) is averaged over the D = 8 samples of
// every bin (volk_32f_accumulator_s32f, :475-476) and the largest drop between neighbouring averages above 0.1 marks the
// symbol boundary (:479-488).  Lane l owns n = 64 j + l as above; here f[j] = ifreq[n] = arg(x[n+1] conj(x[n])) - the successor
// by a wave rotate, lane 63 from the next register - so that the eight samples of bin i = 8 j + (l >> 3) sit in one aligned
// group of eight lanes of register j: the bin average is three v_add with a DPP operand, its left neighbour one lane permute.
// ifreq[sps-1] = ifreq[sps-2] (:243).  bin_out is demodulate()'s bin_idx itself (the FFT path's (s - 1) mod N); en_out the
// window's energy (determine_energy, :368-375) when want_energy.
// ZM = true: the evaluation of a window that holds a sample of exactly zero (see kPoisonBin) - the same estimator from memory, every ifreq value formed as
// the reference forms it (ifreq_prod_z), rolled.
template <int SF, bool ZM = false>
__device__ __forceinline__ void wave_demod_symbol_grad(const DevParams &P, const float *__restrict__ Tv, const float2 *__restrict__ x, bool want_energy,
                                                       uint32_t &bin_out, int32_t &fine_out, float &en_out)
{
    using G = WaveGeom<SF>;
    constexpr int N = G::N, J = G::J, SPS = G::SPS;
    int lane = threadIdx.x & 63;
    asm volatile("" : "+v"(lane)); // (as in wave_demod_symbol: keeps per-lane addresses out of the caller's loop-invariant set)
    if constexpr (ZM) {
        en_out = 0.0f;
        if (want_energy) { // (the same sums in the same order as below)
            const auto xe = (const __attribute__((address_space(1))) v2f *)x;
            v2f e2 = (v2f){0.0f, 0.0f};
#pragma unroll 1
            for (int j = 0; j < J; j++) { const v2f aj = xe[j * 64 + lane]; e2 = __builtin_elementwise_fma(aj, aj, e2); }
            en_out = wave_sum_u(e2.x + e2.y);
        }
        auto fz = [&](int j) { // ifreq[n], n = 64 j + lane; ifreq[sps-1] = ifreq[sps-2] (:243)
            int n = 64 * j + lane;
            n = n == SPS - 1 ? SPS - 2 : n;
            return ifreq_prod_z(x[n], x[n + 1]);
        };
        float bv = 0.1f;
        int bi = 0x7fffffff;
        {
            const int m = lane >> 3;
            const int perm_addr = ((lane - 8) & 63) << 2;
            float prev_perm = 0.0f;
#pragma unroll 1
            for (int j = 0; j < J; j++) {
                float A = fz(j);
                A += dpp_f<kDppQuadXor1>(A); A += dpp_f<kDppQuadXor2>(A); A += dpp_f<kDppRowHalfMirror>(A);
                A *= 0.125f;
                const float perm = __builtin_bit_cast(float, __builtin_amdgcn_ds_bpermute(perm_addr, __builtin_bit_cast(int, A)));
                const float left = (m == 0) ? prev_perm : perm;
                prev_perm = perm;
                const float g = left - A;
                const int i = 8 * j + m;
                if ((j > 0 || m > 0) && g > bv) { bv = g; bi = i; }
            }
        }
        const float best = wave_max_nonneg_u(bv);
        const int first = wave_min_u((bv == best) ? bi : 0x7fffffff);
        const uint32_t max_index = (first == 0x7fffffff) ? 0u : (uint32_t)first + 1u; // :486
        const uint32_t bin_idx = ((uint32_t)N - max_index) % (uint32_t)N;              // :490
        bin_out = bin_idx;
        fine_out = 0;
        if (P.enable_fine_sync == 0u) return;
        const float *__restrict__ vp = Tv + ((int)(bin_idx + 1u) * 8 + SPS) + (lane - 1);
        float c0 = 0.f, c1 = 0.f, c2 = 0.f;
#pragma unroll 1
        for (int j = 0; j < J; j++) {
            const float fj = fz(j);
            c0 += fj * vp[64 * j]; c1 += fj * vp[64 * j + 1]; c2 += fj * vp[64 * j + 2];
        }
        c0 = wave_sum_u(c0); c1 = wave_sum_u(c1); c2 = wave_sum_u(c2);
        float mx = 0.0f;
        int32_t lag = 0;
        if (c0 > mx) { mx = c0; lag = -1; }
        if (c1 > mx) { mx = c1; lag = 0; }
        if (c2 > mx) { mx = c2; lag = 1; }
        fine_out = -lag;
        return;
    }
    const auto xv = (const __attribute__((address_space(1))) v2f *)x;
    v2f a[J];
#pragma unroll
    for (int j = 0; j < J; j++) a[j] = xv[j * 64 + lane];
    __builtin_amdgcn_sched_barrier(0); // all loads issued before the first use
    en_out = 0.0f;
    if (want_energy) {
        v2f e2 = (v2f){0.0f, 0.0f};
#pragma unroll
        for (int j = 0; j < J; j++) e2 = __builtin_elementwise_fma(a[j], a[j], e2);
        en_out = wave_sum_u(e2.x + e2.y);
    }
    float f[J];
    {
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
    }
    // bin averages (:474-477) and the largest drop (:479-488)
    float bv = 0.1f; // max_gradient = 0.1f
    int bi = 0x7fffffff;
    float gs = 0.0f; // (carries the poison of a zero sample when there is no fine_sync sum to carry it)
    {
        const int m = lane >> 3;
        const int perm_addr = ((lane - 8) & 63) << 2;
        float prev_perm = 0.0f;
#pragma unroll
        for (int j = 0; j < J; j++) {
            float A = f[j];
            A += dpp_f<kDppQuadXor1>(A); A += dpp_f<kDppQuadXor2>(A); A += dpp_f<kDppRowHalfMirror>(A); // sum over the 8 lanes of the bin
            A *= 0.125f; // / d_decim_factor
            const float perm = __builtin_bit_cast(float, __builtin_amdgcn_ds_bpermute(perm_addr, __builtin_bit_cast(int, A))); // bin (m - 1) mod 8 of this register
            const float left = (m == 0) ? prev_perm : perm; // bin i - 1: for m = 0 bin 7 of the previous register
            prev_perm = perm;
            const float g = left - A; // samples_ifreq_avg[i - 1] - samples_ifreq_avg[i]
            gs += A;
            const int i = 8 * j + m;
            if ((j > 0 || m > 0) && g > bv) { bv = g; bi = i; } // i runs from 1; strict '>' keeps the first maximum
        }
    }
    const float best = wave_max_nonneg_u(bv);
    const int first = wave_min_u((bv == best) ? bi : 0x7fffffff);
    const uint32_t max_index = (first == 0x7fffffff) ? 0u : (uint32_t)first + 1u; // :486
    const uint32_t bin_idx = ((uint32_t)N - max_index) % (uint32_t)N;              // :490
    bin_out = bin_idx;
    fine_out = 0;
    if (P.enable_fine_sync == 0u) {
        if (poisoned(wave_sum_u(gs))) bin_out = kPoisonBin; // (a sample of exactly zero in the window)
        return;
    }
    // the closed form first (wave_demod_symbol FMODE 2 explains the rule): the ifreq values are in registers, so F = sum_k ifreq[k] is one add per sample;
    // a window it vouches for - lag 0 - skips the three correlations
    if (P.ffs_on != 0u && bin_idx != (uint32_t)N - 1u) { // (uniform; the ZM instantiation has returned above)
        constexpr int CLS = kFfsClass<SF>;
        float fsum = 0.0f, amax = 0.0f;
#pragma unroll
        for (int j = 0; j < J; j++) {
            fsum += f[j];
            if constexpr (CLS != 0) {
                float am = fabsf(f[j]);
                if (j == 0) am = lane < 3 ? 0.0f : am;         // (the ifreq values next to the window's ends are not held to the class)
                if (j == J - 1) am = lane >= 59 ? 0.0f : am;
                amax = fmaxf(amax, am);
            }
        }
        const float F = wave_sum_u(fsum); // (NaN: a sample of exactly zero - not vouched for; the sums below then report the poison)
        const float amx = CLS != 0 ? wave_max_nonneg_u(amax) : 0.0f;
        constexpr float kBound = CLS == 1 ? 1.57079632679489662f : 0.46364760900080609f;
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
    // fine_sync (:300-338), lags -1, 0, +1: c_lag = sum_k f[k] v[(bin_idx + 1) 8 + sps + lag + k], k = 64 j + lane
    const float *__restrict__ vp = Tv + ((int)(bin_idx + 1u) * 8 + SPS) + (lane - 1);
    float c0 = 0.f, c1 = 0.f, c2 = 0.f;
#pragma unroll
    for (int j = 0; j < J; j++) {
        const float fj = f[j];
        c0 += fj * vp[64 * j]; c1 += fj * vp[64 * j + 1]; c2 += fj * vp[64 * j + 2];
    }
    c0 = wave_sum_u(c0); c1 = wave_sum_u(c1); c2 = wave_sum_u(c2);
    if (poisoned3(c0, c1, c2)) { bin_out = kPoisonBin; return; } // (uniform) a sample of the window is exactly zero: the bin averages next to it are NaN as well
    float mx = 0.0f;
    int32_t lag = 0;
    if (c0 > mx) { mx = c0; lag = -1; }
    if (c1 > mx) { mx = c1; lag = 0; }
    if (c2 > mx) { mx = c2; lag = 1; }
    fine_out = -lag;
}

// copies the packed table block (down | twn | tws | xst) and the ifreq template into LDS; all threads of the block; the caller synchronises
template <int SF> constexpr uint32_t kWaveLdsBytes = WaveGeom<SF>::n_ent * 8u + ((3u * WaveGeom<SF>::SPS + 40u + 3u) & ~3u) * 4u; // [tables | ifreq template]
template <int SF>
__device__ __forceinline__ WaveTabs wave_tabs_to_lds(const DevParams &P, v2f *l2, float *lv, uint32_t nthreads)
{
    using G = WaveGeom<SF>;
    const v2f *__restrict__ src = reinterpret_cast<const v2f *>(P.wave_tabs);
    for (uint32_t i = threadIdx.x; i < G::n_ent; i += nthreads) l2[i] = src[i];
    for (uint32_t i = threadIdx.x; i < 3u * G::SPS + 40u; i += nthreads) lv[i] = P.up_ifreq_v[i];
    WaveTabs T{};
    T.down = l2; T.twn = l2 + G::n_down; T.tws = T.twn + G::n_twn; T.xst = T.tws + G::n_tws; T.v = lv;
    return T;
}
template <int SF>
__device__ __forceinline__ WaveTabs wave_tabs_to_lds(const DevParams &P, unsigned char *lds, uint32_t nthreads)
{ // kWaveLdsBytes<SF> bytes
    v2f *l2 = reinterpret_cast<v2f *>(lds);
    return wave_tabs_to_lds<SF>(P, l2, reinterpret_cast<float *>(l2 + WaveGeom<SF>::n_ent), nthreads);
}

// host side: the table block in the layout above, from the handle's downchirp
static void build_wave_tables_host(uint32_t sf, const float2 *down, float *out /* 2 * n_ent floats */)
{
    const int N = 1 << sf, J = N / 8, SPS = 8 * N;
    int logj = 0;
    while ((1 << logj) < J) logj++;
    auto put = [&](size_t idx, double re, double im) {
        out[2 * idx + 0] = (float)re; out[2 * idx + 1] = (float)im;
    };
    size_t o = 0;
    for (int n = 0; n < SPS; n++) { out[2 * o + 0] = down[n].x; out[2 * o + 1] = down[n].y; o++; }
    for (int m = 0; m < J; m++)
        for (int lq = 0; lq < 8; lq++) {
            const int k1 = brev_bits(m, logj);
            const int t = (lq * k1) % N;
            const double a = -2.0 * M_PI * (double)t / (double)N;
            put(o++, std::cos(a), std::sin(a));
        }
    for (int g = 0; g < J; g++)
        for (int lane = 0; lane < 64; lane++) {
            const int r = lane & 7;
            const int jb = wave_layout_bin(J, logj, g, lane);
            const int k = (jb < N / 2) ? jb : jb - N;
            const int e = ((k * r) % SPS + SPS) % SPS;
            const double ang = -2.0 * M_PI * (double)e / (double)SPS;
            double re = std::cos(ang), im = std::sin(ang);
            if (jb == N / 2) { // tmp[N/2] += F[N/2] (:450)
                const int e2 = ((N / 2) * r) % SPS;
                const double a2 = -2.0 * M_PI * (double)e2 / (double)SPS;
                re += std::cos(a2); im += std::sin(a2);
            }
            put(o++, re, im);
        }
    const double rs = 0.70710678118654752440;
    for (int lane = 0; lane < 64; lane++) { // stage 1 (lq bit 2): the difference is multiplied by W_8^{lq & 3}
        const int t1 = (lane >> 3) & 3;
        if (t1 == 0) put(o++, 1.0, 0.0);
        else if (t1 == 1) put(o++, rs, -rs);
        else if (t1 == 2) put(o++, 0.0, -1.0);
        else put(o++, -rs, -rs);
    }
    for (int lane = 0; lane < 64; lane++) { // stage 2 (lq bit 1): the difference is multiplied by W_4^{lq & 1}
        if ((lane >> 3) & 1) put(o++, 0.0, -1.0);
        else put(o++, 1.0, 0.0);
    }
}

uint32_t wave_tables_floats(uint32_t sf) { return sf == 7u ? 2u * WaveGeom<7>::n_ent : (sf == 8u ? 2u * WaveGeom<8>::n_ent : (sf == 9u ? 2u * WaveGeom<9>::n_ent : 0u)); }
void build_wave_tables(uint32_t sf, const float2 *down, float *out) { build_wave_tables_host(sf, down, out); }

// ---- symbol-level kernels: one wavefront per symbol, for lora_hip_demod_symbols_device and the payload pass of a decoupled pass ------------------
template <int SF>
__global__ __launch_bounds__(256) void demod_symbols_wave_grad_kernel(DevParams P, const float2 *iq, const int64_t *offsets, uint32_t n, uint32_t *bins, int32_t *fine)
{
    using G = WaveGeom<SF>;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    float *lds_v = reinterpret_cast<float *>(smem);
    for (uint32_t i = threadIdx.x; i < 3u * G::SPS + 40u; i += 256u) lds_v[i] = P.up_ifreq_v[i];
    __syncthreads();
    const uint32_t wave = threadIdx.x >> 6;
    for (uint32_t s = blockIdx.x * 4u + wave; s < n; s += gridDim.x * 4u) {
        uint32_t b;
        int32_t fs;
        float en;
        wave_demod_symbol_grad<SF>(P, lds_v, iq + offsets[s], false, b, fs, en);
        if (b == kPoisonBin) wave_demod_symbol_grad<SF, true>(P, lds_v, iq + offsets[s], false, b, fs, en); // (uniform) a window with a sample of exactly zero
        if ((threadIdx.x & 63u) == 0u) { bins[s] = b; if (fine) fine[s] = fs; }
    }
}

// The FFT demodulators.  512-thread workgroups; OCC = wavefronts per SIMD the register budget is set for: SF7 6 (80 registers, three workgroups per CU),
// SF8 4 (128, two per CU), SF9 2 (256: 64 samples per lane and fine_sync's 64 ifreq values beside them - a second read of the window instead costs 40 %,
// 0.268 against 0.373 of HBM peak; the cooperative w3_demod_round it replaces: 0.207).  Same-box A/B of the geometries: profiles/r05_ab_sf9_wave_fft.txt.
// second reads (DemodAlt): the wavefront whose symbol moved the symbol clock reads that symbol's successor again, that far on (as demod_symbols_w3_grad_kernel)
template <int SF, int OCC>
__global__ __launch_bounds__(512, OCC) void demod_symbols_wave_kernel(DevParams P, const float2 *iq, const int64_t *offsets, uint32_t n, uint32_t *bins, int32_t *fine, DemodAlt alt)
{
    constexpr int SPS = 8 << SF;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const WaveTabs T = wave_tabs_to_lds<SF>(P, smem, 512u);
    __syncthreads();
    const uint32_t wave = threadIdx.x >> 6, lane = threadIdx.x & 63u;
    for (uint32_t s = blockIdx.x * 8u + wave; s < n; s += gridDim.x * 8u) {
        const int64_t o0 = offsets[s];
        uint32_t b;
        int32_t fs;
        wave_demod_symbol<SF, kWaveFmode<SF>>(P, T, iq + o0, b, fs);
        if (b == kPoisonBin) wave_demod_symbol<SF, 1, true>(P, T, iq + o0, b, fs); // (uniform) a window with a sample of exactly zero
        if (lane == 0u) { bins[s] = b; if (fine) fine[s] = fs; }
        int32_t sh = 0; // the shift this wavefront's second read of its successor was made at (0: none)
        if (alt.shift && fs != 0 && s + 1u < n) {
            const int64_t o1 = offsets[s + 1u], a = o1 + (int64_t)fs;
            if (o1 == o0 + (int64_t)SPS && a >= 0 && a <= alt.max_start) {
                uint32_t b2;
                int32_t f2;
                wave_demod_symbol<SF, kWaveFmode<SF>>(P, T, iq + a, b2, f2);
                if (b2 == kPoisonBin) wave_demod_symbol<SF, 1, true>(P, T, iq + a, b2, f2);
                if (lane == 0u) { alt.bins[s + 1u] = b2; alt.fine[s + 1u] = f2; }
                sh = fs;
            }
        }
        // DemodAlt.shift[s + 1] is this wavefront's to write, taken or not (and shift[0] the first one's): the caller clears nothing
        if (alt.shift && lane == 0u) { if (s + 1u < n) alt.shift[s + 1u] = sh; if (s == 0u) alt.shift[0] = 0; }
    }
}
