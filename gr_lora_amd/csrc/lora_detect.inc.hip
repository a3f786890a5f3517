// lora_detect.inc.hip -- FFT-domain preamble detection (beyond the reference).  Included by lora_kernels.hip.
//
// The reference acquires with a time-domain autocorrelation of adjacent symbols (detect_preamble_autocorr,
// lib/decoder_impl.cc:340-366, gate 0.90 at :755) and an instantaneous-frequency correlation (gate 0.96 at :792): both fail
// once the noise of the full sample-rate band is comparable to the signal (0 of 6 packets at <= 20 dB).  The
// spreading gain is in the dechirped spectrum: this kernel computes, for caller-given window positions, the pruned dechirp
// spectrum of get_shift_fft (:430-464; the N bins k in [-N/2, N/2) of the sps-point DFT of x * d_downchirp, no N/2 fold) of the
// window AND of its complex conjugate - the first sees upchirps, the second downchirps (bins mirrored) - and returns peak bin,
// peak power and total power of both.  The host (lora_hip_detect_preambles_device, lora_runtime.cpp) finds runs of consistent
// peaks, aligns the symbol clock to them and locates the SFD.  Definition and float64 restatement: oracle/preamble_oracle.py.
//
// One 256-thread workgroup per window, the generic polyphase FFT of get_shift_fft above (any SF / decimation): 8 B per item
// read twice (once per reference; the second pass hits L2), not a hot path - a stream is scanned once at one window per symbol.

struct DetectStats { // mirrors lora_hip_window_stats_t
    int32_t bin_down; float peak_down, total_down;
    int32_t bin_up; float peak_up, total_up;
};

// |X[k]|^2 statistics of one window; CONJ: of its complex conjugate.  All threads of the workgroup; results uniform.
template <bool CONJ>
__device__ void detect_spectrum(const DevParams &P, const float2 *__restrict__ x, float2 *work, float *red, int &bin_out, float &peak_out, float &total_out)
{
    const uint32_t N = P.nbins, D = P.decim, G = P.fft_groups, DG = D / G, logN = P.log_nbins;
    const uint32_t stride = P.fft_stride, pts = N * DG, smask = P.sps - 1u;
    float2 acc[kMaxBinsPerThread];
#pragma unroll
    for (int m = 0; m < kMaxBinsPerThread; m++) acc[m] = make_float2(0.0f, 0.0f);
    for (uint32_t g = 0; g < G; g++) {
        for (uint32_t idx = threadIdx.x; idx < pts; idx += kWG) {
            const uint32_t rr = idx % DG, q = idx / DG;
            const uint32_t n = q * D + g * DG + rr;
            float2 v = x[n];
            if (CONJ) v.y = -v.y;
            work[rr * stride + q] = cmul(v, P.down[n]);
        }
        __syncthreads();
        for (uint32_t h = N >> 1; h >= 1u; h >>= 1) {
            const uint32_t tw_step = (N >> 1) / h;
            for (uint32_t b = threadIdx.x; b < (pts >> 1); b += kWG) {
                const uint32_t arr = b / (N >> 1), j = b % (N >> 1);
                const uint32_t off = j & (h - 1u), blk = j / h;
                const uint32_t i0 = arr * stride + blk * 2u * h + off, i1 = i0 + h;
                const float2 a = work[i0], c = work[i1];
                const float2 d = make_float2(a.x - c.x, a.y - c.y);
                work[i0] = make_float2(a.x + c.x, a.y + c.y);
                work[i1] = cmul(d, P.twN[off * tw_step]);
            }
            __syncthreads();
        }
#pragma unroll
        for (int m = 0; m < kMaxBinsPerThread; m++) {
            const uint32_t j = threadIdx.x + (uint32_t)m * kWG;
            if (j < N) {
                const int32_t k = (j < N / 2u) ? (int32_t)j : (int32_t)j - (int32_t)N;
                const uint32_t jr = bitrev(j, logN);
                for (uint32_t rr = 0; rr < DG; rr++) {
                    const uint32_t r = g * DG + rr;
                    const float2 t = cmul(work[rr * stride + jr], P.tws[(uint32_t)(k * (int32_t)r) & smask]);
                    acc[m].x += t.x; acc[m].y += t.y;
                }
            }
        }
        __syncthreads();
    }
    float bv = -1.0f, tot[1] = {0.0f};
    int bi = 0;
#pragma unroll
    for (int m = 0; m < kMaxBinsPerThread; m++) {
        const uint32_t j = threadIdx.x + (uint32_t)m * kWG;
        if (j < N) {
            const float pw = acc[m].x * acc[m].x + acc[m].y * acc[m].y;
            tot[0] += pw;
            if (pw > bv) { bv = pw; bi = (int)j; }
        }
    }
    block_sum<1>(tot, red);
    block_argmax_first(bv, bi, red);
    bin_out = bi; peak_out = bv; total_out = tot[0];
    __syncthreads();
}

__global__ __launch_bounds__(kWG) void detect_windows_kernel(DevParams P, const float2 *iq, const int64_t *offsets, uint32_t n, DetectStats *out)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    float2 *work = reinterpret_cast<float2 *>(smem);
    Shared &sh = *reinterpret_cast<Shared *>(smem + P.lds_work_bytes);
    for (uint32_t s = blockIdx.x; s < n; s += gridDim.x) {
        const float2 *x = iq + offsets[s];
        DetectStats r;
        detect_spectrum<false>(P, x, work, sh.red, r.bin_down, r.peak_down, r.total_down);
        detect_spectrum<true>(P, x, work, sh.red, r.bin_up, r.peak_up, r.total_up);
        if (threadIdx.x == 0) out[s] = r;
    }
}

int launch_detect_windows(const DevParams &p, const float2 *iq, const int64_t *d_offsets, uint32_t n, void *d_out, void *stream)
{
    if (n == 0) return 0;
    const uint32_t lds = walker_lds_bytes(p);
    if (lds > 64u * 1024u) {
        if (hipFuncSetAttribute((const void *)detect_windows_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess) return -1;
    }
    hipLaunchKernelGGL(detect_windows_kernel, dim3(n < 4096u ? n : 4096u), dim3(kWG), lds, (hipStream_t)stream, p, iq, d_offsets, n, (DetectStats *)d_out);
    return hipGetLastError() == hipSuccess ? 0 : -1;
}
