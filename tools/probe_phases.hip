// Micro-benchmark of fast_demod_symbol phases (device clock stamps). Build:
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -I gr_lora_amd/csrc tools/probe_phases.hip -o /tmp/probe_phases
#include "lora_kernels.hip"
#include <vector>
#include <cstdio>
#include <cmath>
namespace lora_hip {
template <int SF>
__global__ __launch_bounds__(256) void probe(DevParams P, const float2 *iq, uint32_t *out, long long *stamps, int reps)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    constexpr int SPS = 8 << SF;
    float *vl = reinterpret_cast<float *>(smem);
    const uint32_t nv = (3u * SPS + 4u * 8u + 8u + 3u) & ~3u;
    float2 *tw_s = reinterpret_cast<float2 *>(vl + nv);
    float2 *tw_n = tw_s + SPS;
    float2 *dn = tw_n + (1 << SF) / 2;
    for (uint32_t i = threadIdx.x; i < SPS; i += 256) dn[i] = P.down[i];
    for (uint32_t i = threadIdx.x; i < 3u * SPS + 40u; i += 256) vl[i] = P.up_ifreq_v[i];
    for (uint32_t i = threadIdx.x; i < SPS; i += 256) tw_s[i] = P.tws[i];
    for (uint32_t i = threadIdx.x; i < P.nbins / 2u; i += 256) tw_n[i] = P.twN[i];
    __syncthreads();
    FastTabs T{vl, tw_s, tw_n, dn};
    const int wave = threadIdx.x >> 6;
    long long st[9];
    uint32_t s = 0; int32_t f = 0;
    for (int r = 0; r < reps; r++) {
        const size_t sym = ((size_t)blockIdx.x * 4 + wave) * reps + r;
        fast_demod_symbol<SF>(P, T, iq + sym * SPS, s, f, (blockIdx.x == 0 && wave == 0 && r == reps - 1) ? st : nullptr);
    }
    if ((threadIdx.x & 63) == 0) { out[(blockIdx.x * 4 + wave) * 2] = s; out[(blockIdx.x * 4 + wave) * 2 + 1] = (uint32_t)f; }
    if (threadIdx.x == 0 && blockIdx.x == 0) for (int i = 0; i < 9; i++) stamps[i] = st[i];
}
}
using namespace lora_hip;
int main(int argc, char **argv)
{
    const int SF = 7, N = 128, SPS = 1024;
    const int blocks = argc > 1 ? atoi(argv[1]) : 1, reps = argc > 2 ? atoi(argv[2]) : 4;
    std::vector<float2> down(SPS), tws(SPS), twN(N / 2);
    std::vector<float> v(3 * SPS + 64, 0.1f);
    for (int i = 0; i < SPS; i++) { double a = 2 * M_PI * i * i / (16.0 * SPS); down[i] = make_float2(cos(a), sin(a)); double b = -2 * M_PI * i / SPS; tws[i] = make_float2(cos(b), sin(b)); }
    for (int i = 0; i < N / 2; i++) { double b = -2 * M_PI * i / N; twN[i] = make_float2(cos(b), sin(b)); }
    DevParams P{}; P.sf = SF; P.nbins = N; P.sps = SPS; P.decim = 8; P.enable_fine_sync = 1; P.demod_mode = 2;
    float2 *d_down, *d_tws, *d_twN, *d_iq; float *d_v; uint32_t *d_out; long long *d_st;
    hipMalloc(&d_down, SPS * 8); hipMalloc(&d_tws, SPS * 8); hipMalloc(&d_twN, N * 4); hipMalloc(&d_v, v.size() * 4);
    const size_t nsym = (size_t)blocks * 4 * reps;
    hipMalloc(&d_iq, nsym * SPS * 8); hipMalloc(&d_out, blocks * 8 * 4); hipMalloc(&d_st, 9 * 8);
    hipMemcpy(d_down, down.data(), SPS * 8, hipMemcpyHostToDevice); hipMemcpy(d_tws, tws.data(), SPS * 8, hipMemcpyHostToDevice);
    hipMemcpy(d_twN, twN.data(), N * 4, hipMemcpyHostToDevice); hipMemcpy(d_v, v.data(), v.size() * 4, hipMemcpyHostToDevice);
    std::vector<float2> iq(nsym * SPS);
    for (size_t i = 0; i < iq.size(); i++) { double a = -2 * M_PI * (i % SPS) * (i % SPS) / (16.0 * SPS) + 0.3 * (i % SPS); iq[i] = make_float2(cos(a), sin(a)); }
    hipMemcpy(d_iq, iq.data(), iq.size() * 8, hipMemcpyHostToDevice);
    P.down = d_down; P.tws = d_tws; P.twN = d_twN; P.up_ifreq_v = d_v;
    const size_t lds = ((3 * SPS + 40 + 3) & ~3) * 4 + (2 * SPS + N / 2) * 8;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int it = 0; it < 3; it++) {
        hipEventRecord(e0);
        hipLaunchKernelGGL(probe<7>, dim3(blocks), dim3(256), lds, 0, P, d_iq, d_out, d_st, reps);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        long long st[9]; hipMemcpy(st, d_st, 72, hipMemcpyDeviceToHost);
        printf("blocks %d reps %d: %.3f ms  (%.1f Gsamples/s, %.1f GB/s) phases:", blocks, reps, ms, nsym * SPS / ms / 1e6, nsym * SPS * 8 / ms / 1e6);
        const char *nm[8] = {"load+atan2+dechirp", "fft16", "twiddle", "xlane8", "combine", "redscat", "argmax", "fine"};
        for (int i = 0; i < 8; i++) printf(" %s=%lld", nm[i], st[i + 1] - st[i]);
        printf(" total=%lld\n", st[8] - st[0]);
    }
    return 0;
}
