// Micro-benchmark of wave_demod_symbol: device-clock stamps per phase for one wavefront, and throughput for a
// grid of `blocks` workgroups of `waves` wavefronts, each demodulating `reps` symbols.  Build + run:
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -I gr_lora_amd/csrc tools/probe_phases.hip -o /tmp/probe_phases
//   /tmp/probe_phases <blocks> <reps> <waves>
#include "lora_kernels.hip"
#include <vector>
#include <cstdio>
#include <cmath>
namespace lora_hip {
template <int SF>
__global__ __launch_bounds__(1024) void probe(DevParams P, const float2 *iq, uint32_t *out, long long *stamps, int reps)
{
    using G = WaveGeom<SF>;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    v4f *lds4 = reinterpret_cast<v4f *>(smem);
    float *lds_v = reinterpret_cast<float *>(lds4 + G::n_v4f);
    const WaveTabs T = wave_tabs_to_lds<SF>(P, lds4, lds_v, blockDim.x);
    __syncthreads();
    const int wave = threadIdx.x >> 6, nw = blockDim.x >> 6;
    long long st[8];
    uint32_t s = 0; int32_t f = 0;
    const long long t0 = clock64();
    for (int r = 0; r < reps; r++) {
        const size_t sym = ((size_t)blockIdx.x * nw + wave) * reps + r;
        wave_demod_symbol<SF, SF == 7 ? 1 : 0>(P, T, iq + sym * G::SPS, s, f, nullptr, nullptr, (blockIdx.x == 0 && wave == 0 && r == reps - 1) ? st : nullptr);
    }
    const long long t1 = clock64();
    if ((threadIdx.x & 63) == 0) { out[(blockIdx.x * nw + wave) * 2] = s; out[(blockIdx.x * nw + wave) * 2 + 1] = (uint32_t)f; }
    if (threadIdx.x == 0 && blockIdx.x == 0) { for (int i = 0; i < 8; i++) stamps[i] = st[i]; stamps[8] = t1 - t0; }
}
}
using namespace lora_hip;
int main(int argc, char **argv)
{
    const int SF = 7, N = 128, SPS = 1024;
    const int blocks = argc > 1 ? atoi(argv[1]) : 1, reps = argc > 2 ? atoi(argv[2]) : 4, waves = argc > 3 ? atoi(argv[3]) : 4, fine = argc > 4 ? atoi(argv[4]) : 1;
    std::vector<float2> down(SPS);
    std::vector<float> v(3 * SPS + 64, 0.1f);
    for (int i = 0; i < SPS; i++) { double a = 2 * M_PI * i * i / (16.0 * SPS); down[i] = make_float2(cos(a), sin(a)); }
    std::vector<float> wt(wave_tables_floats(SF));
    build_wave_tables(SF, down.data(), wt.data());
    DevParams P{}; P.sf = SF; P.nbins = N; P.sps = SPS; P.decim = 8; P.enable_fine_sync = fine; P.demod_mode = 2;
    float *d_wt, *d_v; float2 *d_iq; uint32_t *d_out; long long *d_st;
    hipMalloc(&d_wt, wt.size() * 4); hipMalloc(&d_v, v.size() * 4);
    const size_t nsym = (size_t)blocks * waves * reps;
    hipMalloc(&d_iq, nsym * SPS * 8); hipMalloc(&d_out, (size_t)blocks * waves * 8); hipMalloc(&d_st, 9 * 8);
    hipMemcpy(d_wt, wt.data(), wt.size() * 4, hipMemcpyHostToDevice); hipMemcpy(d_v, v.data(), v.size() * 4, hipMemcpyHostToDevice);
    std::vector<float2> iq(nsym * SPS);
    for (size_t i = 0; i < iq.size(); i++) { double a = -2 * M_PI * (i % SPS) * (i % SPS) / (16.0 * SPS) + 0.3 * (i % SPS); iq[i] = make_float2(cos(a), sin(a)); }
    hipMemcpy(d_iq, iq.data(), iq.size() * 8, hipMemcpyHostToDevice);
    P.wave_tabs = d_wt; P.up_ifreq_v = d_v;
    const size_t lds = wt.size() * 4 + ((3 * SPS + 40 + 3) & ~3) * 4;
    hipFuncSetAttribute((const void *)probe<7>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int it = 0; it < 3; it++) {
        hipEventRecord(e0);
        hipLaunchKernelGGL(probe<7>, dim3(blocks), dim3(64 * waves), lds, 0, P, d_iq, d_out, d_st, reps);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        long long st[9]; hipMemcpy(st, d_st, 72, hipMemcpyDeviceToHost);
        printf("blocks %d waves %d reps %d: %.3f ms (%.1f Gsamples/s, %.1f GB/s) wave0 %.0f cyc/symbol; phases:", blocks, waves, reps, ms, nsym * SPS / ms / 1e6,
               nsym * SPS * 8 / ms / 1e6, (double)st[8] / reps);
        const char *nm[7] = {"load+ifreq", "dechirp+fft16", "twiddle", "xlane8", "combine+redscat", "argmax", "fine"};
        for (int i = 0; i < 7; i++) printf(" %s=%lld", nm[i], st[i + 1] - st[i]);
        printf(" total=%lld\n", st[7] - st[0]);
    }
    return 0;
}
