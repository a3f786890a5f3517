// LDS read latency/throughput as used by the wave demodulator: each wave reads 16-byte table entries
// (lane-contiguous) in batches of B reads per s_waitcnt, multiplies them into an accumulator.
//   hipcc --offload-arch=gfx950 -O3 tools/ubench_lds.hip -o tools/ubench_lds.bin
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float v4f __attribute__((ext_vector_type(4)));
template <int B, int WIDTH>
__global__ void k(float *out, int iters)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    float *f = reinterpret_cast<float *>(smem);
    for (int i = threadIdx.x; i < 12288; i += blockDim.x) f[i] = 1.0f + 1e-6f * i;
    __syncthreads();
    const int lane = threadIdx.x & 63;
    float acc = 0.0f;
    for (int it = 0; it < iters; it++) {
        const int base = ((it * 7) & 15) * 64; // entries
        if (WIDTH == 16) {
            const v4f *t = reinterpret_cast<const v4f *>(smem) + base + lane;
            v4f r[B];
#pragma unroll
            for (int b = 0; b < B; b++) r[b] = t[b * 64];
#pragma unroll
            for (int b = 0; b < B; b++) acc = fmaf(acc, r[b].x, r[b].y + r[b].z * r[b].w);
        } else {
            const float *t = f + base + lane;
            float r[B];
#pragma unroll
            for (int b = 0; b < B; b++) r[b] = t[b * 64];
#pragma unroll
            for (int b = 0; b < B; b++) acc = fmaf(acc, r[b], 1.0f);
        }
        asm volatile("" : "+v"(acc));
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = acc;
}
template <int B, int WIDTH>
static void run()
{
    float *out; hipMalloc(&out, 256 * 1024 * 4);
    const int iters = 20000;
    printf("width %2d B, %2d reads per wait:", WIDTH, B);
    for (int wpc = 4; wpc <= 16; wpc *= 2) {
        hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
        hipLaunchKernelGGL((k<B, WIDTH>), dim3(256), dim3(64 * wpc), 49152, 0, out, iters);
        hipEventRecord(e0);
        hipLaunchKernelGGL((k<B, WIDTH>), dim3(256), dim3(64 * wpc), 49152, 0, out, iters);
        hipEventRecord(e1); hipDeviceSynchronize();
        float ms; hipEventElapsedTime(&ms, e0, e1);
        const double cyc_per_read_per_cu = ms * 1e-3 * 2.4e9 / ((double)iters * B * wpc);
        printf("  %2d waves/CU: %6.1f cyc/read/CU (%5.0f cyc per batch per wave)", wpc, cyc_per_read_per_cu, ms * 1e-3 * 2.4e9 / iters);
    }
    printf("\n");
    hipFree(out);
}
int main()
{
    run<1, 16>(); run<2, 16>(); run<4, 16>(); run<8, 16>(); run<16, 16>();
    run<1, 4>(); run<4, 4>(); run<16, 4>();
    return 0;
}
