// Calibration of rocprofv3's FETCH_SIZE for the walker's access pattern: every wave instruction loads 8 B per lane,
// 512 contiguous bytes (global_load_dwordx2), streaming once through a buffer far larger than the 256 MiB L3.
//   hipcc --offload-arch=gfx950 -O3 tools/calib_fetch.hip -o tools/calib_fetch.bin
//   rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d out -- tools/calib_fetch.bin
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void read8(const float2 *__restrict__ p, size_t n, float *out)
{
    float acc = 0.f;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) { const float2 v = p[i]; acc += v.x + v.y; }
    if (acc == 123.456f) out[0] = acc;
}
__global__ void read16(const float4 *__restrict__ p, size_t n, float *out)
{
    float acc = 0.f;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) { const float4 v = p[i]; acc += v.x + v.y + v.z + v.w; }
    if (acc == 123.456f) out[0] = acc;
}
int main()
{
    const size_t bytes = 2ull << 30; // 2 GiB
    void *p; float *o;
    hipMalloc(&p, bytes); hipMalloc(&o, 4); hipMemset(p, 0, bytes);
    for (int r = 0; r < 2; r++) {
        hipLaunchKernelGGL(read8, dim3(2048), dim3(256), 0, 0, (const float2 *)p, bytes / 8, o);
        hipLaunchKernelGGL(read16, dim3(2048), dim3(256), 0, 0, (const float4 *)p, bytes / 16, o);
    }
    hipDeviceSynchronize();
    printf("read %zu bytes per kernel\n", bytes);
    return 0;
}
