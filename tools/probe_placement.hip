// Where do the wavefronts of co-resident workgroups land?  tools/probe_placement.hip: per workgroup size (threads) the
// SIMD of every wavefront (HW_ID), for two workgroups per CU (LDS-limited like the SF7 walker).
// hipcc --offload-arch=gfx950 -O2 -o tools/probe_placement.bin tools/probe_placement.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <map>
#include <string>
#include <vector>
__global__ void probe(uint32_t *out, int spin)
{
    extern __shared__ unsigned char smem[];
    const int wave = threadIdx.x >> 6, nw = blockDim.x >> 6;
    if ((threadIdx.x & 63) == 0) {
        const uint32_t hw = __builtin_amdgcn_s_getreg((31 << 11) | 4), xcc = __builtin_amdgcn_s_getreg((31 << 11) | 20);
        out[(blockIdx.x * nw + wave) * 2 + 0] = hw;
        out[(blockIdx.x * nw + wave) * 2 + 1] = xcc & 0xf;
        smem[wave] = 1;
    }
    const long long t0 = clock64();
    while (clock64() - t0 < spin) __builtin_amdgcn_s_sleep(8);
    __syncthreads();
}
int main()
{
    for (int threads : {512, 448, 1024, 256}) {
        const int nw = threads / 64, lds = threads == 1024 ? 140 * 1024 : 73 * 1024, per_cu = threads == 1024 ? 1 : 2;
        const int blocks = 256 * per_cu;
        uint32_t *d;
        hipMalloc(&d, blocks * nw * 8);
        hipFuncSetAttribute((const void *)probe, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
        std::vector<uint32_t> h(blocks * nw * 2);
        for (int rep = 0; rep < 2; rep++) {
            hipLaunchKernelGGL(probe, dim3(blocks), dim3(threads), lds, 0, d, 400000);
            hipMemcpy(h.data(), d, h.size() * 4, hipMemcpyDeviceToHost);
        }
        std::map<std::string, int> pat;                 // SIMD sequence of a workgroup's wavefronts
        std::map<uint32_t, std::vector<int>> cu_blocks; // CU -> blocks
        for (int b = 0; b < blocks; b++) {
            std::string s;
            for (int w = 0; w < nw; w++) s += char('0' + ((h[(b * nw + w) * 2] >> 4) & 3));
            pat[s]++;
            const uint32_t hw = h[b * nw * 2], key = (h[b * nw * 2 + 1] << 16) | (hw & 0xff00);
            cu_blocks[key].push_back(b);
        }
        printf("threads %d: %zu CUs used, patterns:", threads, cu_blocks.size());
        for (auto &p : pat) printf(" %s x%d", p.first.c_str(), p.second);
        printf("\n");
        std::map<std::string, int> pairs; // per CU: SIMD of the last wavefront of each block + waves per SIMD
        for (auto &c : cu_blocks) {
            std::string s;
            int cnt[4] = {0, 0, 0, 0};
            for (int b : c.second) {
                s += char('0' + ((h[(b * nw + nw - 1) * 2] >> 4) & 3));
                for (int w = 0; w < nw; w++) cnt[(h[(b * nw + w) * 2] >> 4) & 3]++;
            }
            char buf[64];
            snprintf(buf, sizeof buf, "last@%s n=%d%d%d%d wg=%zu", s.c_str(), cnt[0], cnt[1], cnt[2], cnt[3], c.second.size());
            pairs[buf]++;
        }
        for (auto &p : pairs) printf("   %s x%d\n", p.first.c_str(), p.second);
        hipFree(d);
    }
    return 0;
}
