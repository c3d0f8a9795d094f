// simd_map.hip -- where do the waves of co-resident 256-lane workgroups land?  (diagnostic, not part of the library)
// Launches 1024 workgroups of 4 waves with the headline kernel's footprint (36 KB LDS -> 4 workgroups per CU) that stay
// resident long enough to overlap, and records HW_REG_HW_ID (wave slot, SIMD, CU, SE) + XCC_ID of every wave.
// build/run on the GPU box: hipcc --offload-arch=gfx950 -O2 -o /tmp/simd_map tools/micro/simd_map.hip && /tmp/simd_map
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <map>
#include <vector>
__global__ void __launch_bounds__(256) k(unsigned* out, int spin) {
    extern __shared__ unsigned char smem[];
    const int wave = threadIdx.x >> 6;
    unsigned hw = __builtin_amdgcn_s_getreg(((32 - 1) << 11) | (0 << 6) | 4);   // HW_REG_HW_ID
    unsigned xcc = __builtin_amdgcn_s_getreg(((4 - 1) << 11) | (0 << 6) | 20);  // HW_REG_XCC_ID
    unsigned long long t0 = __builtin_amdgcn_s_memtime();
    while (__builtin_amdgcn_s_memtime() - t0 < (unsigned long long)spin) __builtin_amdgcn_s_sleep(8);
    if ((threadIdx.x & 63) == 0) { out[(blockIdx.x * 4 + wave) * 2] = hw; out[(blockIdx.x * 4 + wave) * 2 + 1] = xcc & 15u; }
    smem[threadIdx.x] = 0;
}
int main() {
    const int NB = 1024;
    unsigned* d;
    hipMalloc(&d, NB * 4 * 2 * sizeof(unsigned));
    hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, 36 * 1024);
    hipLaunchKernelGGL(k, dim3(NB), dim3(256), 36 * 1024, 0, d, 200000);
    hipDeviceSynchronize();
    std::vector<unsigned> h(NB * 8);
    hipMemcpy(h.data(), d, h.size() * 4, hipMemcpyDeviceToHost);
    // histogram: for each wave index, which SIMD; and per CU the (block, wave) -> simd table for the first CUs
    int hist[4][4] = {{0}};
    std::map<unsigned, std::vector<int>> cu;  // key (xcc, se, sh, cu) -> list of block ids
    for (int b = 0; b < NB; b++)
        for (int w = 0; w < 4; w++) {
            unsigned hw = h[(b * 4 + w) * 2], x = h[(b * 4 + w) * 2 + 1];
            int simd = (hw >> 4) & 3;
            hist[w][simd]++;
            if (w == 0) cu[(x << 16) | (hw & 0xff00)].push_back(b);
        }
    for (int w = 0; w < 4; w++) printf("wave %d -> SIMD 0..3: %d %d %d %d\n", w, hist[w][0], hist[w][1], hist[w][2], hist[w][3]);
    printf("distinct CUs: %zu\n", cu.size());
    int shown = 0;
    for (auto& kv : cu) {
        if (shown++ >= 6) break;
        printf("CU key %06x: blocks", kv.first);
        for (int b : kv.second) {
            printf(" %d[", b);
            for (int w = 0; w < 4; w++) printf("%d", (h[(b * 4 + w) * 2] >> 4) & 3);
            printf("]");
        }
        printf("\n");
    }
    return 0;
}
