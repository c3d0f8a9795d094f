// Measured issue rate of v_mfma_f32_32x32x2_f32 on the box (the ceiling the GA3C forward kernel is priced against).
// hipcc --offload-arch=gfx950 -O3 -o /tmp/mfma_rate tools/micro/mfma_rate.hip && /tmp/mfma_rate
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));
template <int NACC>
__global__ void __launch_bounds__(256) k(float* out, int iters, float a, float b) {
    f32x16 acc[NACC];
    for (int i = 0; i < NACC; i++)
        for (int j = 0; j < 16; j++) acc[i][j] = (float)threadIdx.x;
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int i = 0; i < NACC; i++) acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[i], 0, 0, 0);
    }
    float s = 0;
    for (int i = 0; i < NACC; i++)
        for (int j = 0; j < 16; j++) s += acc[i][j];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}
template <int NACC>
void run(int blocks, const char* name) {
    float* out;
    hipMalloc(&out, (size_t)blocks * 256 * 4);
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    const int iters = 4000;
    k<NACC><<<blocks, 256>>>(out, iters, 1.0f, 0.5f);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    k<NACC><<<blocks, 256>>>(out, iters, 1.0f, 0.5f);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    const double flop = (double)blocks * 4 * iters * NACC * 4096.0;
    printf("%s: blocks %d acc %d: %.3f ms, %.1f TFLOP/s\n", name, blocks, NACC, ms, flop / ms / 1e9);
    hipFree(out);
}
int main() {
    run<1>(256, "1 wave/SIMD, dependent chain");
    run<2>(256, "1 wave/SIMD, 2 accumulators");
    run<4>(256, "1 wave/SIMD, 4 accumulators");
    run<2>(512, "2 waves/SIMD, 2 accumulators");
    run<2>(1024, "4 waves/SIMD, 2 accumulators");
    return 0;
}
