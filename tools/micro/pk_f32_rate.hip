// Issue cost of packed fp32 (v_pk_mul_f32 / v_pk_add_f32 / v_pk_fma_f32) against scalar fp32 and fp64 VALU ops on gfx950:
// cycles per wave-instruction on one SIMD, as a dependent chain and as 4 independent chains, at 1 / 2 / 4 waves per SIMD.
// (Round 4, review item 2a: "if a packed op costs less than two scalar ones, put the (x, y) arithmetic of the ORCA half on it".)
// hipcc --offload-arch=gfx950 -O3 -o /tmp/pk_f32_rate tools/micro/pk_f32_rate.hip && /tmp/pk_f32_rate
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float v2f __attribute__((ext_vector_type(2)));

#define REP8(x) x x x x x x x x
#define REP32(x) REP8(x) REP8(x) REP8(x) REP8(x)

enum Op { MUL32, FMA32, PKMUL, PKADD, PKFMA, FMA64, ADD64, MUL64, RCP32, SQRT32, CVT_F64_F32, PKMUL_SWZ };

template <int OP, int CHAINS>
__global__ void __launch_bounds__(256) k(float* out, int iters, float seed) {
    v2f p[4];
    float s[4];
    double d[4];
    for (int i = 0; i < 4; i++) {
        p[i] = v2f{seed + threadIdx.x, seed * 0.5f + i};
        s[i] = seed + i + threadIdx.x;
        d[i] = (double)seed + i;
    }
    const v2f c = v2f{1.0000001f, 0.9999999f};
    const float cs = 1.0000001f;
    const double cd = 1.0000001;
    for (int it = 0; it < iters; it++) {
        // 32 instructions per chain per trip
        if (OP == MUL32) {
            REP32(asm volatile("v_mul_f32 %0, %0, %1" : "+v"(s[0]) : "v"(cs));
                  if (CHAINS > 1) { asm volatile("v_mul_f32 %0, %0, %1" : "+v"(s[1]) : "v"(cs)); asm volatile("v_mul_f32 %0, %0, %1" : "+v"(s[2]) : "v"(cs)); asm volatile("v_mul_f32 %0, %0, %1" : "+v"(s[3]) : "v"(cs)); })
        } else if (OP == FMA32) {
            REP32(asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(s[0]) : "v"(cs));
                  if (CHAINS > 1) { asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(s[1]) : "v"(cs)); asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(s[2]) : "v"(cs)); asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(s[3]) : "v"(cs)); })
        } else if (OP == PKMUL) {
            REP32(asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(p[0]) : "v"(c));
                  if (CHAINS > 1) { asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(p[1]) : "v"(c)); asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(p[2]) : "v"(c)); asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(p[3]) : "v"(c)); })
        } else if (OP == PKMUL_SWZ) {
            REP32(asm volatile("v_pk_mul_f32 %0, %0, %1 op_sel:[0,1] op_sel_hi:[1,0] neg_hi:[0,1]" : "+v"(p[0]) : "v"(c));
                  if (CHAINS > 1) { asm volatile("v_pk_mul_f32 %0, %0, %1 op_sel:[0,1] op_sel_hi:[1,0]" : "+v"(p[1]) : "v"(c)); asm volatile("v_pk_mul_f32 %0, %0, %1 op_sel:[0,1] op_sel_hi:[1,0]" : "+v"(p[2]) : "v"(c)); asm volatile("v_pk_mul_f32 %0, %0, %1 op_sel:[0,1] op_sel_hi:[1,0]" : "+v"(p[3]) : "v"(c)); })
        } else if (OP == PKADD) {
            REP32(asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(p[0]) : "v"(c));
                  if (CHAINS > 1) { asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(p[1]) : "v"(c)); asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(p[2]) : "v"(c)); asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(p[3]) : "v"(c)); })
        } else if (OP == PKFMA) {
            REP32(asm volatile("v_pk_fma_f32 %0, %0, %1, %1" : "+v"(p[0]) : "v"(c));
                  if (CHAINS > 1) { asm volatile("v_pk_fma_f32 %0, %0, %1, %1" : "+v"(p[1]) : "v"(c)); asm volatile("v_pk_fma_f32 %0, %0, %1, %1" : "+v"(p[2]) : "v"(c)); asm volatile("v_pk_fma_f32 %0, %0, %1, %1" : "+v"(p[3]) : "v"(c)); })
        } else if (OP == FMA64) {
            REP32(asm volatile("v_fma_f64 %0, %0, %1, %1" : "+v"(d[0]) : "v"(cd));
                  if (CHAINS > 1) { asm volatile("v_fma_f64 %0, %0, %1, %1" : "+v"(d[1]) : "v"(cd)); asm volatile("v_fma_f64 %0, %0, %1, %1" : "+v"(d[2]) : "v"(cd)); asm volatile("v_fma_f64 %0, %0, %1, %1" : "+v"(d[3]) : "v"(cd)); })
        } else if (OP == ADD64) {
            REP32(asm volatile("v_add_f64 %0, %0, %1" : "+v"(d[0]) : "v"(cd));
                  if (CHAINS > 1) { asm volatile("v_add_f64 %0, %0, %1" : "+v"(d[1]) : "v"(cd)); asm volatile("v_add_f64 %0, %0, %1" : "+v"(d[2]) : "v"(cd)); asm volatile("v_add_f64 %0, %0, %1" : "+v"(d[3]) : "v"(cd)); })
        } else if (OP == MUL64) {
            REP32(asm volatile("v_mul_f64 %0, %0, %1" : "+v"(d[0]) : "v"(cd));
                  if (CHAINS > 1) { asm volatile("v_mul_f64 %0, %0, %1" : "+v"(d[1]) : "v"(cd)); asm volatile("v_mul_f64 %0, %0, %1" : "+v"(d[2]) : "v"(cd)); asm volatile("v_mul_f64 %0, %0, %1" : "+v"(d[3]) : "v"(cd)); })
        } else if (OP == RCP32) {
            REP32(asm volatile("v_rcp_f32 %0, %0" : "+v"(s[0]));
                  if (CHAINS > 1) { asm volatile("v_rcp_f32 %0, %0" : "+v"(s[1])); asm volatile("v_rcp_f32 %0, %0" : "+v"(s[2])); asm volatile("v_rcp_f32 %0, %0" : "+v"(s[3])); })
        } else if (OP == SQRT32) {
            REP32(asm volatile("v_sqrt_f32 %0, %0" : "+v"(s[0]));
                  if (CHAINS > 1) { asm volatile("v_sqrt_f32 %0, %0" : "+v"(s[1])); asm volatile("v_sqrt_f32 %0, %0" : "+v"(s[2])); asm volatile("v_sqrt_f32 %0, %0" : "+v"(s[3])); })
        } else if (OP == CVT_F64_F32) {
            REP32(asm volatile("v_cvt_f32_f64 %0, %1" : "=v"(s[0]) : "v"(d[0]));
                  if (CHAINS > 1) { asm volatile("v_cvt_f32_f64 %0, %1" : "=v"(s[1]) : "v"(d[1])); asm volatile("v_cvt_f32_f64 %0, %1" : "=v"(s[2]) : "v"(d[2])); asm volatile("v_cvt_f32_f64 %0, %1" : "=v"(s[3]) : "v"(d[3])); })
        }
    }
    float acc = 0;
    for (int i = 0; i < 4; i++) acc += p[i].x + p[i].y + s[i] + (float)d[i];
    out[blockIdx.x * 256 + threadIdx.x] = acc;
}

static double g_clk_mhz = 2400.0;

template <int OP, int CHAINS>
void run(const char* name, int waves_per_simd) {
    const int blocks = 256 * waves_per_simd;  // a 256-thread workgroup = one wave on each SIMD of its CU
    float* out;
    hipMalloc(&out, (size_t)blocks * 256 * 4);
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    const int iters = 2000;
    k<OP, CHAINS><<<blocks, 256>>>(out, iters, 1.0f);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    k<OP, CHAINS><<<blocks, 256>>>(out, iters, 1.0f);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    const double insts_per_simd = (double)waves_per_simd * iters * 32 * CHAINS;  // wave-instructions issued by one SIMD
    const double cyc = ms * 1e-3 * g_clk_mhz * 1e6 / insts_per_simd;
    printf("%-14s chains %d  waves/SIMD %d : %8.3f ms  %6.2f cycles per wave-instruction per SIMD\n", name, CHAINS, waves_per_simd, ms, cyc);
    hipFree(out);
    hipEventDestroy(e0);
    hipEventDestroy(e1);
}

template <int OP>
void run_all(const char* name) {
    run<OP, 1>(name, 1);
    run<OP, 4>(name, 1);
    run<OP, 1>(name, 4);
    run<OP, 4>(name, 4);
}

int main() {
    hipDeviceProp_t prop;
    hipGetDeviceProperties(&prop, 0);
    g_clk_mhz = prop.clockRate / 1000.0;
    printf("device %s, %d CUs, clock %.0f MHz (cycles below assume this clock)\n", prop.name, prop.multiProcessorCount, g_clk_mhz);
    run_all<MUL32>("v_mul_f32");
    run_all<FMA32>("v_fma_f32");
    run_all<PKMUL>("v_pk_mul_f32");
    run_all<PKMUL_SWZ>("v_pk_mul swz");
    run_all<PKADD>("v_pk_add_f32");
    run_all<PKFMA>("v_pk_fma_f32");
    run_all<MUL64>("v_mul_f64");
    run_all<ADD64>("v_add_f64");
    run_all<FMA64>("v_fma_f64");
    run_all<RCP32>("v_rcp_f32");
    run_all<SQRT32>("v_sqrt_f32");
    run_all<CVT_F64_F32>("v_cvt_f32_f64");
    return 0;
}
