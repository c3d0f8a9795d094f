// cagym_ga3c.h -- fused GA3C-CADRL forward pass (policies/GA3C_CADRL/network.py:65-98 of the reference:
// input normalisation -> LSTM(64) over the observed agents -> concat host state -> FC 68->256 -> FC 256->256 ->
// FC 256->256 (all ReLU) -> logits 256->11 -> argmax -> action table (network.py:8-17) -> (pref_speed*a0, a1)).
//
// One workgroup (256 lanes) evaluates AG = 32 agents (16 for small batches, so that every CU gets work); three
// workgroups share a CU (49 KB of LDS each at AG = 32).  Lane n owns output neuron n of every layer for all 32
// agents (32 fp32 accumulators in registers); the layer input is kept in LDS as [k][agent] so one
// ds_read_b128 feeds four FMAs, and weight row k (256 floats, [in][out] as TensorFlow stores them) is one
// coalesced, L2-resident load per k.  fp32 like the reference's TF graph; the 683 KB of weights are shared
// by every workgroup.  ~0.67 MFLOP per agent: the fp32 vector rate and the fp32 MFMA rate are equal on gfx950,
// so plain FMAs are used.  TF1 LSTMCell conventions: gates (i, j, f, o), forget_bias 1.0, input = concat[x, h],
// state frozen beyond sequence_length.  "Parity unpinned" (TensorFlow absent): checked against an fp64 numpy
// restatement and the known answer of SURVEY.md 8(c).
#pragma once
#include "cagym_device.h"

#define GA_H 64
#define GA_W 256
// packed weight blob offsets (floats)
#define GA_OFF_WL 0
#define GA_OFF_BL (GA_OFF_WL + 71 * 256)
#define GA_OFF_W1 (GA_OFF_BL + 256)
#define GA_OFF_B1 (GA_OFF_W1 + 68 * 256)
#define GA_OFF_W2 (GA_OFF_B1 + 256)
#define GA_OFF_B2 (GA_OFF_W2 + 256 * 256)
#define GA_OFF_W3 (GA_OFF_B2 + 256)
#define GA_OFF_B3 (GA_OFF_W3 + 256 * 256)
#define GA_OFF_WP (GA_OFF_B3 + 256)
#define GA_OFF_BP (GA_OFF_WP + 256 * 11)
#define GA_NWEIGHTS (GA_OFF_BP + 11)

// acc[g] += in[k][g] * w for the 32 agents of the tile; `in` is LDS [K][32]
template <int GA_AG>
__device__ __forceinline__ void ga_dense(const float* __restrict__ in, const float* __restrict__ Wt, int K, int n,
                                         float (&acc)[GA_AG]) {
    for (int k = 0; k < K; k++) {
        const float w = Wt[(size_t)k * GA_W + n];
        const float4* row = reinterpret_cast<const float4*>(in + k * GA_AG);
#pragma unroll
        for (int q = 0; q < GA_AG / 4; q++) {
            const float4 v = row[q];
            acc[4 * q + 0] = fmaf(v.x, w, acc[4 * q + 0]);  // explicit: the TU is built with -ffp-contract=off
            acc[4 * q + 1] = fmaf(v.y, w, acc[4 * q + 1]);
            acc[4 * q + 2] = fmaf(v.z, w, acc[4 * q + 2]);
            acc[4 * q + 3] = fmaf(v.w, w, acc[4 * q + 3]);
        }
    }
}

__device__ __forceinline__ float ga_sigmoid(float x) { return 1.0f / (1.0f + expf(-x)); }

// state: [*, 76] rows (cagym_ga3c_state); agent_idx[B]: rows to evaluate (flat world*M + slot).
// ext_actions [N*M, 2] receives (pref_speed * a0, a1) for those agents; action_index / probs optional.
template <int GA_AG>
__global__ void __launch_bounds__(256) k_ga3c_forward(const float* __restrict__ Wb, const float* __restrict__ state,
                                                      const int32_t* __restrict__ agent_idx, int B,
                                                      const double* __restrict__ pref, float* ext_actions,
                                                      int32_t* action_index, float* probs) {
    // LDS: 49 KB per workgroup => three workgroups (12 waves) per CU.  The activations are updated IN PLACE (every
    // lane finishes reading the layer input before any lane writes its output row: one extra barrier per layer), the
    // other-agent features are re-read from the state rows each LSTM step, and the FC-1 input [host(4), h(64)] is
    // rows 3..70 of the LSTM input tile.
    __shared__ __attribute__((aligned(16))) float u[(7 + GA_H) * GA_AG];  // LSTM input [k][g]: 7 features + h
    __shared__ __attribute__((aligned(16))) float cst[GA_H * GA_AG];       // cell state [unit][g]
    __shared__ __attribute__((aligned(16))) float za[GA_W * GA_AG];        // gates / layer activations [n][g]
    __shared__ float hostv[4 * GA_AG];                                     // normalised host state [f][g]
    __shared__ int nseq[GA_AG];
    __shared__ int rowof[GA_AG];
    __shared__ float logit[GA_AG * 12];
    const int n = threadIdx.x, tile = blockIdx.x * GA_AG;
    // ---- load + normalise the host part (network.py:125-148): x_hat = (x - avg) / std -----------------------
    for (int e = n; e < GA_AG * 5; e += 256) {
        const int g = e / 5, f = e - g * 5;
        const int a = tile + g < B ? agent_idx[tile + g] : -1;
        const float x = a >= 0 ? state[(size_t)a * 76 + 1 + f] : 0.f;
        if (f == 0) {
            int ns = (int)x;
            nseq[g] = a >= 0 ? (ns < 0 ? 0 : (ns > 10 ? 10 : ns)) : 0;
            rowof[g] = a;
        } else {
            const float avg = f == 3 ? 1.0f : (f == 4 ? 0.5f : 0.0f);
            const float sd = f == 1 ? 5.0f : (f == 2 ? 3.14f : 1.0f);
            hostv[(f - 1) * GA_AG + g] = (x - avg) / sd;
        }
    }
    for (int e = n; e < GA_H * GA_AG; e += 256) {
        cst[e] = 0.f;
        u[7 * GA_AG + e] = 0.f;
    }
    __syncthreads();
    int tmax = 0;
    for (int g = 0; g < GA_AG; g++) tmax = nseq[g] > tmax ? nseq[g] : tmax;
    // ---- LSTM (network.py:83-90) -----------------------------------------------------------------------
    for (int t = 0; t < tmax; t++) {
        if (n < 7 * GA_AG) {  // other-agent features of step t, normalised: u[c][g] (7 * AG <= 256 lanes)
            const int c = n / GA_AG, g = n - c * GA_AG;
            const int a = rowof[g];
            const float x = a >= 0 ? state[(size_t)a * 76 + 6 + t * 7 + c] : 0.f;
            const float avg = c == 4 ? 0.5f : (c == 6 ? 1.0f : 0.0f);
            const float sd = (c == 0 || c == 1 || c == 5) ? 5.0f : 1.0f;
            u[n] = (x - avg) / sd;
        }
        __syncthreads();
        float acc[GA_AG];
        const float b = Wb[GA_OFF_BL + n];
#pragma unroll
        for (int g = 0; g < GA_AG; g++) acc[g] = b;
        ga_dense<GA_AG>(u, Wb + GA_OFF_WL, 7 + GA_H, n, acc);
        float4* zo = reinterpret_cast<float4*>(za + n * GA_AG);
#pragma unroll
        for (int q = 0; q < GA_AG / 4; q++) zo[q] = make_float4(acc[4 * q], acc[4 * q + 1], acc[4 * q + 2], acc[4 * q + 3]);
        __syncthreads();
        for (int e = n; e < GA_H * GA_AG; e += 256) {  // e = unit * 32 + g
            const int g = e & (GA_AG - 1);
            if (t < nseq[g]) {
                const float gi = za[e], gj = za[GA_H * GA_AG + e], gf = za[2 * GA_H * GA_AG + e], go = za[3 * GA_H * GA_AG + e];
                const float c = ga_sigmoid(gf + 1.0f) * cst[e] + ga_sigmoid(gi) * tanhf(gj);
                cst[e] = c;
                u[7 * GA_AG + e] = ga_sigmoid(go) * tanhf(c);
            }
        }
        __syncthreads();
    }
    // ---- layer1: concat[host(4), h(64)] -> 256, ReLU (network.py:92-93): input = rows 3..70 of u -----------
    for (int e = n; e < 4 * GA_AG; e += 256) u[3 * GA_AG + e] = hostv[e];
    __syncthreads();
    {
        float acc[GA_AG];
        const float b = Wb[GA_OFF_B1 + n];
#pragma unroll
        for (int g = 0; g < GA_AG; g++) acc[g] = b;
        ga_dense<GA_AG>(u + 3 * GA_AG, Wb + GA_OFF_W1, 4 + GA_H, n, acc);
        float4* zo = reinterpret_cast<float4*>(za + n * GA_AG);
#pragma unroll
        for (int q = 0; q < GA_AG / 4; q++)
            zo[q] = make_float4(fmaxf(acc[4 * q], 0.f), fmaxf(acc[4 * q + 1], 0.f), fmaxf(acc[4 * q + 2], 0.f), fmaxf(acc[4 * q + 3], 0.f));
    }
    __syncthreads();
    // ---- layer2, fullyconnected1 (network.py:95, 47), in place ------------------------------------------------
#pragma unroll 1
    for (int layer = 0; layer < 2; layer++) {
        float acc[GA_AG];
        const float b = Wb[(layer == 0 ? GA_OFF_B2 : GA_OFF_B3) + n];
#pragma unroll
        for (int g = 0; g < GA_AG; g++) acc[g] = b;
        ga_dense<GA_AG>(za, Wb + (layer == 0 ? GA_OFF_W2 : GA_OFF_W3), GA_W, n, acc);
        __syncthreads();  // every lane has read the whole input before any output row replaces it
        float4* zo = reinterpret_cast<float4*>(za + n * GA_AG);
#pragma unroll
        for (int q = 0; q < GA_AG / 4; q++)
            zo[q] = make_float4(fmaxf(acc[4 * q], 0.f), fmaxf(acc[4 * q + 1], 0.f), fmaxf(acc[4 * q + 2], 0.f), fmaxf(acc[4 * q + 3], 0.f));
        __syncthreads();
    }
    // ---- logits_p 256 -> 11 (network.py:50) -------------------------------------------------------------------
    for (int e = n; e < GA_AG * 11; e += 256) {
        const int g = e / 11, o = e - g * 11;
        float acc = Wb[GA_OFF_BP + o];
        for (int k = 0; k < GA_W; k++) acc = fmaf(za[k * GA_AG + g], Wb[GA_OFF_WP + k * 11 + o], acc);
        logit[g * 12 + o] = acc;
    }
    __syncthreads();
    // ---- softmax_p, argmax, action (network.py:51, GA3CCADRLPolicy.py:39-42) ----------------------------------
    if (n < GA_AG && tile + n < B) {
        const int a = agent_idx[tile + n];
        float mx = logit[n * 12];
        int best = 0;
        for (int o = 1; o < 11; o++)
            if (logit[n * 12 + o] > mx) { mx = logit[n * 12 + o]; best = o; }
        if (probs) {
            float ex[11], s = 0.f;
            for (int o = 0; o < 11; o++) { ex[o] = expf(logit[n * 12 + o] - mx); s += ex[o]; }
            for (int o = 0; o < 11; o++) probs[(size_t)(tile + n) * 11 + o] = (ex[o] / s + 1e-4f) / (1.0f + 1e-4f * 11);
        }
        if (action_index) action_index[tile + n] = best;
        // Actions table (network.py:14-17): rows 0-4 speed 1, dh = -pi/6 + k*pi/12; 5-7 speed .5; 8-10 speed 0
        double a0, a1;
        if (best < 5) { a0 = 1.0; a1 = -kPi / 6 + (double)best * (kPi / 12); }
        else if (best < 8) { a0 = 0.5; a1 = -kPi / 6 + (double)(best - 5) * (kPi / 6); }
        else { a0 = 0.0; a1 = -kPi / 6 + (double)(best - 8) * (kPi / 6); }
        if (ext_actions) {
            ext_actions[2 * (size_t)a] = (float)(pref[a] * a0);
            ext_actions[2 * (size_t)a + 1] = (float)a1;
        }
    }
}


// ---- the same forward pass on the matrix cores (round 2) ------------------------------------------------------------------
// v_mfma_f32_32x32x2_f32: exact fp32 products and sums, k in order - the fmaf chain the kernel above evaluates lane by lane, at
// the same peak rate (measured 147 TFLOP/s, tools/micro/mfma_rate.hip) but with ONE register operand per 32 x 32 x 2 block
// instead of a broadcast LDS read per 4 FMAs.  One workgroup = 32 agents (the M side of the tile); wave w owns 64 output
// neurons = two 32 x 32 accumulator tiles (2 x 16 VGPRs).  A operand: the layer input in LDS as [k][agent] (lane l reads row
// k0 + (l >> 5), agent l & 31: conflict-free ds_read_b32); B operand: weight W[k0 + (l >> 5)][column(l & 31)].
//   LSTM: the wave's 72 x 64 weight block is loaded ONCE into 72 VGPRs and reused by all (<= 10) steps; the columns are dealt
//   so that a wave holds all four gates of its 16 units (tile 0: i | j, tile 1: f | o): the cell update needs one
//   v_permlane16_swap per accumulator pair, the cell state of a (unit, agent) lives in a register for the whole sequence, h is
//   double-buffered in LDS (one barrier per step), the 10 x 7 sequence features are fetched once.
//   Dense layers: weights stream from L2 through a hand-pipelined loop (below).
// Layouts (MI355X guide): A[i = l & 31][k = l >> 5], B[k = l >> 5][j = l & 31], D reg r of lane l = row (r & 3) + 8 (r >> 2) +
// 4 (l >> 5), column l & 31.
typedef float ga_f32x16 __attribute__((ext_vector_type(16)));
// Diagnostic build only (-DCAGYM_STAMPS -DGA_STAMPS, tools/ga3c_phases.py): thread 0 of every workgroup adds the s_memtime
// ticks of the forward kernel's phases to g_stamps[0..5] and counts itself in g_stamps[15]
#if defined(CAGYM_STAMPS) && defined(GA_STAMPS)
#define GASTAMP_BEGIN() unsigned long long ga_prev = __builtin_amdgcn_s_memtime()
#define GASTAMP(i) do { if (threadIdx.x == 0) { unsigned long long _t = __builtin_amdgcn_s_memtime(); atomicAdd(&g_stamps[i], _t - ga_prev); ga_prev = _t; } } while (0)
#else
#define GASTAMP_BEGIN() do { } while (0)
#define GASTAMP(i) do { } while (0)
#endif
#ifndef GA_DENSE_U
#define GA_DENSE_U 8  /* k-pairs per software-pipeline stage of the dense layers */
#endif

// One dense layer's accumulation for this wave's two tiles: acc += in[32 agents x K] * Wt[K x columns c0 / c1].  The weights come
// from L2 (every workgroup reads the same 0.7 MB), so the loop is software-pipelined by hand over two named register sets
// (no copies: the compiler would coalesce them and serialise load -> use): while the 2 U matrix instructions of one block
// issue (U = 8: 1024 matrix-core cycles), the U k-pairs of the next block are in flight.  Loads are unconditional (row index
// clamped, the A operand zeroed beyond K), so the loop body has no branches.  TILES = 1: only c0 / acc0.
template <int U, int TILES>
struct GaOperands {
    float a[U], b0[U], b1[U];
};
// K (even: 68, 256, 64) is a template argument.  Stages of U k-pairs (rows 2 kp, 2 kp + 1) in a rolled loop over PAIRS of stages;
// the k-pairs beyond the last full pair of stages (layer1: 2 of 34) follow as straight-line code, so that no stage is padded:
// the row of every load is UNIFORM - a weight address is a scalar row pointer + the lane's fixed offset (half * ldw + column),
// an A operand a fixed LDS offset from the stage's base - no vector arithmetic per load, no select on the A operand.
// (Round 2 clamped the row per lane: ~6 vector instructions per load, 100+ per stage, all of them issued BEFORE the stage's
// matrix instructions because of the scheduling fences: a dense stage took 1 690 cycles for 1 024 cycles of matrix work,
// tools/ga3c_phases.py.)
template <int U, int TILES>
__device__ __forceinline__ void ga_fetch(GaOperands<U, TILES>& o, const float* __restrict__ ain_blk, const float* __restrict__ wrow_blk, int ldw,
                                         uint32_t boff0, uint32_t boff1) {
#pragma unroll
    for (int i = 0; i < U; i++) {
        const float* row = wrow_blk + (size_t)(2 * i) * ldw;
#ifdef GA_DIAG_NOA  // diagnostic builds (tools/ga3c_phases.py): the layer without its LDS reads / without its weight loads
        o.a[i] = __uint_as_float(boff0 + i);
#else
        o.a[i] = ain_blk[i * 64];
#endif
#ifdef GA_DIAG_NOB
        o.b0[i] = __uint_as_float(boff0 + 2 * i);
        if (TILES == 2) o.b1[i] = __uint_as_float(boff1 + 2 * i);
#else
        o.b0[i] = row[boff0];
        if (TILES == 2) o.b1[i] = row[boff1];
#endif
    }
}
template <int U, int TILES>
__device__ __forceinline__ void ga_issue(const GaOperands<U, TILES>& o, bool use0, ga_f32x16& acc0, ga_f32x16& acc1) {
#pragma unroll
    for (int i = 0; i < U; i++) {
        acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(o.a[i], TILES == 2 ? o.b0[i] : (use0 ? o.b0[i] : 0.f), acc0, 0, 0, 0);
        if (TILES == 2) acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(o.a[i], o.b1[i], acc1, 0, 0, 0);
    }
}
// the order the scheduler is asked for inside one stage: the next stage's loads go between the matrix instructions of the FIRST
// half of the stage, two k-pairs' worth per gap: they issue while the matrix core is busy, and the youngest of them is half a
// stage old when the stage ends (the wait-count pass puts s_waitcnt vmcnt(0) at the top of a rolled loop).
template <int U, int TILES>
__device__ __forceinline__ void ga_stage_order() {
#ifndef GA_NO_INTERLEAVE
#pragma unroll
    for (int i = 0; i < U; i++) {
        __builtin_amdgcn_sched_group_barrier(0x008, TILES, 0);  // MFMA
        if (i < U / 2) {
            __builtin_amdgcn_sched_group_barrier(0x020, 2 * TILES, 0);  // VMEM read
            __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);          // DS read
        }
    }
#endif
}
template <int U, int TILES, int K>
__device__ __forceinline__ void ga_mfma_layer(const float* __restrict__ in, const float* __restrict__ Wt, int ldw, int c0, int c1,
                                              bool use0, ga_f32x16& acc0, ga_f32x16& acc1) {
    static_assert(K % 2 == 0 && U % 2 == 0, "k-pairs, two half stages");
    constexpr int KP = K / 2, NST = KP / (2 * U) * 2, REM = KP - NST * U;  // full stages (an even number), k-pairs behind them
    const int lane = threadIdx.x & 63, half = lane >> 5, j = lane & 31;
    const float* ain = in + half * 32 + j;  // A operand of k-pair kp: row 2 kp + half, agent j
    const uint32_t boff0 = (uint32_t)(half * ldw + c0), boff1 = (uint32_t)(half * ldw + c1);
    const size_t wst = (size_t)(2 * U) * ldw;  // weight words per stage
    if (NST > 0) {
        // two named register sets (no copies: the compiler would coalesce them and serialise load -> use); the fences keep a
        // stage's loads from sinking to just before their use in the NEXT stage
        GaOperands<U, TILES> A, Bq;
        ga_fetch<U, TILES>(A, ain, Wt, ldw, boff0, boff1);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll 1
        for (int blk = 0; blk < NST; blk += 2) {
            ga_fetch<U, TILES>(Bq, ain + (blk + 1) * U * 64, Wt + (blk + 1) * wst, ldw, boff0, boff1);
            ga_issue<U, TILES>(A, use0, acc0, acc1);
            ga_stage_order<U, TILES>();
            __builtin_amdgcn_sched_barrier(0);
            const int nb = blk + 2 < NST ? blk + 2 : NST - 1;  // behind the last stage: any valid stage again, never used
            ga_fetch<U, TILES>(A, ain + nb * U * 64, Wt + nb * wst, ldw, boff0, boff1);
            ga_issue<U, TILES>(Bq, use0, acc0, acc1);
            ga_stage_order<U, TILES>();
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    if (REM > 0) {
        GaOperands<(REM > 0 ? REM : 1), TILES> R;
        ga_fetch<(REM > 0 ? REM : 1), TILES>(R, ain + NST * U * 64, Wt + NST * wst, ldw, boff0, boff1);
        ga_issue<(REM > 0 ? REM : 1), TILES>(R, use0, acc0, acc1);
    }
}

// gate non-linearities on the transcendental unit (v_exp_f32 / v_rcp_f32, ~1 ulp each): the cell update of 2048 (unit, agent) cells
// per step would otherwise cost more issue cycles than the step's 72 matrix instructions.  TensorFlow's own fp32 kernels are
// rational approximations of the same accuracy class; the network is "parity unpinned"  and checked to 1e-4 on probabilities.
__device__ __forceinline__ float ga_fast_sigmoid(float x) { return __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(-1.44269504f * x)); }
__device__ __forceinline__ float ga_fast_tanh(float x) { return 1.0f - 2.0f * __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(2.88539008f * x)); }

__device__ __forceinline__ ga_f32x16 ga_splat(float v) {
    ga_f32x16 r;
#pragma unroll
    for (int i = 0; i < 16; i++) r[i] = v;
    return r;
}

// store one accumulator tile (ReLU applied) as rows [neuron][agent] of the activation buffer: reg r <-> agent (r & 3) +
// 8 (r >> 2) + 4 half, four consecutive agents per 16-byte store
__device__ __forceinline__ void ga_store_tile_relu(float* act, int neuron, const ga_f32x16& acc) {
    const int half = (threadIdx.x & 63) >> 5;
    float4* row = reinterpret_cast<float4*>(act + neuron * 32);
#pragma unroll
    for (int q = 0; q < 4; q++)
        row[2 * q + half] = make_float4(fmaxf(acc[4 * q], 0.f), fmaxf(acc[4 * q + 1], 0.f), fmaxf(acc[4 * q + 2], 0.f), fmaxf(acc[4 * q + 3], 0.f));
}

// 200 VGPRs (72 of them the LSTM weights) -> 2 workgroups per CU; forcing 3 spills weights and measures the same
__global__ void __launch_bounds__(256, 2) k_ga3c_forward_mfma(const float* __restrict__ Wb, const float* __restrict__ state,
                                                           const int32_t* __restrict__ agent_idx, int B_host,
                                                           const int32_t* __restrict__ B_dev, const double* __restrict__ pref,
                                                           float* ext_actions, int32_t* action_index, float* probs, uint32_t* list_ctr) {
    const int B = B_dev ? *B_dev : B_host;  // device-side count (cagym_ga3c_act): the grid covers the worst case
    // the last kernel of cagym_ga3c_act's chain starts the next list where this one ended (k_ga3c_select; nobody reads these two words now)
    if (list_ctr && blockIdx.x == 0 && threadIdx.x == 0) list_ctr[1] = list_ctr[0];
    if ((int)blockIdx.x * 32 >= B) return;
    GASTAMP_BEGIN();
    constexpr int AG = 32, HB = (4 + GA_H + 1) * AG;
    // One 40.6 KB LDS block (3 workgroups per CU):
    //   hb0 | hb1: [4 host features | 64 hidden | one zero row][agent], double-buffered over the LSTM steps (one barrier per step);
    //              the sequence starts in buffer tmax & 1 so that it always ends in hb0, which is layer1's input as it stands;
    //   xs:        the normalised features of all 10 sequence slots, fetched once;
    //   za:        layer activations [neuron][agent], over hb1 and xs (both dead when layer1 writes its output);
    //   part, logit: over hb0 (dead once layer1 has been read).
    __shared__ __attribute__((aligned(16))) float lds[HB + GA_W * AG];
    __shared__ int nseq[AG];
    float* hb = lds;
    float* xs = lds + 2 * HB;
    float* za = lds + HB;
    float* part = lds;                 // logits: partial sums of the four waves [4][AG][12]
    float* logit = lds + 4 * AG * 12;  // [AG][12]
    static_assert(2 * HB + 10 * 7 * AG <= HB + GA_W * AG && 5 * AG * 12 <= HB, "LDS aliasing");
    const int n = threadIdx.x, tile = blockIdx.x * AG;
    const int lane = n & 63, wave = n >> 6, half = lane >> 5, j = lane & 31;
    // the agent behind each of the tile's rows and its preferred speed, requested now for the action write at the very end (two
    // dependent loads from HBM that used to sit behind the last layer: ~1 us of the kernel's tail)
    int act_a = 0;
    double act_pref = 0.0;
    if (n < AG && tile + n < B) {
        act_a = agent_idx[tile + n];
        act_pref = pref[act_a];
    }
    // list_ctr != null (cagym_ga3c_act): the state rows are stored by place in the list - row tile + g, no index look-up in front
    const bool by_place = list_ctr != nullptr;
    for (int e = n; e < AG * 5; e += 256) {
        const int g = e / 5, f = e - g * 5;
        const int a = tile + g < B ? (by_place ? tile + g : agent_idx[tile + g]) : -1;
        const float x = a >= 0 ? state[(size_t)a * 76 + 1 + f] : 0.f;
        if (f == 0) {
            int ns = (int)x;
            nseq[g] = a >= 0 ? (ns < 0 ? 0 : (ns > 10 ? 10 : ns)) : 0;
        } else {
            const float avg = f == 3 ? 1.0f : (f == 4 ? 0.5f : 0.0f);
            const float sd = f == 1 ? 5.0f : (f == 2 ? 3.14f : 1.0f);
            hb[(f - 1) * AG + g] = hb[HB + (f - 1) * AG + g] = (x - avg) / sd;
        }
    }
    for (int e = n; e < 10 * 7 * AG; e += 256) {
        const int tc = e / AG, g = e - tc * AG, c = tc % 7;
        const int a = tile + g < B ? (by_place ? tile + g : agent_idx[tile + g]) : -1;
        const float x = a >= 0 ? state[(size_t)a * 76 + 6 + tc] : 0.f;
        const float avg = c == 4 ? 0.5f : (c == 6 ? 1.0f : 0.0f);
        const float sd = (c == 0 || c == 1 || c == 5) ? 5.0f : 1.0f;
        xs[e] = (x - avg) / sd;
    }
    for (int e = n; e < 65 * AG; e += 256) hb[4 * AG + e] = hb[HB + 4 * AG + e] = 0.f;  // h = 0 and the padding rows
    // ---- LSTM: this lane's (unit, agent) cells: unit = 16 wave + (j & 15), agents = regs 8 (j >> 4) .. + 7 of its half ----
    const int unit = 16 * wave + (j & 15);
    const int rbase = 8 * (j >> 4);  // lanes j < 16 finish regs 0..7, their partners (j >= 16) regs 8..15
    float cst[8];  // cell states; the hidden state of a finished sequence is carried over from the current h buffer
#pragma unroll
    for (int q = 0; q < 8; q++) cst[q] = 0.f;
    const int lc0 = (j < 16 ? 0 : 64) + 16 * wave + (j & 15);     // gates i | j
    const int lc1 = (j < 16 ? 128 : 192) + 16 * wave + (j & 15);  // gates f | o
    const float bl0 = Wb[GA_OFF_BL + lc0], bl1 = Wb[GA_OFF_BL + lc1];
    // the wave's 72 x 64 block of the LSTM kernel stays in registers for the whole sequence (72 VGPRs, dead afterwards)
    constexpr int LKP = (7 + GA_H + 1) / 2;
    float wl0[LKP], wl1[LKP];
#pragma unroll
    for (int p = 0; p < LKP; p++) {
        const int k = 2 * p + half;
        wl0[p] = k < 7 + GA_H ? Wb[GA_OFF_WL + (size_t)k * GA_W + lc0] : 0.f;
        wl1[p] = k < 7 + GA_H ? Wb[GA_OFF_WL + (size_t)k * GA_W + lc1] : 0.f;
    }
    __syncthreads();
    GASTAMP(0);
    int tmax = 0;
    for (int g = 0; g < AG; g++) tmax = nseq[g] > tmax ? nseq[g] : tmax;
    uint32_t live_until = 0;  // sequence lengths (<= 10) of this lane's eight agents, 4 bits each
#pragma unroll
    for (int q = 0; q < 8; q++) {
        const int r = rbase + q;
        live_until |= (uint32_t)nseq[(r & 3) + 8 * (r >> 2) + 4 * half] << (4 * q);
    }
    for (int t = 0; t < tmax; t++) {
        const float* hcur = hb + ((t + tmax) & 1) * HB;
        float* hnext = hb + ((t + tmax + 1) & 1) * HB;
        const float* xt = xs + t * 7 * AG;
        ga_f32x16 a0 = ga_splat(bl0), a1 = ga_splat(bl1);
        // (Round 3 measured a split of the step - tile 0 first, i * j of the eight cells between tile 1's matrix instructions,
        // with one and with two accumulation chains per tile -: the LSTM part 66 000 -> 73 000 cycles per workgroup both times,
        // profiles/r3/ga3c_forward_phases.txt.  Interleaving vector work into the wave's own matrix stream does not pay here.)
#pragma unroll
        for (int p = 0; p < LKP; p++) {
            // input row k = 2 p + half of concat[x_t (7), h (64), 0]: rows 0..6 from xs, row k >= 7 is row k - 3 of the h buffer
            float a;
            if (p < 3) a = xt[(2 * p + half) * AG + j];
            else if (p == 3) a = half ? hcur[4 * AG + j] : xt[6 * AG + j];
            else a = hcur[(2 * p + half - 3) * AG + j];
            a0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a, wl0[p], a0, 0, 0, 0);
            a1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a, wl1[p], a1, 0, 0, 0);
        }
        // lane j < 16 holds (i, f) of its unit and needs (j, o) from lane j + 16 for regs 0..7; lane j >= 16 holds (j, o) and
        // needs (i, f) from lane j - 16 for regs 8..15
#pragma unroll
        for (int q = 0; q < 8; q++) {
            // v_permlane16_swap (gfx950): rows 1 / 3 of the first operand <-> rows 0 / 2 of the second, i.e. exactly this exchange:
            // afterwards the first result holds gate i (f) and the second gate j (o) of the lane's own (unit, agent) cell
            const auto s0 = __builtin_amdgcn_permlane16_swap(__float_as_uint(a0[q]), __float_as_uint(a0[8 + q]), false, false);
            const auto s1 = __builtin_amdgcn_permlane16_swap(__float_as_uint(a1[q]), __float_as_uint(a1[8 + q]), false, false);
            const float gi = __uint_as_float(s0[0]), gj = __uint_as_float(s0[1]);
            const float gf = __uint_as_float(s1[0]), go = __uint_as_float(s1[1]);
            const int r = rbase + q;
            const int agent = (r & 3) + 8 * (r >> 2) + 4 * half;
            float hnew = hcur[(4 + unit) * AG + agent];
            if (t < (int)((live_until >> (4 * q)) & 15u)) {
                const float c = ga_fast_sigmoid(gf + 1.0f) * cst[q] + ga_fast_sigmoid(gi) * ga_fast_tanh(gj);
                cst[q] = c;
                hnew = ga_fast_sigmoid(go) * ga_fast_tanh(c);
            }
            hnext[(4 + unit) * AG + agent] = hnew;
        }
        __syncthreads();  // the other buffer is complete; everyone has finished reading this one
    }
    // ---- layer1: concat[host(4), h(64)] -> 256, ReLU: the final h buffer as it stands -------------------------------------
#ifdef GA_STAGGER
    for (int i = 0; i < (int)(blockIdx.x & 7); i++) __builtin_amdgcn_s_sleep(GA_STAGGER);
#endif
    GASTAMP(1);
    const float* hfin = hb;
    const int c0 = 64 * wave + j, c1 = 64 * wave + 32 + j;
    {
        ga_f32x16 a0 = ga_splat(Wb[GA_OFF_B1 + c0]), a1 = ga_splat(Wb[GA_OFF_B1 + c1]);
        ga_mfma_layer<GA_DENSE_U, 2, 4 + GA_H>(hfin, Wb + GA_OFF_W1, GA_W, c0, c1, true, a0, a1);
        ga_store_tile_relu(za, c0, a0);
        ga_store_tile_relu(za, c1, a1);
    }
    __syncthreads();
    GASTAMP(2);
#pragma unroll 1
    for (int layer = 0; layer < 2; layer++) {
        const int ob = layer == 0 ? GA_OFF_B2 : GA_OFF_B3;
        ga_f32x16 a0 = ga_splat(Wb[ob + c0]), a1 = ga_splat(Wb[ob + c1]);
        ga_mfma_layer<GA_DENSE_U, 2, GA_W>(za, Wb + (layer == 0 ? GA_OFF_W2 : GA_OFF_W3), GA_W, c0, c1, true, a0, a1);
        __syncthreads();  // in place: every wave has read the whole input
        ga_store_tile_relu(za, c0, a0);
        ga_store_tile_relu(za, c1, a1);
        __syncthreads();
        GASTAMP(3 + layer);
    }
    // ---- logits_p 256 -> 11: each wave sums its quarter of k on one tile (columns >= 11 are zero weights) ---------------------
    {
        ga_f32x16 a0 = ga_splat(0.f), a1 = ga_splat(0.f);
        ga_mfma_layer<8, 1, 64>(za + 64 * wave * AG, Wb + GA_OFF_WP + (size_t)64 * __builtin_amdgcn_readfirstlane(wave) * 11, 11, j < 11 ? j : 0, 0, j < 11, a0, a1);
        if (j < 11) {
#pragma unroll
            for (int r = 0; r < 16; r++) part[(wave * AG + (r & 3) + 8 * (r >> 2) + 4 * half) * 12 + j] = a0[r];
        }
    }
    __syncthreads();
    for (int e = n; e < AG * 11; e += 256) {
        const int g = e / 11, o = e - g * 11;
        logit[g * 12 + o] = Wb[GA_OFF_BP + o] + ((part[g * 12 + o] + part[(AG + g) * 12 + o]) + (part[(2 * AG + g) * 12 + o] + part[(3 * AG + g) * 12 + o]));
    }
    __syncthreads();
    if (n < AG && tile + n < B) {
        const int a = act_a;
        float mx = logit[n * 12];
        int best = 0;
        for (int o = 1; o < 11; o++)
            if (logit[n * 12 + o] > mx) { mx = logit[n * 12 + o]; best = o; }
        if (probs) {
            float ex[11], s = 0.f;
            for (int o = 0; o < 11; o++) { ex[o] = expf(logit[n * 12 + o] - mx); s += ex[o]; }
            for (int o = 0; o < 11; o++) probs[(size_t)(tile + n) * 11 + o] = (ex[o] / s + 1e-4f) / (1.0f + 1e-4f * 11);
        }
        if (action_index) action_index[tile + n] = best;
        double a0, a1;
        if (best < 5) { a0 = 1.0; a1 = -kPi / 6 + (double)best * (kPi / 12); }
        else if (best < 8) { a0 = 0.5; a1 = -kPi / 6 + (double)(best - 5) * (kPi / 6); }
        else { a0 = 0.0; a1 = -kPi / 6 + (double)(best - 8) * (kPi / 6); }
        if (ext_actions) {
            ext_actions[2 * (size_t)a] = (float)(act_pref * a0);
            ext_actions[2 * (size_t)a + 1] = (float)a1;
        }
    }
    GASTAMP(5);
#if defined(CAGYM_STAMPS) && defined(GA_STAMPS)
    if (threadIdx.x == 0) atomicAdd(&g_stamps[15], 1ull);
#endif
}
