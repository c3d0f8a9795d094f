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
