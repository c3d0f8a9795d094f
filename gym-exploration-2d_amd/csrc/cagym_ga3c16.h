// cagym_ga3c16.h -- the GA3C-CADRL forward pass (policies/GA3C_CADRL/network.py:65-98) on gfx950's 16-bit matrix cores with
// SPLIT fp32 operands (round 4).
//
// Why.  v_mfma_f32_32x32x2_f32 (cagym_ga3c.h) runs at the fp32 vector rate: 64 cycles for 4 096 flops.  v_mfma_f32_32x32x16_f16
// does 32 768 flops in 32 cycles - 16 x - but takes 11-bit operands.  Every fp32 operand v is therefore handed over as TWO halves
//     hi = f16(v),   lo = f16((v - hi) * 2^11)          (v - hi is exact in fp32; hi + lo 2^-11 carries 22 - 23 bits of v)
// and a product is three matrix instructions instead of one:
//     main  += hi_w * hi_x                               (products of 11-bit numbers are exact in the fp32 accumulator)
//     cross += hi_w * lo_x + lo_w * hi_x                 (scaled by 2^11, its own accumulator: no small term is rounded against a big one)
//     result = main + cross * 2^-11                      (the dropped lo * lo term is 2^-22 of the product)
// = 5.3 x the fp32 matrix rate at fp32-class accuracy: a product is accurate to ~2^-22 relative, the same size as the rounding
// an fp32 fmaf chain of K = 256 accumulates; against the fp64 restatement (oracle/ga3c_ref.py) the probabilities of this kernel
// and of the fp32 kernels differ by the same few 1e-6 (tests/test_ga3c.py prints both).  The scaling keeps `lo` a normal f16
// number whatever the magnitude of v (no reliance on f16 subnormals).  Activations are clamped to the f16 range (65 504) before
// the split; the network's are below 100.
//
// Orientation.  D[neuron][agent] = W^T[neuron][k] * X[k][agent]: the WEIGHTS are the A operand (row = output neuron), the layer
// input the B operand (column = agent).  A lane of the 32 x 32 result then holds ONE agent (its column) and 16 neurons (its
// registers) - and the B operand of the next layer's K-step wants, per lane, 8 consecutive k of ONE agent: exactly registers
// 8 s' .. 8 s' + 7 of the result, if the K-slots of a step are dealt the way the accumulator deals its rows:
//     slot (s, half, e)  <->  k = 16 s + (e & 3) + 8 (e >> 2) + 4 half        (dense layers)
// so a wave converts its 16 results to f16 halves and stores them as the next layer's operand fragments with four 16-byte LDS
// writes; readers fetch a fragment with one ds_read_b128, conflict-free, no transposes anywhere.  The weights are packed once per
// blob (k_ga3c_pack16) in exactly the order a lane consumes them: every weight load is one coalesced 1 KB global_load_dwordx4.
// LSTM: wave w owns units 8 w .. 8 w + 7 with all four gates (tile row 8 g + u), so a lane holds (i, j, f, o) of four (unit, agent)
// cells in its own registers - no lane exchange at all - keeps their cell states in registers for the whole sequence and writes
// its four h values (8 bytes of hi, 8 of lo) straight into the K-slot  (s, half', e) <-> unit 16 s + 8 half' + e  of the next step.
//
// One workgroup = 32 agents x 8 waves (512 lanes): two waves per SIMD, so that one wave's cell update (vector work) runs under
// the other's matrix instructions; the fp32 kernel's one wave per SIMD left either the matrix core or the vector unit idle.
#pragma once
#include "cagym_ga3c.h"

typedef _Float16 ga_h8 __attribute__((ext_vector_type(8)));

// packed blob (bytes).  A "fragment" is 64 lanes x 16 B = 1 KB; a K-step of a wave is [hi fragment | lo fragment] = 2 KB.
#define GA16_KSTEP 2048
#define GA16_OFF_LSTM 0                                   /* [wave 8][K-step 5] */
#define GA16_OFF_L1 (GA16_OFF_LSTM + 8 * 5 * GA16_KSTEP)  /* [wave 8][K-step 5] */
#define GA16_OFF_L2 (GA16_OFF_L1 + 8 * 5 * GA16_KSTEP)    /* [wave 8][K-step 16] */
#define GA16_OFF_L3 (GA16_OFF_L2 + 8 * 16 * GA16_KSTEP)
#define GA16_OFF_LOG (GA16_OFF_L3 + 8 * 16 * GA16_KSTEP)  /* [K-step 16] */
#define GA16_OFF_BIAS (GA16_OFF_LOG + 16 * GA16_KSTEP)    /* fp32: [layer 4][wave 8][half 2][reg 16], then the 11 logit biases */
#define GA16_PACKED_BYTES (GA16_OFF_BIAS + (4 * 256 + 16) * 4)
#define GA16_SC 2048.0f
#define GA16_ISC (1.0f / 2048.0f)

union GaU4 {
    uint4 u;
    ga_h8 h;
    _Float16 f[8];
};
union GaU2 {
    uint2 u;
    _Float16 f[4];
};

__device__ __forceinline__ void ga16_split(float v, _Float16& hi, _Float16& lo) {
    hi = (_Float16)v;
    lo = (_Float16)((v - (float)hi) * GA16_SC);  // both operations exact
}

// ---- packing: TensorFlow's [in][out] fp32 matrices -> the fragments the forward kernel's lanes consume ----------------------
// weight of (layer, wave, K-step s, lane, element e); the K-slot rules of the header comment
__device__ __forceinline__ float ga16_weight(const float* __restrict__ Wb, int layer, int w, int s, int lane, int e) {
    const int i = lane & 31, hp = lane >> 5;
    if (layer == 0) {  // LSTM kernel [7 + 64][4 x 64], gate order i, j, f, o: tile row i = 8 gate + unit
        const int col = (i >> 3) * 64 + 8 * w + (i & 7);
        int k;
        if (s < 4) k = 7 + 16 * s + 8 * hp + e;  // h part: K-slot <-> unit 16 s + 8 half + e
        else if (hp == 0 && e < 7) k = e;        // the observed agent's 7 features
        else return 0.f;
        return Wb[GA_OFF_WL + (size_t)k * GA_W + col];
    }
    if (layer == 1) {  // layer1 kernel [4 + 64][256]: input = concat[host, h]
        int k;
        if (s < 4) k = 4 + 16 * s + 8 * hp + e;
        else if (hp == 0 && e < 4) k = e;
        else return 0.f;
        return Wb[GA_OFF_W1 + (size_t)k * GA_W + 32 * w + i];
    }
    const int k = 16 * s + (e & 3) + 8 * (e >> 2) + 4 * hp;
    if (layer == 2) return Wb[GA_OFF_W2 + (size_t)k * GA_W + 32 * w + i];
    if (layer == 3) return Wb[GA_OFF_W3 + (size_t)k * GA_W + 32 * w + i];
    return i < 11 ? Wb[GA_OFF_WP + (size_t)k * 11 + i] : 0.f;  // logits_p [256][11]: rows 11..31 of the tile are zero
}

// one thread per (layer, wave, K-step, lane); then the biases in accumulator order: reg r of (wave, half) <-> tile row
// (r & 3) + 8 (r >> 2) + 4 half
__global__ void __launch_bounds__(256) k_ga3c_pack16(const float* __restrict__ Wb, unsigned char* __restrict__ P) {
    const int NF0 = 8 * 5 * 64, NF2 = 8 * 16 * 64, NFL = 16 * 64, NF = 2 * NF0 + 2 * NF2 + NFL;
    int id = blockIdx.x * blockDim.x + threadIdx.x;
    if (id < NF) {
        int layer, off, ks;
        if (id < NF0) { layer = 0; off = GA16_OFF_LSTM; ks = 5; }
        else if (id < 2 * NF0) { layer = 1; off = GA16_OFF_L1; ks = 5; id -= NF0; }
        else if (id < 2 * NF0 + NF2) { layer = 2; off = GA16_OFF_L2; ks = 16; id -= 2 * NF0; }
        else if (id < 2 * NF0 + 2 * NF2) { layer = 3; off = GA16_OFF_L3; ks = 16; id -= 2 * NF0 + NF2; }
        else { layer = 4; off = GA16_OFF_LOG; ks = 16; id -= 2 * NF0 + 2 * NF2; }
        const int lane = id & 63, s = (id >> 6) % ks, w = (id >> 6) / ks;
        GaU4 hi, lo;
#pragma unroll
        for (int e = 0; e < 8; e++) ga16_split(ga16_weight(Wb, layer, w, s, lane, e), hi.f[e], lo.f[e]);
        unsigned char* dst = P + off + (size_t)(w * ks + s) * GA16_KSTEP + lane * 16;
        *reinterpret_cast<uint4*>(dst) = hi.u;
        *reinterpret_cast<uint4*>(dst + 1024) = lo.u;
        return;
    }
    id -= NF;
    float* bias = reinterpret_cast<float*>(P + GA16_OFF_BIAS);
    if (id < 4 * 256) {
        const int layer = id >> 8, w = (id >> 5) & 7, half = (id >> 4) & 1, r = id & 15;
        const int row = (r & 3) + 8 * (r >> 2) + 4 * half;
        float b;
        if (layer == 0) b = Wb[GA_OFF_BL + (row >> 3) * 64 + 8 * w + (row & 7)];
        else b = Wb[(layer == 1 ? GA_OFF_B1 : layer == 2 ? GA_OFF_B2 : GA_OFF_B3) + 32 * w + row];
        bias[id] = b;
    } else if (id < 4 * 256 + 16) {
        bias[id] = id - 4 * 256 < 11 ? Wb[GA_OFF_BP + id - 4 * 256] : 0.f;
    }
}
#define GA16_PACK_THREADS (2 * 8 * 5 * 64 + 2 * 8 * 16 * 64 + 16 * 64 + 4 * 256 + 16)

// ---- forward -------------------------------------------------------------------------------------------------------------------
__device__ __forceinline__ ga_h8 ga16_ld(const unsigned char* p) { return *reinterpret_cast<const ga_h8*>(p); }
__device__ __forceinline__ ga_h8 ga16_ldg(const uint4* p) {
    GaU4 x;
    x.u = *p;
    return x.h;
}
// acc += W^T X for one K-step: three matrix instructions (header comment)
__device__ __forceinline__ void ga16_mac(const ga_h8& ah, const ga_h8& al, const ga_h8& bh, const ga_h8& bl, ga_f32x16& m, ga_f32x16& c) {
    m = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bh, m, 0, 0, 0);
    c = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bl, c, 0, 0, 0);
    c = __builtin_amdgcn_mfma_f32_32x32x16_f16(al, bh, c, 0, 0, 0);
}
__device__ __forceinline__ ga_f32x16 ga16_bias(const unsigned char* __restrict__ P, int layer, int wave, int half) {
    const float4* b = reinterpret_cast<const float4*>(P + GA16_OFF_BIAS) + ((layer * 8 + wave) * 2 + half) * 4;
    ga_f32x16 r;
#pragma unroll
    for (int q = 0; q < 4; q++) {
        const float4 v = b[q];
        r[4 * q] = v.x; r[4 * q + 1] = v.y; r[4 * q + 2] = v.z; r[4 * q + 3] = v.w;
    }
    return r;
}
// ReLU(main + cross 2^-11) of the wave's 32 neurons -> the next layer's operand fragments: registers 8 s' .. 8 s' + 7 of this lane ARE
// fragment (K-step 2 wave + s', this lane's half, this lane's agent)
__device__ __forceinline__ void ga16_store_act(unsigned char* act, int wave, int lane, const ga_f32x16& m, const ga_f32x16& c) {
#pragma unroll
    for (int sp = 0; sp < 2; sp++) {
        GaU4 hi, lo;
#pragma unroll
        for (int e = 0; e < 8; e++) {
            const float v = fminf(fmaxf(fmaf(c[8 * sp + e], GA16_ISC, m[8 * sp + e]), 0.f), 65504.f);
            ga16_split(v, hi.f[e], lo.f[e]);
        }
        unsigned char* dst = act + (2 * wave + sp) * GA16_KSTEP + lane * 16;
        *reinterpret_cast<uint4*>(dst) = hi.u;
        *reinterpret_cast<uint4*>(dst + 1024) = lo.u;
    }
}
// a 256-input dense layer of this wave's 32 neurons: 16 K-steps, the weights streamed from L2 in stages of four K-steps (8 KB per
// wave), two stages in flight while a third is multiplied
struct Ga16Stage {
    ga_h8 ah[4], al[4];
};
__device__ __forceinline__ void ga16_fetch(Ga16Stage& S, const uint4* __restrict__ PA, int st) {
#pragma unroll
    for (int i = 0; i < 4; i++) {
        S.ah[i] = ga16_ldg(PA + (st * 4 + i) * 128);
        S.al[i] = ga16_ldg(PA + (st * 4 + i) * 128 + 64);
    }
}
__device__ __forceinline__ void ga16_issue(const Ga16Stage& S, const unsigned char* actl, int st, ga_f32x16& m, ga_f32x16& c) {
#pragma unroll
    for (int i = 0; i < 4; i++) {
        const ga_h8 bh = ga16_ld(actl + (st * 4 + i) * GA16_KSTEP), bl = ga16_ld(actl + (st * 4 + i) * GA16_KSTEP + 1024);
        ga16_mac(S.ah[i], S.al[i], bh, bl, m, c);
    }
}
__device__ __forceinline__ void ga16_dense256(const uint4* __restrict__ PA, const unsigned char* actl, ga_f32x16& m, ga_f32x16& c) {
    Ga16Stage X, Y;
    ga16_fetch(X, PA, 0);
    ga16_fetch(Y, PA, 1);
    __builtin_amdgcn_sched_barrier(0);
    ga16_issue(X, actl, 0, m, c);
    __builtin_amdgcn_sched_barrier(0);
    ga16_fetch(X, PA, 2);
    ga16_issue(Y, actl, 1, m, c);
    __builtin_amdgcn_sched_barrier(0);
    ga16_fetch(Y, PA, 3);
    ga16_issue(X, actl, 2, m, c);
    __builtin_amdgcn_sched_barrier(0);
    ga16_issue(Y, actl, 3, m, c);
}

#define GA16_LDS_H 0                                 /* 2 buffers x 4 K-steps of h fragments */
#define GA16_LDS_X (GA16_LDS_H + 2 * 4 * GA16_KSTEP)  /* 10 sequence slots x [hi | lo] x 32 agents x 16 B (half 0 only) */
#define GA16_LDS_HOST (GA16_LDS_X + 10 * 1024)       /* [hi | lo] x 32 agents x 16 B */
#define GA16_LDS_ZERO (GA16_LDS_HOST + 1024)         /* what half 1 reads in the K-steps that hold fewer than 9 inputs */
#define GA16_LDS_ACT (GA16_LDS_ZERO + 1024)          /* 16 K-steps of layer activations */
#define GA16_LDS_BYTES (GA16_LDS_ACT + 16 * GA16_KSTEP)

__global__ void __launch_bounds__(512) k_ga3c_forward_h16(const unsigned char* __restrict__ P, const float* __restrict__ state,
                                                          const int32_t* __restrict__ agent_idx, int B_host,
                                                          const int32_t* __restrict__ B_dev, const double* __restrict__ pref,
                                                          float* ext_actions, int32_t* action_index, float* probs, uint32_t* list_ctr) {
    const int B = B_dev ? *B_dev : B_host;  // device-side count (cagym_ga3c_act): the grid covers the worst case
    // the last kernel of cagym_ga3c_act's chain starts the next list where this one ended (k_ga3c_select)
    if (list_ctr && blockIdx.x == 0 && threadIdx.x == 0) list_ctr[1] = list_ctr[0];
    if ((int)blockIdx.x * 32 >= B) return;
    GASTAMP_BEGIN();
    constexpr int AG = 32;
    __shared__ __attribute__((aligned(16))) unsigned char lds[GA16_LDS_BYTES];
    __shared__ int nseq[AG];
    unsigned char* hbuf = lds + GA16_LDS_H;
    unsigned char* xf = lds + GA16_LDS_X;
    unsigned char* hostf = lds + GA16_LDS_HOST;
    unsigned char* zero = lds + GA16_LDS_ZERO;
    unsigned char* act = lds + GA16_LDS_ACT;
    float* part = reinterpret_cast<float*>(lds + GA16_LDS_H);  // logits: the eight waves' partial sums [wave][11][agent] (h is dead)
    float* logit = reinterpret_cast<float*>(lds + GA16_LDS_X);  // [agent][12]
    static_assert(8 * 11 * AG * 4 <= 2 * 4 * GA16_KSTEP && AG * 12 * 4 <= 10 * 1024, "LDS aliasing");
    const int n = threadIdx.x, lane = n & 63, wave = __builtin_amdgcn_readfirstlane(n >> 6), half = lane >> 5, j = lane & 31;
    const int tile = blockIdx.x * AG;
    // the agent behind each of the tile's rows and its preferred speed, requested now for the action write at the very end
    int act_a = 0;
    double act_pref = 0.0;
    if (n < AG && tile + n < B) {
        act_a = agent_idx[tile + n];
        act_pref = pref[act_a];
    }
    // list_ctr != null (cagym_ga3c_act): the state rows are stored by place in the list - row tile + g, no index look-up in front
    const bool by_place = list_ctr != nullptr;
    // ---- inputs, normalised (network.py:125-148: x_hat = (x - avg) / std) and split, as operand fragments ------------------------
    if (n < AG + 10 * AG) {
        const bool is_host = n < AG;
        const int g = n & (AG - 1), t = (n >> 5) - 1;
        const int a = tile + g < B ? (by_place ? tile + g : agent_idx[tile + g]) : -1;
        const float* row = state + (size_t)(a >= 0 ? a : 0) * 76 + (is_host ? 1 : 6 + t * 7);
        float x[7];
#pragma unroll
        for (int c = 0; c < 7; c++) x[c] = a >= 0 && (c < 5 || !is_host) ? row[c] : 0.f;
        GaU4 hi, lo;
        hi.u = make_uint4(0, 0, 0, 0);
        lo.u = make_uint4(0, 0, 0, 0);
        if (is_host) {  // [n_others, dist_to_goal, heading_ego, pref_speed, radius]
            const int ns = (int)x[0];
            nseq[g] = a >= 0 ? (ns < 0 ? 0 : (ns > 10 ? 10 : ns)) : 0;
#pragma unroll
            for (int f = 1; f < 5; f++) {
                const float avg = f == 3 ? 1.0f : (f == 4 ? 0.5f : 0.0f);
                const float sd = f == 1 ? 5.0f : (f == 2 ? 3.14f : 1.0f);
                ga16_split((x[f] - avg) / sd, hi.f[f - 1], lo.f[f - 1]);
            }
            *reinterpret_cast<uint4*>(hostf + g * 16) = hi.u;
            *reinterpret_cast<uint4*>(hostf + 512 + g * 16) = lo.u;
        } else {  // [p_prll, p_orth, v_prll, v_orth, r_other, r_host + r_other, edge distance] of sequence slot t
#pragma unroll
            for (int c = 0; c < 7; c++) {
                const float avg = c == 4 ? 0.5f : (c == 6 ? 1.0f : 0.0f);
                const float sd = (c == 0 || c == 1 || c == 5) ? 5.0f : 1.0f;
                ga16_split((x[c] - avg) / sd, hi.f[c], lo.f[c]);
            }
            *reinterpret_cast<uint4*>(xf + t * 1024 + g * 16) = hi.u;
            *reinterpret_cast<uint4*>(xf + t * 1024 + 512 + g * 16) = lo.u;
        }
    }
    for (int e = n; e < (2 * 4 * GA16_KSTEP) / 16; e += 512) reinterpret_cast<uint4*>(hbuf)[e] = make_uint4(0, 0, 0, 0);  // h = 0
    if (n < 64) reinterpret_cast<uint4*>(zero)[n] = make_uint4(0, 0, 0, 0);
    // this wave's rows of the LSTM kernel (and of layer1's, which follows without a pause) stay in registers: 2 x 40 VGPRs
    const uint4* PAL = reinterpret_cast<const uint4*>(P + GA16_OFF_LSTM) + wave * (5 * 128) + lane;
    const uint4* PA1 = reinterpret_cast<const uint4*>(P + GA16_OFF_L1) + wave * (5 * 128) + lane;
    ga_h8 Ah[5], Al[5], A1h[5], A1l[5];
#pragma unroll
    for (int s = 0; s < 5; s++) {
        Ah[s] = ga16_ldg(PAL + s * 128);
        Al[s] = ga16_ldg(PAL + s * 128 + 64);
    }
#pragma unroll
    for (int s = 0; s < 5; s++) {
        A1h[s] = ga16_ldg(PA1 + s * 128);
        A1l[s] = ga16_ldg(PA1 + s * 128 + 64);
    }
    const ga_f32x16 bl = ga16_bias(P, 0, wave, half);
    const ga_f32x16 zero16 = ga_splat(0.f);
    __syncthreads();
    GASTAMP(0);
    int tmax = 0;
    for (int g = 0; g < AG; g++) tmax = nseq[g] > tmax ? nseq[g] : tmax;
    const int my_n = nseq[j];
    // ---- LSTM (network.py:83-90): this lane's cells = (unit 8 wave + 4 half + q, agent j), q = 0..3 ---------------------------------
    float cst[4] = {0.f, 0.f, 0.f, 0.f};
    const unsigned char* xl = half ? zero : xf + j * 16;  // half 1 of the feature K-step holds no inputs
    const int xstride = half ? 0 : 1024;
    const int wofs = (wave >> 1) * GA16_KSTEP + (wave & 1) * 512 + j * 16 + half * 8;  // where this lane's four h values live
    for (int t = 0; t < tmax; t++) {
        // the sequence starts in buffer tmax & 1 so that it always ends in buffer 0, which is layer1's operand as it stands
        const unsigned char* cur = hbuf + ((t + tmax) & 1) * (4 * GA16_KSTEP);
        unsigned char* nxt = hbuf + ((t + tmax + 1) & 1) * (4 * GA16_KSTEP);
        // every operand of the step requested before the first matrix instruction (one LDS latency per step, not five)
        ga_h8 xh[5], xl5[5];
#pragma unroll
        for (int s = 0; s < 4; s++) {
            xh[s] = ga16_ld(cur + s * GA16_KSTEP + lane * 16);
            xl5[s] = ga16_ld(cur + s * GA16_KSTEP + 1024 + lane * 16);
        }
        xh[4] = ga16_ld(xl + t * xstride);
        xl5[4] = ga16_ld(xl + t * xstride + 512);
        GaU2 oh, ol;  // a finished sequence carries its hidden state over
        oh.u = *reinterpret_cast<const uint2*>(cur + wofs);
        ol.u = *reinterpret_cast<const uint2*>(cur + wofs + 1024);
        __builtin_amdgcn_sched_barrier(0);
        ga_f32x16 m = bl, c = zero16;
#pragma unroll
        for (int s = 0; s < 5; s++) ga16_mac(Ah[s], Al[s], xh[s], xl5[s], m, c);
        const bool live = t < my_n;
        GaU2 nh, nl;
        float cn[4];
#pragma unroll
        for (int q = 0; q < 4; q++) {  // the four cells are independent: straight-line code, selected at the end
            const float gi = fmaf(c[q], GA16_ISC, m[q]), gj = fmaf(c[4 + q], GA16_ISC, m[4 + q]);
            const float gf = fmaf(c[8 + q], GA16_ISC, m[8 + q]), go = fmaf(c[12 + q], GA16_ISC, m[12 + q]);
            cn[q] = ga_fast_sigmoid(gf + 1.0f) * cst[q] + ga_fast_sigmoid(gi) * ga_fast_tanh(gj);
            const float hn = ga_fast_sigmoid(go) * ga_fast_tanh(cn[q]);
            ga16_split(hn, nh.f[q], nl.f[q]);
        }
#pragma unroll
        for (int q = 0; q < 4; q++) cst[q] = live ? cn[q] : cst[q];
        oh.u.x = live ? nh.u.x : oh.u.x;
        oh.u.y = live ? nh.u.y : oh.u.y;
        ol.u.x = live ? nl.u.x : ol.u.x;
        ol.u.y = live ? nl.u.y : ol.u.y;
        *reinterpret_cast<uint2*>(nxt + wofs) = oh.u;
        *reinterpret_cast<uint2*>(nxt + wofs + 1024) = ol.u;
        __syncthreads();  // the other buffer is complete; everyone has finished reading this one
    }
    GASTAMP(1);
    // ---- layer1: concat[host(4), h(64)] -> 256, ReLU (network.py:92-93): the final h buffer as it stands + the host K-step ----------
    {
        ga_f32x16 m = ga16_bias(P, 1, wave, half), c = zero16;
#pragma unroll
        for (int s = 0; s < 4; s++)
            ga16_mac(A1h[s], A1l[s], ga16_ld(hbuf + s * GA16_KSTEP + lane * 16), ga16_ld(hbuf + s * GA16_KSTEP + 1024 + lane * 16), m, c);
        const unsigned char* hl = half ? zero : hostf + j * 16;
        ga16_mac(A1h[4], A1l[4], ga16_ld(hl), ga16_ld(hl + 512), m, c);
        ga16_store_act(act, wave, lane, m, c);
    }
    __syncthreads();
    GASTAMP(2);
    // ---- layer2, fullyconnected1 (network.py:95, 47), in place ------------------------------------------------------------------------
#pragma unroll 1
    for (int layer = 0; layer < 2; layer++) {
        const uint4* PA = reinterpret_cast<const uint4*>(P + (layer == 0 ? GA16_OFF_L2 : GA16_OFF_L3)) + wave * (16 * 128) + lane;
        ga_f32x16 m = ga16_bias(P, 2 + layer, wave, half), c = zero16;
        ga16_dense256(PA, act + lane * 16, m, c);
        __syncthreads();  // in place: every wave has read the whole input
        ga16_store_act(act, wave, lane, m, c);
        __syncthreads();
        GASTAMP(3 + layer);
    }
    // ---- logits_p 256 -> 11 (network.py:50): each wave sums its two K-steps on one tile (rows >= 11 are zero weights) ---------------
    {
        const uint4* PA = reinterpret_cast<const uint4*>(P + GA16_OFF_LOG) + (2 * wave) * 128 + lane;
        ga_f32x16 m = zero16, c = zero16;
#pragma unroll
        for (int sp = 0; sp < 2; sp++)
            ga16_mac(ga16_ldg(PA + sp * 128), ga16_ldg(PA + sp * 128 + 64), ga16_ld(act + (2 * wave + sp) * GA16_KSTEP + lane * 16),
                     ga16_ld(act + (2 * wave + sp) * GA16_KSTEP + 1024 + lane * 16), m, c);
#pragma unroll
        for (int r = 0; r < 12; r++) {  // reg r <-> logit (r & 3) + 8 (r >> 2) + 4 half
            const int o = (r & 3) + 8 * (r >> 2) + 4 * half;
            if (o < 11) part[(wave * 11 + o) * AG + j] = fmaf(c[r], GA16_ISC, m[r]);
        }
    }
    __syncthreads();
    const float* bp = reinterpret_cast<const float*>(P + GA16_OFF_BIAS) + 4 * 256;
    for (int e = n; e < AG * 11; e += 512) {
        const int g = e / 11, o = e - g * 11;
        float s8[8];
#pragma unroll
        for (int w = 0; w < 8; w++) s8[w] = part[(w * 11 + o) * AG + g];
        logit[g * 12 + o] = bp[o] + (((s8[0] + s8[1]) + (s8[2] + s8[3])) + ((s8[4] + s8[5]) + (s8[6] + s8[7])));
    }
    __syncthreads();
    // ---- softmax_p, argmax, action (network.py:51, GA3CCADRLPolicy.py:39-42) ----------------------------------------------------------
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
