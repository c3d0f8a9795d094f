// cagym_ga3c16.h -- the GA3C-CADRL forward pass (policies/GA3C_CADRL/network.py:65-98) on gfx950's 16-bit matrix cores with
// SPLIT fp32 operands (round 4).
//
// Why.  v_mfma_f32_32x32x2_f32 (cagym_ga3c.h) runs at the fp32 vector rate: 64 cycles for 4 096 flops.  v_mfma_f32_32x32x16_f16
// does 32 768 flops in 32 cycles - 16 x - but takes 11-bit operands.  Every fp32 operand v is therefore handed over as TWO halves
//     hi = f16(v),   lo = f16(v - hi)                    (v - hi is exact in fp32; hi + lo carries 22 - 23 bits of v)
// and a product is three matrix instructions instead of one, smallest terms first, into the fp32 accumulator:
//     acc += hi_w * lo_x;   acc += lo_w * hi_x;   acc += hi_w * hi_x        (products of 11-bit numbers are exact in fp32;
//                                                                           the dropped lo * lo term is 2^-22 of the product)
// = 5.3 x the fp32 matrix rate at fp32-CLASS accuracy: a product is accurate to ~2^-22 relative, the same size as the rounding
// an fp32 fmaf chain of K = 256 accumulates.  Measured against the fp64 restatement (oracle/ga3c_ref.py, tests/test_ga3c.py prints
// it): max |p - p_fp64| = 2.2e-6 for this kernel, 1.7e-6 / 1.4e-6 for the fp32 matrix-core / vector kernels on live states; 1.0e-5
// against the fp32 kernel's 1.3e-5 on inputs scaled by 1e-3 .. 30.  For |v| < 2^-3 the low half is an f16 SUBNORMAL (absolute
// resolution 2^-24): the matrix core multiplies subnormals exactly as long as MODE.fp_denorm keeps them - the kernel sets the field
// itself - and the measured accuracy is the evidence (flushed low halves would leave 11-bit operands, errors of 1e-4).  The variant
// with the low half scaled by 2^11 into a second accumulator (-DGA16_SCALED: no subnormals, 16 more registers, one fma per result)
// measures 2.6e-6 / 8.2e-6 and 4 % slower (profiles/r4/ga3c16_ab.txt).  Activations are clamped to the f16 range (65 504) before
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
#ifdef GA16_SCALED  /* A/B only (tools/ga3c16_ab.sh): the low half scaled by 2^11 into its own accumulator - no f16 subnormals anywhere */
#define GA16_SC 2048.0f
#define GA16_ISC (1.0f / 2048.0f)
#else
#define GA16_SC 1.0f
#define GA16_ISC 1.0f
#endif

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

// The gates' non-linearities are evaluated as rcp(1 + exp2(.)): sigmoid(x) = 1 / (1 + 2^(-x log2 e)), tanh(x) = 1 - 2 / (1 + 2^(2 x log2 e)).
// The factor in front of x is folded into the packed LSTM kernel and bias (gate order i, j, f, o; the forget gate's + 1.0 - TF1's
// forget_bias - goes into the bias as well), so the matrix product delivers the exponent itself: 5 vector instructions less per cell.
__device__ __forceinline__ float ga16_gate_scale(int gate) { return gate == 1 ? 2.88539008f : -1.44269504f; }

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
        return Wb[GA_OFF_WL + (size_t)k * GA_W + col] * ga16_gate_scale(i >> 3);
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
        if (layer == 0) b = (Wb[GA_OFF_BL + (row >> 3) * 64 + 8 * w + (row & 7)] + ((row >> 3) == 2 ? 1.0f : 0.f)) * ga16_gate_scale(row >> 3);
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
#ifndef GA16_SCALED
    m = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bl, m, 0, 0, 0);
    m = __builtin_amdgcn_mfma_f32_32x32x16_f16(al, bh, m, 0, 0, 0);
    m = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bh, m, 0, 0, 0);
#else
    m = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bh, m, 0, 0, 0);
    c = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bl, c, 0, 0, 0);
    c = __builtin_amdgcn_mfma_f32_32x32x16_f16(al, bh, c, 0, 0, 0);
#endif
}
// the value of result register r
__device__ __forceinline__ float ga16_out(const ga_f32x16& m, const ga_f32x16& c, int r) {
#ifndef GA16_SCALED
    return m[r];
#else
    return fmaf(c[r], GA16_ISC, m[r]);
#endif
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
            const float v = fminf(fmaxf(ga16_out(m, c, 8 * sp + e), 0.f), 65504.f);
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
// X, Y hold stages 0 and 1 on entry (requested by the caller BEFORE the barriers in front of the layer: the L2 round trip runs
// under the previous layer's conversion and stores)
__device__ __forceinline__ void ga16_dense256(Ga16Stage& X, Ga16Stage& Y, const uint4* __restrict__ PA, const unsigned char* actl, ga_f32x16& m,
                                              ga_f32x16& c) {
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

// What a wave keeps in registers from the start of a tile: its rows of the LSTM kernel and of layer1's (which follows the sequence
// without a pause), their biases, one logit bias.  Requested by the caller BEFORE it builds / fetches the tile's state rows, so
// that the (cold: the env kernel has streamed through L2 since the last call) loads travel while that happens.
struct Ga16Pre {
    ga_h8 Ah[5], Al[5], A1h[5], A1l[5];
    ga_f32x16 bl, b1;
    float lo_bias;
};
__device__ __forceinline__ void ga16_preload(const unsigned char* __restrict__ P, Ga16Pre& W) {
    const int n = threadIdx.x, lane = n & 63, wave = __builtin_amdgcn_readfirstlane(n >> 6), half = lane >> 5;
    const uint4* PAL = reinterpret_cast<const uint4*>(P + GA16_OFF_LSTM) + wave * (5 * 128) + lane;
    const uint4* PA1 = reinterpret_cast<const uint4*>(P + GA16_OFF_L1) + wave * (5 * 128) + lane;
#pragma unroll
    for (int s = 0; s < 5; s++) {
        W.Ah[s] = ga16_ldg(PAL + s * 128);
        W.Al[s] = ga16_ldg(PAL + s * 128 + 64);
    }
#pragma unroll
    for (int s = 0; s < 5; s++) {
        W.A1h[s] = ga16_ldg(PA1 + s * 128);
        W.A1l[s] = ga16_ldg(PA1 + s * 128 + 64);
    }
    W.lo_bias = reinterpret_cast<const float*>(P + GA16_OFF_BIAS)[4 * 256 + (n - (n / 11) * 11)];  // thread n sums logit n % 11 of agent n / 11
    W.bl = ga16_bias(P, 0, wave, half);
    W.b1 = ga16_bias(P, 1, wave, half);
}

// where a tile's 32 state rows ([id, n_others, dist_to_goal, heading_ego, pref_speed, radius, 10 x 7 features], cagym_ga3c_state) come from
struct Ga16RowsGlobal {  // cagym_ga3c_forward: a [*, 76] table in HBM, rows by agent index (or by place in the list)
    const float* state;
    const int32_t* agent_idx;
    int tile, B;
    bool by_place;
    __device__ __forceinline__ const float* operator()(int g) const {
        return tile + g < B ? state + (size_t)(by_place ? tile + g : agent_idx[tile + g]) * 76 : nullptr;
    }
};
struct Ga16RowsLds {  // cagym_ga3c_act: the rows the workgroup has just built itself
    const float* rows;
    int nvalid;
    __device__ __forceinline__ const float* operator()(int g) const { return g < nvalid ? rows + g * 76 : nullptr; }
};

// The forward pass of one tile of 32 agents on the 512 lanes of the workgroup.  act_*: thread n < 32 carries agent n's flat index and
// preferred speed (requested by the caller ahead of time: the action write is the very last thing); action_index / probs_row: this
// thread's own output slots or null.  lds: GA16_LDS_BYTES; ends with every lane past the last barrier that reads it.
template <class Rows>
__device__ __forceinline__ void ga16_forward_tile(const unsigned char* __restrict__ P, const Ga16Pre& W, unsigned char* lds, int* nseq, const Rows& rows, bool act_valid,
                                                  int act_a, double act_pref, float* ext_actions, int32_t* action_index, float* probs_row) {
    constexpr int AG = 32;
    unsigned char* hbuf = lds + GA16_LDS_H;
    unsigned char* xf = lds + GA16_LDS_X;
    unsigned char* hostf = lds + GA16_LDS_HOST;
    unsigned char* zero = lds + GA16_LDS_ZERO;
    unsigned char* act = lds + GA16_LDS_ACT;
    float* part = reinterpret_cast<float*>(lds + GA16_LDS_H);  // logits: the eight waves' partial sums [wave][11][agent] (h is dead)
    float* logit = reinterpret_cast<float*>(lds + GA16_LDS_X);  // [agent][12]
    static_assert(8 * 11 * AG * 4 <= 2 * 4 * GA16_KSTEP && AG * 12 * 4 <= 10 * 1024, "LDS aliasing");
    const int n = threadIdx.x, lane = n & 63, wave = __builtin_amdgcn_readfirstlane(n >> 6), half = lane >> 5, j = lane & 31;
    GASTAMP_BEGIN();
    // ---- inputs, normalised (network.py:125-148: x_hat = (x - avg) / std) and split, as operand fragments ------------------------
    if (n < AG + 10 * AG) {
        const bool is_host = n < AG;
        const int g = n & (AG - 1), t = (n >> 5) - 1;
        const float* rowp = rows(g);
        const int a = rowp ? 0 : -1;
        const float* row = rowp + (is_host ? 1 : 6 + t * 7);
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
    const ga_h8 (&Ah)[5] = W.Ah, (&Al)[5] = W.Al, (&A1h)[5] = W.A1h, (&A1l)[5] = W.A1l;
    const ga_f32x16 bl = W.bl, b1 = W.b1;
    const int lo_g = n / 11, lo_o = n - lo_g * 11;  // logit (agent lo_g, action lo_o) is summed by thread n < 352 at the very end
    const float lo_bias = W.lo_bias;
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
            const float gi = ga16_out(m, c, q), gj = ga16_out(m, c, 4 + q);
            const float gf = ga16_out(m, c, 8 + q), go = ga16_out(m, c, 12 + q);
            // gi, gj, gf, go are exponents already (ga16_gate_scale)
            const float si = __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(gi)), sf = __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(gf));
            const float so = __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(go));
            const float tj = fmaf(-2.0f, __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(gj)), 1.0f);
            cn[q] = sf * cst[q] + si * tj;
            const float hn = so * fmaf(-2.0f, __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(2.88539008f * cn[q])), 1.0f);
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
    // Every layer requests the NEXT layer's first two weight stages and its bias before its own results are converted and stored:
    // the loads' round trip runs under the conversion, the LDS stores and the barrier(s).
    const uint4* PA2 = reinterpret_cast<const uint4*>(P + GA16_OFF_L2) + wave * (16 * 128) + lane;
    const uint4* PA3 = reinterpret_cast<const uint4*>(P + GA16_OFF_L3) + wave * (16 * 128) + lane;
    const uint4* PAP = reinterpret_cast<const uint4*>(P + GA16_OFF_LOG) + (2 * wave) * 128 + lane;
    Ga16Stage X, Y;
    ga_f32x16 bnext;
    {
        ga_f32x16 m = b1, c = zero16;
#pragma unroll
        for (int s = 0; s < 4; s++)
            ga16_mac(A1h[s], A1l[s], ga16_ld(hbuf + s * GA16_KSTEP + lane * 16), ga16_ld(hbuf + s * GA16_KSTEP + 1024 + lane * 16), m, c);
        const unsigned char* hl = half ? zero : hostf + j * 16;
        ga16_mac(A1h[4], A1l[4], ga16_ld(hl), ga16_ld(hl + 512), m, c);
        ga16_fetch(X, PA2, 0);
        ga16_fetch(Y, PA2, 1);
        bnext = ga16_bias(P, 2, wave, half);
        __builtin_amdgcn_sched_barrier(0);
        ga16_store_act(act, wave, lane, m, c);
    }
    __syncthreads();
    GASTAMP(2);
    // ---- layer2, fullyconnected1 (network.py:95, 47), in place ------------------------------------------------------------------------
    {
        ga_f32x16 m = bnext, c = zero16;
        ga16_dense256(X, Y, PA2, act + lane * 16, m, c);
        ga16_fetch(X, PA3, 0);
        ga16_fetch(Y, PA3, 1);
        bnext = ga16_bias(P, 3, wave, half);
        __builtin_amdgcn_sched_barrier(0);
        __syncthreads();  // in place: every wave has read the whole input
        ga16_store_act(act, wave, lane, m, c);
        __syncthreads();
        GASTAMP(3);
    }
    ga_h8 pah[2], pal[2];
    {
        ga_f32x16 m = bnext, c = zero16;
        ga16_dense256(X, Y, PA3, act + lane * 16, m, c);
#pragma unroll
        for (int sp = 0; sp < 2; sp++) {
            pah[sp] = ga16_ldg(PAP + sp * 128);
            pal[sp] = ga16_ldg(PAP + sp * 128 + 64);
        }
        __builtin_amdgcn_sched_barrier(0);
        __syncthreads();
        ga16_store_act(act, wave, lane, m, c);
        __syncthreads();
        GASTAMP(4);
    }
    // ---- logits_p 256 -> 11 (network.py:50): each wave sums its two K-steps on one tile (rows >= 11 are zero weights) ---------------
    {
        ga_f32x16 m = zero16, c = zero16;
#pragma unroll
        for (int sp = 0; sp < 2; sp++)
            ga16_mac(pah[sp], pal[sp], ga16_ld(act + (2 * wave + sp) * GA16_KSTEP + lane * 16),
                     ga16_ld(act + (2 * wave + sp) * GA16_KSTEP + 1024 + lane * 16), m, c);
#pragma unroll
        for (int r = 0; r < 12; r++) {  // reg r <-> logit (r & 3) + 8 (r >> 2) + 4 half
            const int o = (r & 3) + 8 * (r >> 2) + 4 * half;
            if (o < 11) part[(wave * 11 + o) * AG + j] = ga16_out(m, c, r);
        }
    }
    __syncthreads();
    if (n < AG * 11) {
        float s8[8];
#pragma unroll
        for (int w = 0; w < 8; w++) s8[w] = part[(w * 11 + lo_o) * AG + lo_g];
        logit[lo_g * 12 + lo_o] = lo_bias + (((s8[0] + s8[1]) + (s8[2] + s8[3])) + ((s8[4] + s8[5]) + (s8[6] + s8[7])));
    }
    __syncthreads();
    // ---- softmax_p, argmax, action (network.py:51, GA3CCADRLPolicy.py:39-42) ----------------------------------------------------------
    if (n < AG && act_valid) {
        const int a = act_a;
        float mx = logit[n * 12];
        int best = 0;
        for (int o = 1; o < 11; o++)
            if (logit[n * 12 + o] > mx) { mx = logit[n * 12 + o]; best = o; }
        if (probs_row) {
            float ex[11], s = 0.f;
            for (int o = 0; o < 11; o++) { ex[o] = expf(logit[n * 12 + o] - mx); s += ex[o]; }
            for (int o = 0; o < 11; o++) probs_row[o] = (ex[o] / s + 1e-4f) / (1.0f + 1e-4f * 11);
        }
        if (action_index) *action_index = best;
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

// MODE.fp_denorm[3:2] (f16 / f64) = 3: subnormal f16 operands and results are kept (hipcc's default; said here because the low
// halves depend on it).  hwreg(HW_REG_MODE = 1, offset 6, size 2)
__device__ __forceinline__ void ga16_keep_f16_subnormals() { __builtin_amdgcn_s_setreg((1 << 11) | (6 << 6) | 1, 3); }

// cagym_ga3c_forward: the B listed agents, their rows in a [*, 76] table (cagym_ga3c_state); one workgroup per 32 of them
__global__ void __launch_bounds__(512) k_ga3c_forward_h16(const unsigned char* __restrict__ P, const float* __restrict__ state,
                                                          const int32_t* __restrict__ agent_idx, int B, const double* __restrict__ pref,
                                                          float* ext_actions, int32_t* action_index, float* probs) {
    ga16_keep_f16_subnormals();
    __shared__ __attribute__((aligned(16))) unsigned char lds[GA16_LDS_BYTES];
    __shared__ int nseq[32];
    const int n = threadIdx.x, tile = blockIdx.x * 32;
    const bool act_valid = n < 32 && tile + n < B;
    int act_a = 0;
    double act_pref = 0.0;
    if (act_valid) {
        act_a = agent_idx[tile + n];
        act_pref = pref[act_a];
    }
    Ga16Pre W;
    ga16_preload(P, W);
    const Ga16RowsGlobal rows{state, agent_idx, tile, B, false};
    ga16_forward_tile(P, W, lds, nseq, rows, act_valid, act_a, act_pref, ext_actions, action_index && act_valid ? action_index + tile + n : nullptr,
                      probs && act_valid ? probs + (size_t)(tile + n) * 11 : nullptr);
}

// cagym_ga3c_act in ONE launch (GA3CCADRLPolicy.find_next_action, policies/GA3CCADRLPolicy.py:34-43, for every active GA3C agent):
// workgroup b owns the agent slots of worlds 32 b .. 32 b + 31, lists its active GA3C agents in slot order (ballots + one prefix over
// the waves' counts), and per tile of 32 listed agents builds the state rows in LDS (ga3c_state_row: the arithmetic of
// k_ga3c_state, 16 or 32 lanes per agent) and runs the forward pass on them.  cfg4 (one GA3C agent per world): exactly one tile per
// workgroup, 256 workgroups; a handle whose every slot is a GA3C agent runs max_agents tiles per workgroup one after the other (what
// the three-kernel chain did in as many rounds of workgroups).  Until round 4 (second session) the call was three launches -
// selection with ticket words, state rows through HBM, forward - 5.1 + 5.9 us in front of the network (profiles/r4).
template <int LPA>
__global__ void __launch_bounds__(512) k_ga3c_act_h16(CagymDev D, const unsigned char* __restrict__ P, int max_observed, float* ext_actions) {
    GASTAMP_BEGIN();
    ga16_keep_f16_subnormals();
    __shared__ __attribute__((aligned(16))) unsigned char lds[GA16_LDS_BYTES];
    __shared__ int nseq[32];
    __shared__ uint16_t list[32 * 32];  // this workgroup's active GA3C agents: slot index within its 32 worlds (<= 32 x 32 slots)
    __shared__ int wcnt[2][8];
    const int n = threadIdx.x, lane = n & 63, wave = n >> 6;
    const size_t first = (size_t)blockIdx.x * 32 * D.M, total = (size_t)D.N * D.M;
    const int nslots = (int)((total - first) < (size_t)(32 * D.M) ? (total - first) : (size_t)(32 * D.M));  // <= 1024: two passes of 512
    unsigned long long m[2];
    bool take[2];
    uint32_t st[2];
#pragma unroll
    for (int pass = 0; pass < 2; pass++) st[pass] = pass * 512 + n < nslots ? D.status[first + pass * 512 + n] : 0u;
    // the first tile's weights travel while the agents are listed and their state rows built (requested BEHIND the status words:
    // vector loads return in order)
    Ga16Pre W;
#ifndef GA16_NO_HOIST  /* A/B only */
    ga16_preload(P, W);
#endif
#pragma unroll
    for (int pass = 0; pass < 2; pass++) {
        take[pass] = (st[pass] & CAGYM_FLAG_ACTIVE) && ST_POLICY(st[pass]) == CAGYM_POL_GA3C;
        m[pass] = __ballot(take[pass]);
        if (lane == 0) wcnt[pass][wave] = __popcll(m[pass]);
    }
    __syncthreads();
    int cnt = 0;
#pragma unroll
    for (int pass = 0; pass < 2; pass++) {
        int base = 0;
        for (int w = 0; w < 8; w++) {
            const int c = wcnt[pass][w];
            if (w < wave) base += c;
            cnt += c;
        }
        base += pass ? wcnt[0][0] + wcnt[0][1] + wcnt[0][2] + wcnt[0][3] + wcnt[0][4] + wcnt[0][5] + wcnt[0][6] + wcnt[0][7] : 0;
        if (take[pass]) list[base + __popcll(m[pass] & ((1ull << lane) - 1ull))] = (uint16_t)(pass * 512 + n);
    }
    __syncthreads();
    // the state rows of a tile and the sort keys of its agents' lanes live where the forward pass keeps its layer activations (dead
    // until layer1's results are stored; the forward pass has read every row by then)
    float* rows = reinterpret_cast<float*>(lds + GA16_LDS_ACT);
    double(*sk1)[LPA] = reinterpret_cast<double(*)[LPA]>(lds + GA16_LDS_ACT + 32 * 76 * 4);
    double(*sk2)[LPA] = sk1 + 512 / LPA;
    static_assert(32 * 76 * 4 + 2 * 512 * 8 <= 16 * GA16_KSTEP, "state rows + keys fit the activation buffer");
    for (int tile = 0; tile < cnt; tile += 32) {
        const int nvalid = cnt - tile < 32 ? cnt - tile : 32;
        const bool act_valid = n < 32 && n < nvalid;
        int act_a = 0;
        double act_pref = 0.0;
        if (act_valid) {
            act_a = (int)(first + list[tile + n]);
            act_pref = D.pref[act_a];
        }
#pragma unroll
        for (int pass = 0; pass < LPA / 16; pass++) {  // 512 / LPA agents per pass
            const int al = n / LPA, g = pass * (512 / LPA) + al;
            const bool have = g < nvalid;
            ga3c_state_row<LPA>(D, max_observed, have, have ? first + list[tile + g] : 0, (size_t)g, al, n % LPA, sk1, sk2, rows);
            __syncthreads();  // rows complete; the next pass may overwrite the keys
        }
        GASTAMP(6);
#ifdef GA16_NO_HOIST
        if (tile == 0) ga16_preload(P, W);
#endif
        const Ga16RowsLds r{rows, nvalid};
        ga16_forward_tile(P, W, lds, nseq, r, act_valid, act_a, act_pref, ext_actions, (int32_t*)nullptr, (float*)nullptr);
        __syncthreads();  // the next tile's rows overwrite the activation buffer
        if (tile + 32 < cnt) ga16_preload(P, W);
    }
}

