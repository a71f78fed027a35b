// pve_actor.h -- MADDPG actor inference for every controlled vehicle of every intersection
// (reference model_agent_maddpg.py:23-49, called per vehicle with batch 1 from main.py:36-45, 404).
//
//   x(28) -> LayerNorm -> Dense 28x64 -> LayerNorm -> ReLU -> Dense 64x64 -> LayerNorm -> ReLU -> Dense 64x1 -> 3*tanh
//
// float32 like the TF graph (placeholder dtype, :15), on the matrix cores (k_actor_t below): this IS a dense
// contraction (2 x 5 952 flop per controlled vehicle), unlike the tick.
#pragma once
#include "pve_types.h"

namespace pve {

constexpr int ACT_IN = 28, ACT_H = 64;
// flat float32 weight vector (6393 values), in this order:
constexpr int AW_LN0_G = 0, AW_LN0_B = AW_LN0_G + ACT_IN, AW_W1 = AW_LN0_B + ACT_IN,
              AW_B1 = AW_W1 + ACT_IN * ACT_H, AW_LN1_G = AW_B1 + ACT_H, AW_LN1_B = AW_LN1_G + ACT_H,
              AW_W2 = AW_LN1_B + ACT_H, AW_B2 = AW_W2 + ACT_H * ACT_H, AW_LN2_G = AW_B2 + ACT_H,
              AW_LN2_B = AW_LN2_G + ACT_H, AW_W3 = AW_LN2_B + ACT_H, AW_B3 = AW_W3 + ACT_H, AW_TOTAL = AW_B3 + 1;
static_assert(AW_TOTAL == 6393, "actor weight count (SURVEY 8f-1)");

// ------------------------------------------------------------------------------------------------------
// Canonical float32 evaluation order of the network (plain code, host or device).  The matrix-core kernel below computes
// exactly this: every dot product and every LayerNorm sum in this order, one rounding per fused multiply-add.  The CPU
// test emulator calls it directly.  The order follows the data layout of v_mfma_f32_16x16x4_f32 in the transposed form
// H^T = W^T X^T (hidden units x vehicles): lane group q = 0..3 of a wave holds the hidden units 16 m + 4 q + r
// (m, r = 0..3) of its vehicle, and one instruction contracts the four values k(q), q = 0..3, in that order.
inline float actor_ln_combine(const float p[4]) { return (p[0] + p[1]) + (p[2] + p[3]); }   // xor-16 then xor-32 exchange

inline float actor_canonical(const float *W, const float *x)
{
    // LayerNorm over the 28 inputs: lane group q holds features 4 s + q
    float p[4], a0[ACT_IN];
    for (int q = 0; q < 4; q++) { p[q] = 0.f; for (int s = 0; s < ACT_IN / 4; s++) p[q] += x[4 * s + q]; }
    float mean = actor_ln_combine(p) / (float)ACT_IN;
    for (int q = 0; q < 4; q++) {
        p[q] = 0.f;
        for (int s = 0; s < ACT_IN / 4; s++) { const float d = x[4 * s + q] - mean; p[q] = fmaf(d, d, p[q]); }
    }
    float rstd = 1.0f / sqrtf(actor_ln_combine(p) / (float)ACT_IN + 1e-12f);
    for (int k = 0; k < ACT_IN; k++) {
        const float inv = rstd * W[AW_LN0_G + k];
        a0[k] = fmaf(x[k], inv, W[AW_LN0_B + k] - mean * inv);
    }
    // dense 28 -> 64: k = 0 .. 27 in order (step s contracts k = 4 s + q, q = 0..3)
    float h[ACT_H], g[ACT_H];
    for (int u = 0; u < ACT_H; u++) {
        float acc = W[AW_B1 + u];
        for (int k = 0; k < ACT_IN; k++) acc = fmaf(W[AW_W1 + k * ACT_H + u], a0[k], acc);
        h[u] = acc;
    }
    for (int layer = 1; layer <= 2; layer++) {
        const int G = layer == 1 ? AW_LN1_G : AW_LN2_G, B = layer == 1 ? AW_LN1_B : AW_LN2_B;
        float *src = layer == 1 ? h : g;
        // LayerNorm + ReLU over the 64 hidden units.  Lane group q holds u[m][r] = unit 16 m + 4 q + r as four float4
        // (one per m): sums run component-wise over m first, then over the 4 components, then over the lane groups
        for (int q = 0; q < 4; q++) {
            float a4[4];
            for (int r = 0; r < 4; r++)
                a4[r] = (src[4 * q + r] + src[16 + 4 * q + r]) + (src[32 + 4 * q + r] + src[48 + 4 * q + r]);
            p[q] = (a4[0] + a4[1]) + (a4[2] + a4[3]);
        }
        mean = actor_ln_combine(p) / (float)ACT_H;
        for (int q = 0; q < 4; q++) {
            float e4[4];
            for (int r = 0; r < 4; r++) {
                float e = 0.f;
                for (int m = 0; m < 4; m++) { const float d = src[16 * m + 4 * q + r] - mean; e = fmaf(d, d, e); }
                e4[r] = e;
            }
            p[q] = (e4[0] + e4[1]) + (e4[2] + e4[3]);
        }
        rstd = 1.0f / sqrtf(actor_ln_combine(p) / (float)ACT_H + 1e-12f);
        for (int u = 0; u < ACT_H; u++) {
            const float inv = rstd * W[G + u];
            src[u] = fmaxf(fmaf(src[u], inv, W[B + u] - mean * inv), 0.f);
        }
        if (layer == 1) {
            // dense 64 -> 64: contraction order (m, r, q): k = 16 m + 4 q + r
            for (int u = 0; u < ACT_H; u++) {
                float acc = W[AW_B2 + u];
                for (int m = 0; m < 4; m++) for (int r = 0; r < 4; r++) for (int q = 0; q < 4; q++) {
                    const int k = 16 * m + 4 * q + r;
                    acc = fmaf(W[AW_W2 + k * ACT_H + u], h[k], acc);
                }
                g[u] = acc;
            }
        }
    }
    // dense 64 -> 1: per lane group in (m, r) order, groups combined like the LayerNorm sums
    for (int q = 0; q < 4; q++) {
        p[q] = 0.f;
        for (int m = 0; m < 4; m++) for (int r = 0; r < 4; r++) { const int k = 16 * m + 4 * q + r; p[q] = fmaf(g[k], W[AW_W3 + k], p[q]); }
    }
    return 3.0f * tanhf(actor_ln_combine(p) + W[AW_B3]);
}

#if defined(__HIPCC__)

// ------------------------------------------------------------------------------------------------------
// k_actor_t: the network on the matrix cores in the TRANSPOSED form H^T = W^T X^T.  v_mfma_f32_16x16x4_f32 is an exact
// float32 FMA chain at the f32 vector rate; operands are per lane:
//   A[i = lane & 15][k = lane >> 4]  = W[k][16 m' + i]            (a weight, straight from L1 / L2: 64-B segments)
//   B[k = lane >> 4][j = lane & 15]  = activation k of vehicle j   (already in this lane's registers, see below)
//   D[i = 4 (lane >> 4) + r][j]      -> lane (j, q = lane >> 4) holds hidden units 16 m' + 4 q + r of vehicle j.
// The output layout of one layer IS the B-operand layout of the next (the contraction index is merely enumerated in the
// order (m, r, q)), so activations never move: no LDS tile, no transposes, no weights parked in registers.  LayerNorm =
// 16 in-lane values + two cross-lane exchanges (lanes j, j + 16, j + 32, j + 48 hold one vehicle).  16 controlled
// vehicles per wave and pass (ballot compaction), 28 + 64 MFMAs per pass.
typedef float pve_v4f __attribute__((ext_vector_type(4)));

typedef unsigned pve_v2u __attribute__((ext_vector_type(2)));
// sum over the four lanes (j, j + 16, j + 32, j + 48) that hold one vehicle, every lane gets the total: two gfx950 row
// swaps (VALU, no LDS crossbar round trip): ((row 0 + row 1) + (row 2 + row 3)) in every lane
__device__ __forceinline__ float actor_xsum(float s)
{
    unsigned u = __float_as_uint(s);
    pve_v2u r = __builtin_amdgcn_permlane16_swap(u, u, false, false);       // odd rows of one copy <-> even rows of the other
    s = __uint_as_float(r[0]) + __uint_as_float(r[1]);
    u = __float_as_uint(s);
    r = __builtin_amdgcn_permlane32_swap(u, u, false, false);
    return __uint_as_float(r[0]) + __uint_as_float(r[1]);
}

// LayerNorm + ReLU over the 64 hidden units of a vehicle: 16 of them in this lane (units 16 m + 4 q + r), the rest in the
// lanes 16 / 32 / 48 further on
__device__ __forceinline__ void actor_ln_relu16(pve_v4f (&v)[4], const float *__restrict__ gamma, const float *__restrict__ beta, int q)
{   // float4 arithmetic = packed f32 instructions (v_pk_add / v_pk_mul / v_pk_fma): half the VALU issue of scalar code
    const pve_v4f a4 = (v[0] + v[1]) + (v[2] + v[3]);
    const float mean = actor_xsum((a4[0] + a4[1]) + (a4[2] + a4[3])) / (float)ACT_H;
    pve_v4f e4 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int m = 0; m < 4; m++) { const pve_v4f d = v[m] - mean; e4 = __builtin_elementwise_fma(d, d, e4); }
    const float rstd = 1.0f / sqrtf(actor_xsum((e4[0] + e4[1]) + (e4[2] + e4[3])) / (float)ACT_H + 1e-12f);
    const pve_v4f zero = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int m = 0; m < 4; m++) {
        const pve_v4f ga = *(const pve_v4f *)(gamma + 16 * m + 4 * q), be = *(const pve_v4f *)(beta + 16 * m + 4 * q);
        const pve_v4f inv = ga * rstd;
        v[m] = __builtin_elementwise_max(__builtin_elementwise_fma(v[m], inv, be - inv * mean), zero);
    }
}

// Workgroup = 4 waves sharing ONE copy of the two dense kernels in LDS (rows padded to 68 floats: the four lane groups of
// an A-operand read hit disjoint banks).  Every wave is on its own: it loops over intersections (persistent), compacts
// the controlled vehicles of its intersection with ballots and runs their 16-vehicle tiles one after the other -- no
// workgroup barrier after the weights are staged, so the waves of a SIMD drift apart and one wave's loads / LayerNorms
// hide under another's MFMAs.  The next intersection's flags and the next tile's rows are loaded one step ahead.
constexpr int ACT_WPAD = 68;
// the small parameter vectors, staged next to the dense kernels (reads of the parameters inside a tile are LDS reads, not
// L1 / L2 round trips in front of the MFMAs that need them)
constexpr int SM_LN0_G = 0, SM_LN0_B = 28, SM_B1 = 56, SM_LN1_G = 120, SM_LN1_B = 184, SM_B2 = 248, SM_LN2_G = 312,
              SM_LN2_B = 376, SM_W3 = 440, SM_B3 = 504, SM_TOTAL = 508;

// LayerNorm + ReLU over the 64 hidden units of a vehicle: 16 of them in this lane (units 16 m + 4 q + r), the rest in the
// lanes 16 / 32 / 48 further on; parameters from LDS
__device__ __forceinline__ void actor_ln_relu16_lds(pve_v4f (&v)[4], const float *gamma, const float *beta, int q)
{   // float4 arithmetic = packed f32 instructions (v_pk_add / v_pk_mul / v_pk_fma): half the VALU issue of scalar code
    const pve_v4f a4 = (v[0] + v[1]) + (v[2] + v[3]);
    const float mean = actor_xsum((a4[0] + a4[1]) + (a4[2] + a4[3])) / (float)ACT_H;
    pve_v4f e4 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int m = 0; m < 4; m++) { const pve_v4f d = v[m] - mean; e4 = __builtin_elementwise_fma(d, d, e4); }
    const float rstd = 1.0f / sqrtf(actor_xsum((e4[0] + e4[1]) + (e4[2] + e4[3])) / (float)ACT_H + 1e-12f);
    const pve_v4f zero = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int m = 0; m < 4; m++) {
        const pve_v4f ga = *(const pve_v4f *)(gamma + 16 * m + 4 * q), be = *(const pve_v4f *)(beta + 16 * m + 4 * q);
        const pve_v4f inv = ga * rstd;
        v[m] = __builtin_elementwise_max(__builtin_elementwise_fma(v[m], inv, be - inv * mean), zero);
    }
}

template <int CAP, typename OBS_T>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(4, 4))) void k_actor_t(const float *__restrict__ W, const OBS_T *__restrict__ obs,
                                                 const int32_t *__restrict__ meta, double *__restrict__ actions,
                                                 int n_envs)
{
    __shared__ float Ws1[ACT_IN][ACT_WPAD], Ws2[ACT_H][ACT_WPAD];
    __shared__ __attribute__((aligned(16))) float Wsm[SM_TOTAL];
    __shared__ unsigned char slot_of_s[4][CAP];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, jl = lane & 15, q = lane >> 4;
    unsigned char *slot_of = slot_of_s[wave];
    const int stride = gridDim.x * 4;
    int env = blockIdx.x * 4 + wave;
    int mt[CAP / 64];                                         // flags of the wave's next intersection (loaded one ahead)
#pragma unroll
    for (int sub = 0; sub < CAP / 64; sub++) mt[sub] = env < n_envs ? meta[(size_t)env * CAP + sub * 64 + lane] : 0;
    for (int i = tid; i < ACT_IN * ACT_H; i += 256) Ws1[i >> 6][i & 63] = W[AW_W1 + i];
    for (int i = tid; i < ACT_H * ACT_H; i += 256) Ws2[i >> 6][i & 63] = W[AW_W2 + i];
    if (tid < 2 * ACT_IN) Wsm[SM_LN0_G + tid] = W[AW_LN0_G + tid];
    if (tid < ACT_H) {
        Wsm[SM_B1 + tid] = W[AW_B1 + tid]; Wsm[SM_B2 + tid] = W[AW_B2 + tid]; Wsm[SM_W3 + tid] = W[AW_W3 + tid];
    }
    if (tid < 2 * ACT_H) { Wsm[SM_LN1_G + tid] = W[AW_LN1_G + tid]; Wsm[SM_LN2_G + tid] = W[AW_LN2_G + tid]; }
    if (tid == 0) Wsm[SM_B3] = W[AW_B3];
    __syncthreads();                                          // weights staged
    for (; env < n_envs; env += stride) {
        const size_t base = (size_t)env * CAP;
        // controlled-vehicle compaction (wave-local: DS operations of one wave execute in order)
        int nctl = 0;
#pragma unroll
        for (int sub = 0; sub < CAP / 64; sub++) {
            const int s = sub * 64 + lane;
            const bool c = (mt[sub] & (M_ALIVE | M_CONTROL)) == (M_ALIVE | M_CONTROL);
            const unsigned long long b = __ballot(c);
            const int rank = nctl + __builtin_popcountll(b & ((1ull << lane) - 1ull));
            if (c) slot_of[rank] = (unsigned char)s;
            else actions[base + s] = 0.0;                     // main.py:401: uncontrolled vehicles get 0
            nctl += __builtin_popcountll(b);
        }
        {
            const int en = env + stride;
#pragma unroll
            for (int sub = 0; sub < CAP / 64; sub++) mt[sub] = en < n_envs ? meta[(size_t)en * CAP + sub * 64 + lane] : 0;
        }
        // rows of the first tile
        float xr[ACT_IN / 4];
        int slot = slot_of[jl < nctl ? jl : 0];
        if (nctl > 0) {
            const OBS_T *row = obs + (base + slot) * OBSW;
#pragma unroll
            for (int s = 0; s < ACT_IN / 4; s++) xr[s] = (float)row[4 * s + q];
        }
        for (int v0 = 0; v0 < nctl; v0 += 16) {              // 16 vehicles per pass
            // the A operands are read from LDS one step ahead of the MFMAs that consume them; an offset the compiler cannot
            // see through keeps it from hoisting all 92 reads out of the loops into 250 registers (1 wave per SIMD)
            int wo = 0;
            asm volatile("" : "+v"(wo));
            const bool valid = v0 + jl < nctl;
            const int cur_slot = slot;
            // ---- inputs: lane (j, q) holds features 4 s + q of vehicle j; the next tile's rows are requested now
            float x[ACT_IN / 4];
#pragma unroll
            for (int s = 0; s < ACT_IN / 4; s++) x[s] = xr[s];
            if (v0 + 16 < nctl) {
                slot = slot_of[v0 + 16 + jl < nctl ? v0 + 16 + jl : v0 + 16];
                const OBS_T *row = obs + (base + slot) * OBSW;
#pragma unroll
                for (int s = 0; s < ACT_IN / 4; s++) xr[s] = (float)row[4 * s + q];
            }
            {   // LayerNorm over the 28 inputs
                float sum = 0.f;
#pragma unroll
                for (int s = 0; s < ACT_IN / 4; s++) sum += x[s];
                const float mean = actor_xsum(sum) / (float)ACT_IN;
                float var = 0.f;
#pragma unroll
                for (int s = 0; s < ACT_IN / 4; s++) { const float d = x[s] - mean; var = fmaf(d, d, var); }
                const float rstd = 1.0f / sqrtf(actor_xsum(var) / (float)ACT_IN + 1e-12f);
#pragma unroll
                for (int s = 0; s < ACT_IN / 4; s++) {
                    const float inv = rstd * Wsm[SM_LN0_G + 4 * s + q];
                    x[s] = fmaf(x[s], inv, Wsm[SM_LN0_B + 4 * s + q] - mean * inv);
                }
            }
            // ---- dense 28 -> 64 (+ bias): 7 steps x 4 unit tiles, A operands double-buffered
            pve_v4f h[4];
#pragma unroll
            for (int m = 0; m < 4; m++) h[m] = *(const pve_v4f *)(Wsm + SM_B1 + 16 * m + 4 * q);
            {
                float wa[2][4];
#pragma unroll
                for (int m = 0; m < 4; m++) wa[0][m] = Ws1[q][16 * m + jl + wo];
#pragma unroll
                for (int s = 0; s < ACT_IN / 4; s++) {
                    if (s + 1 < ACT_IN / 4) {
#pragma unroll
                        for (int m = 0; m < 4; m++) wa[(s + 1) & 1][m] = Ws1[4 * (s + 1) + q][16 * m + jl + wo];
                    }
#pragma unroll
                    for (int m = 0; m < 4; m++) h[m] = __builtin_amdgcn_mfma_f32_16x16x4f32(wa[s & 1][m], x[s], h[m], 0, 0, 0);
                }
            }
            // ---- LayerNorm_1 + ReLU, dense 64 -> 64, LayerNorm_2 + ReLU
            actor_ln_relu16_lds(h, Wsm + SM_LN1_G, Wsm + SM_LN1_B, q);
            pve_v4f g[4];
#pragma unroll
            for (int m2 = 0; m2 < 4; m2++) g[m2] = *(const pve_v4f *)(Wsm + SM_B2 + 16 * m2 + 4 * q);
            {
                float wa[2][4];
#pragma unroll
                for (int m2 = 0; m2 < 4; m2++) wa[0][m2] = Ws2[4 * q][16 * m2 + jl + wo];
#pragma unroll
                for (int st = 0; st < 16; st++) {             // step st = 4 m + r contracts k = 16 m + 4 q + r
                    const int m = st >> 2, r = st & 3;
                    if (st + 1 < 16) {
                        const int mn = (st + 1) >> 2, rn = (st + 1) & 3;
#pragma unroll
                        for (int m2 = 0; m2 < 4; m2++) wa[(st + 1) & 1][m2] = Ws2[16 * mn + 4 * q + rn][16 * m2 + jl + wo];
                    }
#pragma unroll
                    for (int m2 = 0; m2 < 4; m2++)
                        g[m2] = __builtin_amdgcn_mfma_f32_16x16x4f32(wa[st & 1][m2], h[m][r], g[m2], 0, 0, 0);
                }
            }
            actor_ln_relu16_lds(g, Wsm + SM_LN2_G, Wsm + SM_LN2_B, q);
            // ---- dense 64 -> 1, 3 tanh
            float part = 0.f;
#pragma unroll
            for (int m = 0; m < 4; m++) {
                const pve_v4f w3 = *(const pve_v4f *)(Wsm + SM_W3 + 16 * m + 4 * q);
#pragma unroll
                for (int r = 0; r < 4; r++) part = fmaf(g[m][r], w3[r], part);
            }
            const float a = 3.0f * tanhf(actor_xsum(part) + Wsm[SM_B3]);
            if (q == 0 && valid) actions[base + cur_slot] = (double)a;
        }
    }
}

// ------------------------------------------------------------------------------------------------------
// k_actor_h: the same network with every float32 operand split into two halves, x = hi + lo (hi = half(x), lo =
// half(x - hi): 22 significant bits; gfx950's matrix cores honour f16 subnormals -- tools/mfma_layout_probe.hip), and
// three v_mfma_f32_16x16x32_f16 per product block (hi*hi + hi*lo + lo*hi, float32 accumulation; the lo*lo term is
// 2^-22 relative).  The f16 instruction contracts K = 32 at 16x the rate of the f32 one, so a 16-vehicle tile costs 36
// MFMAs x 16 cycles instead of 92 x 32.  Same transposed formulation: lane (j, q) supplies B[k = 8 q + e][j] as 8
// consecutive halves -- for layer 2 exactly the two accumulator registers h[2 b], h[2 b + 1] it already holds for K-block
// b (k = 32 b + 8 q + e  <->  hidden unit 16 (2 b + e / 4) + 4 q + e % 4) -- and the weights are staged in LDS in the
// A-operand layout ([block][unit tile][lane][8 halves]: one conflict-free 16-byte read per MFMA).
// Not bit-identical to the float32 chain of k_actor_t (actor_canonical): ~1e-6 relative per dot product; the action
// parity bar is 5e-4 (tests/actor_scenarios.py).  PVE_CFG_ACTOR_F32 selects k_actor_t.
typedef _Float16 pve_v8h __attribute__((ext_vector_type(8)));

__device__ __forceinline__ void split8(const float (&x)[8], pve_v8h &hi, pve_v8h &lo)
{
#pragma unroll
    for (int e = 0; e < 8; e++) { const _Float16 h = (_Float16)x[e]; hi[e] = h; lo[e] = (_Float16)(x[e] - (float)h); }
}

template <int CAP, typename OBS_T>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(4, 4))) void k_actor_h(const float *__restrict__ W, const OBS_T *__restrict__ obs,
                                                 const int32_t *__restrict__ meta, double *__restrict__ actions,
                                                 int n_envs)
{
    // A operands: [hi | lo][unit tile m][lane][8] for layer 1 (K = 28, padded to 32), [hi | lo][K-block b][m][lane][8] for layer 2
    __shared__ __attribute__((aligned(16))) _Float16 A1[2][4][64][8], A2[2][2][4][64][8];
    __shared__ __attribute__((aligned(16))) float Wsm[SM_TOTAL];
    __shared__ unsigned char slot_of_s[4][CAP];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, jl = lane & 15, q = lane >> 4;
    unsigned char *slot_of = slot_of_s[wave];
    const int stride = gridDim.x * 4;
    int env = blockIdx.x * 4 + wave;
    int mt[CAP / 64];                                         // flags of the wave's next intersection (loaded one ahead)
#pragma unroll
    for (int sub = 0; sub < CAP / 64; sub++) mt[sub] = env < n_envs ? meta[(size_t)env * CAP + sub * 64 + lane] : 0;
    // staging: one thread per operand vector (8 halves of one lane: loads 8 weights, splits them, two 16-byte LDS writes)
    for (int n = tid; n < (4 + 8) * 64; n += 256) {
        const int l = n & 63, t8 = n >> 6, qq = l >> 4, col = l & 15;
        float w[8];
        if (t8 < 4) {                                         // layer 1, unit tile m = t8: W1[8 q + e][16 m + j]
#pragma unroll
            for (int e = 0; e < 8; e++) { const int k = 8 * qq + e; w[e] = k < ACT_IN ? W[AW_W1 + k * ACT_H + 16 * t8 + col] : 0.f; }
        } else {                                              // layer 2, K-block bk, unit tile m: W2[16 (2 bk + e / 4) + 4 q + e % 4][16 m + j]
            const int bk = (t8 - 4) >> 2, m = (t8 - 4) & 3;
#pragma unroll
            for (int e = 0; e < 8; e++) w[e] = W[AW_W2 + (16 * (2 * bk + (e >> 2)) + 4 * qq + (e & 3)) * ACT_H + 16 * m + col];
        }
        pve_v8h hi, lo;
        split8(w, hi, lo);
        if (t8 < 4) { *(pve_v8h *)&A1[0][t8][l][0] = hi; *(pve_v8h *)&A1[1][t8][l][0] = lo; }
        else { *(pve_v8h *)&A2[0][(t8 - 4) >> 2][(t8 - 4) & 3][l][0] = hi; *(pve_v8h *)&A2[1][(t8 - 4) >> 2][(t8 - 4) & 3][l][0] = lo; }
    }
    if (tid < 2 * ACT_IN) Wsm[SM_LN0_G + tid] = W[AW_LN0_G + tid];
    if (tid < ACT_H) {
        Wsm[SM_B1 + tid] = W[AW_B1 + tid]; Wsm[SM_B2 + tid] = W[AW_B2 + tid]; Wsm[SM_W3 + tid] = W[AW_W3 + tid];
    }
    if (tid < 2 * ACT_H) { Wsm[SM_LN1_G + tid] = W[AW_LN1_G + tid]; Wsm[SM_LN2_G + tid] = W[AW_LN2_G + tid]; }
    if (tid == 0) Wsm[SM_B3] = W[AW_B3];
    __syncthreads();                                          // weights staged
    const int nf = q < 3 ? 8 : ACT_IN - 24;                   // features 8 q .. 8 q + nf - 1 of the 28 live in this lane
    for (; env < n_envs; env += stride) {
        const size_t base = (size_t)env * CAP;
        int nctl = 0;
#pragma unroll
        for (int sub = 0; sub < CAP / 64; sub++) {
            const int s = sub * 64 + lane;
            const bool c = (mt[sub] & (M_ALIVE | M_CONTROL)) == (M_ALIVE | M_CONTROL);
            const unsigned long long b = __ballot(c);
            const int rank = nctl + __builtin_popcountll(b & ((1ull << lane) - 1ull));
            if (c) slot_of[rank] = (unsigned char)s;
            else actions[base + s] = 0.0;                     // main.py:401: uncontrolled vehicles get 0
            nctl += __builtin_popcountll(b);
        }
        {
            const int en = env + stride;
#pragma unroll
            for (int sub = 0; sub < CAP / 64; sub++) mt[sub] = en < n_envs ? meta[(size_t)en * CAP + sub * 64 + lane] : 0;
        }
        float xr[8];
        int slot = slot_of[jl < nctl ? jl : 0];
        if (nctl > 0) {
            const OBS_T *row = obs + (base + slot) * OBSW + 8 * q;
#pragma unroll
            for (int e = 0; e < 8; e++) xr[e] = e < nf ? (float)row[e] : 0.f;
        }
        for (int v0 = 0; v0 < nctl; v0 += 16) {              // 16 vehicles per pass
            int wo = 0;
            asm volatile("" : "+v"(wo));                      // (keeps the A-operand reads inside the loop)
            const bool valid = v0 + jl < nctl;
            const int cur_slot = slot;
            float x[8];
#pragma unroll
            for (int e = 0; e < 8; e++) x[e] = xr[e];
            if (v0 + 16 < nctl) {
                slot = slot_of[v0 + 16 + jl < nctl ? v0 + 16 + jl : v0 + 16];
                const OBS_T *row = obs + (base + slot) * OBSW + 8 * q;
#pragma unroll
                for (int e = 0; e < 8; e++) xr[e] = e < nf ? (float)row[e] : 0.f;
            }
            {   // LayerNorm over the 28 inputs (8, 8, 8, 4 per lane group)
                float sum = 0.f;
#pragma unroll
                for (int e = 0; e < 8; e++) sum += x[e];
                const float mean = actor_xsum(sum) / (float)ACT_IN;
                float var = 0.f;
#pragma unroll
                for (int e = 0; e < 8; e++) { const float d = e < nf ? x[e] - mean : 0.f; var = fmaf(d, d, var); }
                const float rstd = 1.0f / sqrtf(actor_xsum(var) / (float)ACT_IN + 1e-12f);
#pragma unroll
                for (int e = 0; e < 8; e++) {
                    const int k = e < nf ? 8 * q + e : 0;
                    const float inv = rstd * Wsm[SM_LN0_G + k];
                    x[e] = e < nf ? fmaf(x[e], inv, Wsm[SM_LN0_B + k] - mean * inv) : 0.f;
                }
            }
            // ---- dense 28 -> 64 (+ bias)
            pve_v4f h[4];
#pragma unroll
            for (int m = 0; m < 4; m++) h[m] = *(const pve_v4f *)(Wsm + SM_B1 + 16 * m + 4 * q);
            {
                pve_v8h xh, xl;
                split8(x, xh, xl);
#pragma unroll
                for (int m = 0; m < 4; m++) h[m] = __builtin_amdgcn_mfma_f32_16x16x32_f16(*(const pve_v8h *)&A1[0][m][lane + wo][0], xh, h[m], 0, 0, 0);
#pragma unroll
                for (int m = 0; m < 4; m++) h[m] = __builtin_amdgcn_mfma_f32_16x16x32_f16(*(const pve_v8h *)&A1[0][m][lane + wo][0], xl, h[m], 0, 0, 0);
#pragma unroll
                for (int m = 0; m < 4; m++) h[m] = __builtin_amdgcn_mfma_f32_16x16x32_f16(*(const pve_v8h *)&A1[1][m][lane + wo][0], xh, h[m], 0, 0, 0);
            }
            // ---- LayerNorm_1 + ReLU, dense 64 -> 64, LayerNorm_2 + ReLU
            actor_ln_relu16_lds(h, Wsm + SM_LN1_G, Wsm + SM_LN1_B, q);
            pve_v4f g[4];
#pragma unroll
            for (int m2 = 0; m2 < 4; m2++) g[m2] = *(const pve_v4f *)(Wsm + SM_B2 + 16 * m2 + 4 * q);
#pragma unroll
            for (int bk = 0; bk < 2; bk++) {                  // K-block bk: this lane's units of h[2 bk], h[2 bk + 1]
                float hv[8];
#pragma unroll
                for (int e = 0; e < 8; e++) hv[e] = h[2 * bk + (e >> 2)][e & 3];
                pve_v8h hh, hl;
                split8(hv, hh, hl);
#pragma unroll
                for (int m2 = 0; m2 < 4; m2++) g[m2] = __builtin_amdgcn_mfma_f32_16x16x32_f16(*(const pve_v8h *)&A2[0][bk][m2][lane + wo][0], hh, g[m2], 0, 0, 0);
#pragma unroll
                for (int m2 = 0; m2 < 4; m2++) g[m2] = __builtin_amdgcn_mfma_f32_16x16x32_f16(*(const pve_v8h *)&A2[0][bk][m2][lane + wo][0], hl, g[m2], 0, 0, 0);
#pragma unroll
                for (int m2 = 0; m2 < 4; m2++) g[m2] = __builtin_amdgcn_mfma_f32_16x16x32_f16(*(const pve_v8h *)&A2[1][bk][m2][lane + wo][0], hh, g[m2], 0, 0, 0);
            }
            actor_ln_relu16_lds(g, Wsm + SM_LN2_G, Wsm + SM_LN2_B, q);
            // ---- dense 64 -> 1, 3 tanh
            float part = 0.f;
#pragma unroll
            for (int m = 0; m < 4; m++) {
                const pve_v4f w3 = *(const pve_v4f *)(Wsm + SM_W3 + 16 * m + 4 * q);
#pragma unroll
                for (int r = 0; r < 4; r++) part = fmaf(g[m][r], w3[r], part);
            }
            const float a = 3.0f * tanhf(actor_xsum(part) + Wsm[SM_B3]);
            if (q == 0 && valid) actions[base + cur_slot] = (double)a;
        }
    }
}

#endif  // __HIPCC__
}  // namespace pve
