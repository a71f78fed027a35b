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
// The split-half actor (default): every float32 operand is split into two halves, x = hi + lo (hi = half(x), lo =
// half(x - hi): 22 significant bits; gfx950's matrix cores honour f16 subnormals -- tools/mfma32_layout_probe.hip), and
// each product block is three v_mfma_f32_32x32x16_f16 (hi*hi + hi*lo + lo*hi, float32 accumulation; the lo*lo term is
// 2^-22 relative).  One TILE = 32 vehicles on one wave: lane (j = lane & 31, hf = lane >> 5) holds, for vehicle j,
//   inputs   x[c], c = 0..15: feature 16 (c >> 3) + 8 hf + (c & 7)  (features 28..31 = zero padding of K to 32)
//   hidden   acc[m][r], m = 0..1, r = 0..15: unit u(m, r, hf) = 32 m + 8 (r >> 2) + 4 hf + (r & 3)
// which is at once the C/D layout of one layer and -- with the contraction index enumerated accordingly -- the B-operand
// layout of the next (K-block kb = 2 m + g supplies registers r = 8 g .. 8 g + 7 of unit tile m): activations never move.
//
// What round 3 changed against the 16-wide tile of round 2 (VALU issue is what bounds the actor, not the matrix pipe):
//  * the dense kernels are stored CENTERED over their output units (W[k][u] - mean_u W[k][.], b[u] - mean b): the matrix
//    cores then deliver h - mean(h) directly and LayerNorm's mean pass (sum, exchange, 64 subtractions) disappears;
//  * a vehicle lives in 2 lanes instead of 4: one v_permlane32_swap per LayerNorm sum instead of two exchanges;
//  * 1 / sqrt(var + eps) is v_rsq_f32, 3 tanh(z) = 3 - 6 / (exp(2 z) + 1) with v_exp_f32 / v_rcp_f32 (1 ulp each; measured
//    3.6e-7 absolute on the action), divisions by the layer widths are multiplications;
//  * every weight is centered, split and laid out in A-operand order ONCE (k_actor_pack, pve_set_actor) instead of by
//    every workgroup of every launch.
// The same device function (`actor_tile32`) is called by the stand-alone kernel k_actor_h (weights staged in LDS by persistent
// workgroups) and from inside k_rollout (pve_step_many(PVE_SRC_ACTOR): the closed loop resident on the chip, A operands
// streamed from L1 / L2), so the two paths are bit-identical by construction.
// Not bit-identical to the float32 chain of k_actor_t (actor_canonical); the action parity bar is 5e-4
// (tests/actor_scenarios.py).  PVE_CFG_ACTOR_F32 selects k_actor_t.
typedef _Float16 pve_v8h __attribute__((ext_vector_type(8)));
typedef float pve_v16f __attribute__((ext_vector_type(16)));
typedef float pve_v2f __attribute__((ext_vector_type(2)));

// packed actor parameters (device buffer written by k_actor_pack), byte offsets
constexpr int AP_A1 = 0;                                   // _Float16 [hl 2][m 2][kb 2][lane 64][8]
constexpr int AP_A2 = AP_A1 + 2 * 2 * 2 * 64 * 8 * 2;      // _Float16 [hl 2][m2 2][kb 4][lane 64][8]
constexpr int AP_PRM = AP_A2 + 2 * 2 * 4 * 64 * 8 * 2;     // float [PV_TOTAL]
// float parameter vectors in LANE ORDER: vec[(hf * 2 + m) * 16 + r] = parameter of unit u(m, r, hf)
constexpr int PV_B1 = 0, PV_G1 = 64, PV_BE1 = 128, PV_B2 = 192, PV_G2 = 256, PV_BE2 = 320, PV_W3 = 384,
              PV_LN0G = 448 /* [hf][c 16] */, PV_LN0B = 480, PV_B3 = 512, PV_A0 = 513 /* action of the all-zero row */,
              PV_TOTAL = 516;
constexpr int AP_BYTES = AP_PRM + PV_TOTAL * 4;
constexpr int AP_BYTES_PADDED = (AP_BYTES + 255) / 256 * 256;
static_assert(AP_PRM == 24576 && AP_BYTES_PADDED == (int)ACTOR_PACKED_BYTES && AW_TOTAL * 4 <= (int)ACTOR_FLAT_BYTES, "packed actor layout");

__device__ __forceinline__ int actor_unit(int m, int r, int hf) { return 32 * m + 8 * (r >> 2) + 4 * hf + (r & 3); }
__device__ __forceinline__ int actor_feature(int c, int hf) { return 16 * (c >> 3) + 8 * hf + (c & 7); }

// sum over the two lanes (j, j + 32) that hold one vehicle: lo + hi in every lane (one gfx950 row swap, VALU)
__device__ __forceinline__ float actor_xsum2(float s)
{
    const unsigned u = __float_as_uint(s);
    const pve_v2u r = __builtin_amdgcn_permlane32_swap(u, u, false, false);   // r[0] = [lo | lo], r[1] = [hi | hi]
    return __uint_as_float(r[0]) + __uint_as_float(r[1]);
}
// x = hi + lo in half precision: hi = half(x) (v_cvt_pk_f16_f32, two values per instruction), lo = half(x - hi) with the
// mixed-precision FMA (v_fma_mixlo / mixhi_f16: fma(x, 1.0, -hi) evaluated in float32 -- exact, x - hi has <= 13 significant
// bits -- and rounded once to half, straight into its half of the packed register): 1.5 instructions per value instead of
// 3.5 (convert back, subtract, convert).  Same values as `lo = (_Float16)(x - (float)hi)`.
// HAZARD: the compiler does not know that these inline-asm statements are vector instructions, so its hazard recognizer
// inserts no wait states between them and a matrix instruction that reads their result (gfx90a+: 2 wait states between a
// vector write of a register and an MFMA reading it) -- without the closing `s_nop 1`, which is tied to the results as
// operands and therefore sits between the last write and the first matrix read, the two kernels that inline this function
// returned schedule-dependent (different) values.
typedef _Float16 pve_v2h __attribute__((ext_vector_type(2)));
__device__ __forceinline__ void actor_split8(const float *x, pve_v8h &hi, pve_v8h &lo)
{
    typedef unsigned pve_v4u __attribute__((ext_vector_type(4)));
    pve_v4u hq, lq;
#pragma unroll
    for (int e = 0; e < 8; e += 2) {
        unsigned hp, lp;
        asm("v_cvt_pk_f16_f32 %0, %1, %2" : "=v"(hp) : "v"(x[e]), "v"(x[e + 1]));
        asm("v_fma_mixlo_f16 %0, %1, 1.0, -%2 op_sel_hi:[0,0,1]" : "=v"(lp) : "v"(x[e]), "v"(hp));
        asm("v_fma_mixhi_f16 %0, %1, 1.0, -%2 op_sel:[0,0,1] op_sel_hi:[0,0,1]" : "+v"(lp) : "v"(x[e + 1]), "v"(hp));
        hq[e >> 1] = hp; lq[e >> 1] = lp;
    }
    asm("s_nop 1" : "+v"(hq), "+v"(lq));
    hi = __builtin_bit_cast(pve_v8h, hq); lo = __builtin_bit_cast(pve_v8h, lq);
}
// 3 tanh(z): exp(2 z) overflows to +inf -> 3, underflows to 0 -> -3
__device__ __forceinline__ float actor_tanh3(float z)
{
    const float t = __expf(2.0f * z);
    return 3.0f - 6.0f * __builtin_amdgcn_rcpf(t + 1.0f);
}

// Sum of the 16 values of a lane by HALVES: (e[0..7] + e[8..15]) -> 8, -> 4, -> 2, -> 1.  Every level is a packed float32 add
// on register pairs that are adjacent already (4 + 2 + 1 v_pk_add_f32 + 1 v_add_f32 = 8 instructions, no moves); the
// adjacent-pair tree ((e0 + e1) + (e2 + e3)) + ... compiled to the same 8 packed adds plus ~17 v_mov_b32 that built the pairs.
typedef float pve_v8f __attribute__((ext_vector_type(8)));
__device__ __forceinline__ float actor_hsum16(pve_v16f e)
{
    const pve_v8f a = __builtin_shufflevector(e, e, 0, 1, 2, 3, 4, 5, 6, 7) + __builtin_shufflevector(e, e, 8, 9, 10, 11, 12, 13, 14, 15);
    const pve_v4f b = __builtin_shufflevector(a, a, 0, 1, 2, 3) + __builtin_shufflevector(a, a, 4, 5, 6, 7);
    const pve_v2f c = __builtin_shufflevector(b, b, 0, 1) + __builtin_shufflevector(b, b, 2, 3);
    return c[0] + c[1];
}

// LayerNorm (input already centered by the centered dense kernel) + ReLU over the 64 hidden units of a vehicle, 32 of
// them in this lane; gam / bet point at this lane's 32 parameters (lane order)
__device__ __forceinline__ void actor_ln_relu32(pve_v16f (&v)[2], const float *gam, const float *bet)
{
    pve_v16f e = v[0] * v[0];
    e = __builtin_elementwise_fma(v[1], v[1], e);
    const float s = actor_hsum16(e);
    const float rstd = __builtin_amdgcn_rsqf(actor_xsum2(s) * (1.0f / (float)ACT_H) + 1e-12f);
#pragma unroll
    for (int m = 0; m < 2; m++) {
        const pve_v16f ga = *(const pve_v16f *)(gam + 16 * m), be = *(const pve_v16f *)(bet + 16 * m);
        const pve_v16f y = __builtin_elementwise_fma(v[m], ga * rstd, be);
        pve_v16f z;
#pragma unroll
        for (int r = 0; r < 16; r++) z[r] = fmaxf(y[r], 0.f);
        v[m] = z;
    }
}

// One tile: x[16] = this lane's raw observation features (layout above) -> the action of vehicle j (in both of its lanes).
// A1 / A2: the packed A operands (LDS or global memory), prm: the packed float parameters (LDS).
// The 12 product blocks (layer 1: unit tile m x K-block kb; layer 2: m2 x kb) run as one software pipeline: the operand pair
// (hi, lo) of block s + 1 is requested while block s multiplies (three matrix instructions, ~100 cycles -- about an L1 round
// trip when the operands stream from global memory in k_rollout); two pairs (16 registers) in flight.  A ring of three
// (request s + 2) covered an L2 round trip too but cost 8 more registers: inside k_rollout that was 34 instead of 12 spilled
// registers and 38.1 instead of 36.6 us per closed-loop step.  The opaque offset `o` pins every request to its place:
// without it the compiler hoists all 24 operand loads to the top (96 registers) and spills.  Unit tiles are processed one
// after the other, all K-blocks of the input are split into half pairs up front.
// (ob = the lane's BYTE offset inside a block, 16 x lane, as an unsigned 32-bit value: uniform base + 32-bit lane offset +
//  immediate is the address form of a global load; an `int` element index is a sign extension and a 64-bit add per request)
__device__ __forceinline__ pve_v8h actor_a_operand(const pve_v8h *A1, const pve_v8h *A2, int s, int hl, unsigned ob)
{   // block s: 0..3 = layer 1 (m = s >> 1, kb = s & 1), 4..11 = layer 2 (m2 = (s - 4) >> 2, kb = (s - 4) & 3)
    const pve_v8h *blk = s < 4 ? A1 + ((hl * 2 + (s >> 1)) * 2 + (s & 1)) * 64 : A2 + ((hl * 2 + ((s - 4) >> 2)) * 4 + ((s - 4) & 3)) * 64;
#ifdef PVE_WIDE_INDEX                              // A/B build knob: element index, 64-bit address per request
    return blk[(int)(ob >> 4)];
#else
    return *(const pve_v8h *)((const char *)blk + ob);
#endif
}
__device__ __forceinline__ float actor_tile32(const pve_v8h *A1, const pve_v8h *A2, const float *prm, const float (&x)[16],
                                              int lane)
{
    const int hf = lane >> 5;
    unsigned o = (unsigned)lane * 16u;
    asm volatile("" : "+v"(o));
    pve_v8h ah[2], al[2];                                     // operand ring
    ah[0] = actor_a_operand(A1, A2, 0, 0, o); al[0] = actor_a_operand(A1, A2, 0, 1, o);
    // ---- LayerNorm over the 28 inputs (14 + 14 of them in the two lanes; the padding entries are zeros)
    pve_v16f d;
#pragma unroll
    for (int c = 0; c < 16; c++) d[c] = x[c];
    const float mean = actor_xsum2(actor_hsum16(d)) * (1.0f / (float)ACT_IN);
    d = d - mean;
    if (hf) { d[12] = 0.f; d[13] = 0.f; d[14] = 0.f; d[15] = 0.f; }   // features 28..31 do not exist
    const float var = actor_hsum16(d * d);
    const float rstd0 = __builtin_amdgcn_rsqf(actor_xsum2(var) * (1.0f / (float)ACT_IN) + 1e-12f);
    pve_v8h bh[4], bl[4];                                     // B operands of the current layer: K-blocks as half pairs
    {
        const pve_v16f ga = *(const pve_v16f *)(prm + PV_LN0G + 16 * hf), be = *(const pve_v16f *)(prm + PV_LN0B + 16 * hf);
        const pve_v16f yv = __builtin_elementwise_fma(d, ga * rstd0, be);
        float y[16];
#pragma unroll
        for (int c = 0; c < 16; c++) y[c] = yv[c];
        actor_split8(y, bh[0], bl[0]);
        actor_split8(y + 8, bh[1], bl[1]);
    }
    pve_v16f h[2], g[2];
#pragma unroll
    for (int s = 0; s < 12; s++) {
        if (s + 1 < 12) {                                     // request block s + 1
            asm volatile("" : "+v"(o));
            ah[(s + 1) % 2] = actor_a_operand(A1, A2, s + 1, 0, o); al[(s + 1) % 2] = actor_a_operand(A1, A2, s + 1, 1, o);
        }
        if (s == 4) {
            // ---- LayerNorm_1 + ReLU; K-block kb = 2 m + g of the next layer is exactly registers 8 g .. 8 g + 7 of h[m]
            actor_ln_relu32(h, prm + PV_G1 + hf * 32, prm + PV_BE1 + hf * 32);
#pragma unroll
            for (int kb = 0; kb < 4; kb++) {
                float hv[8];
#pragma unroll
                for (int e = 0; e < 8; e++) hv[e] = h[kb >> 1][8 * (kb & 1) + e];
                actor_split8(hv, bh[kb], bl[kb]);
            }
        }
        const int layer2 = s >= 4, m = layer2 ? (s - 4) >> 2 : s >> 1, kb = layer2 ? (s - 4) & 3 : s & 1;
        pve_v16f &acc = layer2 ? g[m] : h[m];
        if (kb == 0) acc = *(const pve_v16f *)(prm + (layer2 ? PV_B2 : PV_B1) + (hf * 2 + m) * 16);   // centered bias
        acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[s % 2], bh[kb], acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[s % 2], bl[kb], acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(al[s % 2], bh[kb], acc, 0, 0, 0);
    }
    actor_ln_relu32(g, prm + PV_G2 + hf * 32, prm + PV_BE2 + hf * 32);
    // ---- dense 64 -> 1, 3 tanh
    pve_v16f pv = g[0] * *(const pve_v16f *)(prm + PV_W3 + (hf * 2 + 0) * 16);
    pv = __builtin_elementwise_fma(g[1], *(const pve_v16f *)(prm + PV_W3 + (hf * 2 + 1) * 16), pv);
    return actor_tanh3(actor_xsum2(actor_hsum16(pv)) + prm[PV_B3]);
}

// The <= 16 raw features lane (j, hf) contracts, straight from the observation row of `slot` (float32 or float64 rows):
// features 8 hf .. 8 hf + 7 and 16 + 8 hf .. 16 + 8 hf + 7 (28 .. 31 do not exist: zeros) -- four 16-byte loads per lane
// for float32 rows, every row read by exactly two lanes
template <typename OBS_T>
__device__ __forceinline__ void actor_fetch(const OBS_T *rows, size_t row, int hf, float (&x)[16])
{
    const OBS_T *src = rows + row * OBSW + 8 * hf;
#pragma unroll
    for (int c = 0; c < 16; c++) x[c] = (c < 12 || !hf) ? (float)src[actor_feature(c, 0)] : 0.f;
}
// the same from the rows of ONE intersection (k_rollout's actor pass)
template <typename OBS_T>
__device__ __forceinline__ void actor_fetch_env(const OBS_T *env_rows, int slot, int hf, float (&x)[16])
{
    // (env_rows: the intersection's first row, uniform; the lane's part as a 32-bit byte offset: cf. actor_a_operand)
    const OBS_T *src = (const OBS_T *)((const char *)env_rows + (unsigned)((slot * OBSW + 8 * hf) * (int)sizeof(OBS_T)));
#pragma unroll
    for (int c = 0; c < 16; c++) x[c] = (c < 12 || !hf) ? (float)src[actor_feature(c, 0)] : 0.f;
}

// pve_set_actor: flat float32 weights -> the packed buffer (one workgroup of 256 threads)
__global__ __launch_bounds__(256) void k_actor_pack(const float *__restrict__ W, unsigned char *__restrict__ packed)
{
    __shared__ float cm1[ACT_IN], cm2[ACT_H], bm[2];
    const int tid = threadIdx.x;
    // means over the OUTPUT units (what LayerNorm subtracts): per input row of each dense kernel, and of the biases
    if (tid < ACT_IN) { double s = 0; for (int u = 0; u < ACT_H; u++) s += (double)W[AW_W1 + tid * ACT_H + u]; cm1[tid] = (float)(s / ACT_H); }
    if (tid >= 64 && tid < 64 + ACT_H) { const int k = tid - 64; double s = 0; for (int u = 0; u < ACT_H; u++) s += (double)W[AW_W2 + k * ACT_H + u]; cm2[k] = (float)(s / ACT_H); }
    if (tid == 128 || tid == 129) { const int o = tid == 128 ? AW_B1 : AW_B2; double s = 0; for (int u = 0; u < ACT_H; u++) s += (double)W[o + u]; bm[tid - 128] = (float)(s / ACT_H); }
    __syncthreads();
    pve_v8h *A1 = (pve_v8h *)(packed + AP_A1), *A2 = (pve_v8h *)(packed + AP_A2);
    float *prm = (float *)(packed + AP_PRM);
    for (int n = tid; n < (4 + 8) * 64; n += 256) {           // one thread per operand vector (8 halves of one lane)
        const int l = n & 63, t = n >> 6, hf = l >> 5, i = l & 31;
        float w[8];
        if (t < 4) {                                          // layer 1: m = t >> 1, kb = t & 1: A[i][8 hf + e] = W1c[16 kb + 8 hf + e][32 m + i]
            const int m = t >> 1, kb = t & 1;
#pragma unroll
            for (int e = 0; e < 8; e++) { const int k = 16 * kb + 8 * hf + e; w[e] = k < ACT_IN ? W[AW_W1 + k * ACT_H + 32 * m + i] - cm1[k] : 0.f; }
            pve_v8h hi, lo;
            actor_split8(w, hi, lo);
            A1[((0 * 2 + m) * 2 + kb) * 64 + l] = hi; A1[((1 * 2 + m) * 2 + kb) * 64 + l] = lo;
        } else {                                              // layer 2: m2, kb: A[i][8 hf + e] = W2c[u(kb >> 1, 8 (kb & 1) + e, hf)][32 m2 + i]
            const int m2 = (t - 4) >> 2, kb = (t - 4) & 3;
#pragma unroll
            for (int e = 0; e < 8; e++) { const int k = actor_unit(kb >> 1, 8 * (kb & 1) + e, hf); w[e] = W[AW_W2 + k * ACT_H + 32 * m2 + i] - cm2[k]; }
            pve_v8h hi, lo;
            actor_split8(w, hi, lo);
            A2[((0 * 2 + m2) * 4 + kb) * 64 + l] = hi; A2[((1 * 2 + m2) * 4 + kb) * 64 + l] = lo;
        }
    }
    if (tid < 64) {                                           // parameter vectors in lane order
        const int hf = tid >> 5, m = (tid >> 4) & 1, r = tid & 15, u = actor_unit(m, r, hf);
        prm[PV_B1 + tid] = W[AW_B1 + u] - bm[0]; prm[PV_G1 + tid] = W[AW_LN1_G + u]; prm[PV_BE1 + tid] = W[AW_LN1_B + u];
        prm[PV_B2 + tid] = W[AW_B2 + u] - bm[1]; prm[PV_G2 + tid] = W[AW_LN2_G + u]; prm[PV_BE2 + tid] = W[AW_LN2_B + u];
        prm[PV_W3 + tid] = W[AW_W3 + u];
    }
    if (tid < 32) {
        const int hf = tid >> 4, c = tid & 15, f = actor_feature(c, hf);
        prm[PV_LN0G + tid] = f < ACT_IN ? W[AW_LN0_G + f] : 0.f; prm[PV_LN0B + tid] = f < ACT_IN ? W[AW_LN0_B + f] : 0.f;
    }
    if (tid == 0) { prm[PV_B3] = W[AW_B3]; prm[PV_A0] = 0.f; prm[PV_A0 + 1] = 0.f; prm[PV_A0 + 2] = 0.f; }
    __threadfence();
    __syncthreads();
    if (tid < 64) {                                           // the action of an all-zero row (what a newly spawned vehicle gets)
        float x[16];
#pragma unroll
        for (int c = 0; c < 16; c++) x[c] = 0.f;
        const float a0 = actor_tile32(A1, A2, prm, x, tid);
        if (tid == 0) prm[PV_A0] = a0;
    }
}

// Stand-alone actor pass (pve_actor_forward / pve_step_all_actor): persistent workgroups of 4 waves sharing ONE copy of the
// packed parameters in LDS; every wave is on its own: it loops over intersections, compacts the controlled vehicles of its
// intersection with ballots and runs their tiles of 32 (lane (j, hf) fetches the half row it contracts).
template <int CAP, typename OBS_T>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(4, 4))) void k_actor_h(const unsigned char *__restrict__ packed, const OBS_T *__restrict__ obs,
                                                 const int32_t *__restrict__ meta, double *__restrict__ actions,
                                                 int n_envs)
{
    __shared__ __attribute__((aligned(64))) unsigned char sp[AP_BYTES_PADDED];
    __shared__ unsigned char slot_of_s[4][CAP];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    unsigned char *slot_of = slot_of_s[wave];
    const int stride = gridDim.x * 4;
    int env = blockIdx.x * 4 + wave;
    int mt[CAP / 64];                                         // flags of the wave's next intersection (loaded one ahead)
#pragma unroll
    for (int sub = 0; sub < CAP / 64; sub++) mt[sub] = env < n_envs ? meta[(size_t)env * CAP + sub * 64 + lane] : 0;
    for (int n = tid; n < AP_BYTES_PADDED / 16; n += 256) ((uint4 *)sp)[n] = ((const uint4 *)packed)[n];
    __syncthreads();                                          // parameters staged
    const pve_v8h *A1 = (const pve_v8h *)(sp + AP_A1), *A2 = (const pve_v8h *)(sp + AP_A2);
    const float *prm = (const float *)(sp + AP_PRM);
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
        for (int v0 = 0; v0 < nctl; v0 += 32) {              // one tile = 32 vehicles, two lanes each
            int wo = 0;
            asm volatile("" : "+v"(wo));                      // (keeps the parameter reads inside the loop)
            const int j = lane & 31, hf = lane >> 5;
            const bool valid = v0 + j < nctl;
            const int slot = slot_of[valid ? v0 + j : 0];
            float x[16];
            actor_fetch(obs, base + slot, hf, x);
            const float a = actor_tile32(A1, A2, prm + wo, x, lane);
            if (valid && hf == 0) actions[base + slot] = (double)a;
        }
    }
}

#endif  // __HIPCC__
}  // namespace pve
