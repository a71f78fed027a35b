// pve_actor.h -- MADDPG actor inference for every controlled vehicle of every intersection
// (reference model_agent_maddpg.py:23-49, called per vehicle with batch 1 from main.py:36-45, 404).
//
//   x(28) -> LayerNorm -> Dense 28x64 -> LayerNorm -> ReLU -> Dense 64x64 -> LayerNorm -> ReLU -> Dense 64x1 -> 3*tanh
//
// float32 like the TF graph (placeholder dtype, :15).  One wave64 per intersection; lane r owns the
// r-th controlled vehicle (ballot + popcount compaction, so no lane is wasted on empty / exit-leg slots).
// The weights are wave-uniform: they are fetched with scalar loads and fed to v_fmac_f32 as the SGPR operand,
// the activations of the previous layer are staged in LDS ([k][lane], conflict-free) so the k-loop stays rolled
// and the 64 accumulators of a layer live in registers.  No MFMA: f32-input MFMA runs at the f32 vector rate on
// gfx950, and M = controlled vehicles of one env (<= 64) is too ragged to tile across envs without a gather.
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

#if defined(__HIPCC__)

template <int N>
__device__ __forceinline__ void layer_norm_relu(float (&h)[N], const float *gamma, const float *beta, bool relu)
{   // tc.layers.layer_norm: biased variance over the last axis, eps = 1e-12; y = x*inv + (beta - mean*inv)
    float s = 0.f;
#pragma unroll
    for (int k = 0; k < N; k++) s += h[k];
    const float mean = s / (float)N;
    float v = 0.f;
#pragma unroll
    for (int k = 0; k < N; k++) { const float d = h[k] - mean; v = fmaf(d, d, v); }
    const float rstd = 1.0f / sqrtf(v / (float)N + 1e-12f);
#pragma unroll
    for (int k = 0; k < N; k++) {
        const float inv = rstd * gamma[k];
        float y = fmaf(h[k], inv, beta[k] - mean * inv);
        h[k] = relu ? fmaxf(y, 0.f) : y;
    }
}

// acc[0..63] += x_i * row_i[0..63] for NROWS staged inputs; the row is wave-uniform (scalar loads)
#define PVE_ACTOR_DENSE(ACC, NROWS, WBASE, STAGE_ROW0)                                        \
    for (int i = 0; i < (NROWS); i++) {                                                       \
        const float xi = stage[(STAGE_ROW0) + i][lane];                                       \
        const float *__restrict__ wr = W + (WBASE) + i * ACT_H;                               \
        _Pragma("unroll") for (int j = 0; j < ACT_H; j++) ACC[j] = fmaf(xi, wr[j], ACC[j]);   \
    }

template <int CAP, typename OBS_T>
__global__ __launch_bounds__(64) void k_actor(const float *__restrict__ W, const OBS_T *__restrict__ obs,
                                              const int32_t *__restrict__ meta, double *__restrict__ actions,
                                              int n_envs)
{
    __shared__ float stage[32][64];           // 32 activations of the previous layer at a time, [k][lane]
    __shared__ unsigned char slot_of[CAP];
    const int env = blockIdx.x, lane = threadIdx.x;
    const size_t base = (size_t)env * CAP;
    int nctl = 0;
#pragma unroll
    for (int sub = 0; sub < CAP / 64; sub++) {
        const int s = sub * 64 + lane;
        const int m = meta[base + s];
        const bool c = (m & (M_ALIVE | M_CONTROL)) == (M_ALIVE | M_CONTROL);
        const unsigned long long b = __ballot(c);
        const int rank = nctl + __builtin_popcountll(b & ((1ull << lane) - 1ull));
        if (c) slot_of[rank] = (unsigned char)s;
        else actions[base + s] = 0.0;                     // main.py:401: uncontrolled vehicles get 0
        nctl += __builtin_popcountll(b);
    }
    __syncthreads();
    for (int r0 = 0; r0 < nctl; r0 += 64) {
        const bool active = r0 + lane < nctl;
        const int slot = active ? slot_of[r0 + lane] : slot_of[r0];
        // ---- input row (veh["state"][0], float64 in HBM -> float32 like the TF placeholder)
        float x[ACT_IN];
        {
            const OBS_T *row = obs + (base + slot) * OBSW;
#pragma unroll
            for (int k = 0; k < ACT_IN; k++) x[k] = (float)row[k];
        }
        layer_norm_relu<ACT_IN>(x, W + AW_LN0_G, W + AW_LN0_B, false);
#pragma unroll
        for (int k = 0; k < ACT_IN; k++) stage[k][lane] = x[k];
        // ---- dense 28 -> 64
        float h[ACT_H];
#pragma unroll
        for (int j = 0; j < ACT_H; j++) h[j] = W[AW_B1 + j];
        PVE_ACTOR_DENSE(h, ACT_IN, AW_W1, 0)
        layer_norm_relu<ACT_H>(h, W + AW_LN1_G, W + AW_LN1_B, true);
        // ---- dense 64 -> 64, the 64 inputs staged 32 at a time (the lane's LDS column is private: DS
        //      operations of one wave execute in order, no barrier needed)
        float g[ACT_H];
#pragma unroll
        for (int j = 0; j < ACT_H; j++) g[j] = W[AW_B2 + j];
#pragma unroll
        for (int k = 0; k < 32; k++) stage[k][lane] = h[k];
        PVE_ACTOR_DENSE(g, 32, AW_W2, 0)
#pragma unroll
        for (int k = 0; k < 32; k++) stage[k][lane] = h[32 + k];
        PVE_ACTOR_DENSE(g, 32, AW_W2 + 32 * ACT_H, 0)
        layer_norm_relu<ACT_H>(g, W + AW_LN2_G, W + AW_LN2_B, true);
        // ---- dense 64 -> 1, 3*tanh
        float y = W[AW_B3];
#pragma unroll
        for (int j = 0; j < ACT_H; j++) y = fmaf(g[j], W[AW_W3 + j], y);
        const float a = 3.0f * tanhf(y);
        if (active) actions[base + slot] = (double)a;
    }
}

// ------------------------------------------------------------------------------------------------------
// k_actor_mfma: the same network on the matrix cores.  v_mfma_f32_32x32x2_f32 is an exact float32 FMA chain
// (k-ordered, one rounding per product: bitwise the order of the VALU kernel above), at the f32 vector rate,
// but it takes its operands per lane: A[i = lane & 31][k = lane >> 5] and B[k = lane >> 5][j = lane & 31].
// That removes every wave-uniform operand: the 28x64 + 64x64 weights live in 92 VGPRs per lane for the whole
// kernel (loaded once, coalesced), activations are exchanged between the C layout (column per lane) and the A
// layout (row per lane) through a padded 64x65 LDS tile, and the LayerNorm statistics of a vehicle are a sum
// over the 14 / 32 features its two lanes (lane, lane ^ 32) hold plus one cross-lane exchange.
// One wave64 per intersection; 64 controlled vehicles (2 row tiles of 32) per pass; 184 MFMAs per pass.
typedef float pve_v16f __attribute__((ext_vector_type(16)));

template <int CAP, typename OBS_T>
__global__ __launch_bounds__(64) void k_actor_mfma(const float *__restrict__ W, const OBS_T *__restrict__ obs,
                                                   const int32_t *__restrict__ meta, double *__restrict__ actions,
                                                   int n_envs)
{
    __shared__ float Hs[32][65];               // activations of 32 vehicles x 64 features (+1 pad: conflict-free)
    __shared__ float lnp[2 * ACT_IN + 4 * ACT_H + ACT_H];   // g0,b0 | g1,b1 | g2,b2 | w3
    __shared__ unsigned char slot_of[CAP];
    const int lane = threadIdx.x, lo = lane & 31, hi = lane >> 5;
    for (int k = lane; k < 2 * ACT_IN; k += 64) lnp[k] = W[AW_LN0_G + k];
    for (int k = lane; k < 2 * ACT_H; k += 64) {
        lnp[2 * ACT_IN + k] = W[AW_LN1_G + k];
        lnp[2 * ACT_IN + 2 * ACT_H + k] = W[AW_LN2_G + k];
    }
    lnp[2 * ACT_IN + 4 * ACT_H + lane] = W[AW_W3 + lane];
    // weight fragments, loaded once per (persistent) wave: B[k = 2s + hi][j = 32nt + lo]
    float B1[ACT_IN / 2][2], B2[ACT_H / 2][2], b1c[2], b2c[2];
#pragma unroll
    for (int s = 0; s < ACT_IN / 2; s++)
#pragma unroll
        for (int nt = 0; nt < 2; nt++) B1[s][nt] = W[AW_W1 + (2 * s + hi) * ACT_H + 32 * nt + lo];
#pragma unroll
    for (int s = 0; s < ACT_H / 2; s++)
#pragma unroll
        for (int nt = 0; nt < 2; nt++) B2[s][nt] = W[AW_W2 + (2 * s + hi) * ACT_H + 32 * nt + lo];
#pragma unroll
    for (int nt = 0; nt < 2; nt++) { b1c[nt] = W[AW_B1 + 32 * nt + lo]; b2c[nt] = W[AW_B2 + 32 * nt + lo]; }
    const float b3 = W[AW_B3];
    const float *g0 = lnp, *be0 = lnp + ACT_IN, *g1 = lnp + 2 * ACT_IN, *be1 = g1 + ACT_H, *g2 = be1 + ACT_H,
                *be2 = g2 + ACT_H, *w3 = be2 + ACT_H;
    __syncthreads();
    for (int env = blockIdx.x; env < n_envs; env += gridDim.x) {
        const size_t base = (size_t)env * CAP;
        // controlled-vehicle compaction
        int nctl = 0;
#pragma unroll
        for (int sub = 0; sub < CAP / 64; sub++) {
            const int s = sub * 64 + lane;
            const int m = meta[base + s];
            const bool c = (m & (M_ALIVE | M_CONTROL)) == (M_ALIVE | M_CONTROL);
            const unsigned long long b = __ballot(c);
            const int rank = nctl + __builtin_popcountll(b & ((1ull << lane) - 1ull));
            if (c) slot_of[rank] = (unsigned char)s;
            else actions[base + s] = 0.0;                 // main.py:401: uncontrolled vehicles get 0
            nctl += __builtin_popcountll(b);
        }
        __syncthreads();
        for (int v0 = 0; v0 < nctl; v0 += 32) {           // one row tile = 32 vehicles per pass
            const int v = v0 + lo;
            const bool valid = v < nctl;
            const int slot = slot_of[valid ? v : v0];
            pve_v16f acc[2];
            // ---- layer 1: LayerNorm(28) -> dense 28x64.  Lane holds features k = 2s + hi of vehicle v0 + lo.
            {
                const OBS_T *row = obs + (base + slot) * OBSW;
                float x[ACT_IN / 2], sum = 0.f;
#pragma unroll
                for (int s = 0; s < ACT_IN / 2; s++) { x[s] = (float)row[2 * s + hi]; sum += x[s]; }
                const float mean = (sum + __shfl_xor(sum, 32)) / (float)ACT_IN;
                float var = 0.f;
#pragma unroll
                for (int s = 0; s < ACT_IN / 2; s++) { const float d = x[s] - mean; var = fmaf(d, d, var); }
                const float rstd = 1.0f / sqrtf((var + __shfl_xor(var, 32)) / (float)ACT_IN + 1e-12f);
#pragma unroll
                for (int nt = 0; nt < 2; nt++)
#pragma unroll
                    for (int q = 0; q < 16; q++) acc[nt][q] = b1c[nt];
#pragma unroll
                for (int s = 0; s < ACT_IN / 2; s++) {
                    const float inv = rstd * g0[2 * s + hi];
                    const float a = fmaf(x[s], inv, be0[2 * s + hi] - mean * inv);
#pragma unroll
                    for (int nt = 0; nt < 2; nt++)
                        acc[nt] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, B1[s][nt], acc[nt], 0, 0, 0);
                }
            }
            // ---- C layout -> LDS [vehicle][feature]: lane holds column 32nt + lo, rows (q&3) + 8(q>>2) + 4hi
#pragma unroll
            for (int nt = 0; nt < 2; nt++)
#pragma unroll
                for (int q = 0; q < 16; q++) Hs[(q & 3) + 8 * (q >> 2) + 4 * hi][32 * nt + lo] = acc[nt][q];
            __syncthreads();
            // ---- layer 2: LayerNorm_1 -> ReLU -> dense 64x64
            {
                float raw[ACT_H / 2], sum = 0.f;
#pragma unroll
                for (int s = 0; s < ACT_H / 2; s++) { raw[s] = Hs[lo][2 * s + hi]; sum += raw[s]; }
                const float mean = (sum + __shfl_xor(sum, 32)) / (float)ACT_H;
                float var = 0.f;
#pragma unroll
                for (int s = 0; s < ACT_H / 2; s++) { const float d = raw[s] - mean; var = fmaf(d, d, var); }
                const float rstd = 1.0f / sqrtf((var + __shfl_xor(var, 32)) / (float)ACT_H + 1e-12f);
#pragma unroll
                for (int nt = 0; nt < 2; nt++)
#pragma unroll
                    for (int q = 0; q < 16; q++) acc[nt][q] = b2c[nt];
#pragma unroll
                for (int s = 0; s < ACT_H / 2; s++) {
                    const float inv = rstd * g1[2 * s + hi];
                    const float a = fmaxf(fmaf(raw[s], inv, be1[2 * s + hi] - mean * inv), 0.f);
#pragma unroll
                    for (int nt = 0; nt < 2; nt++)
                        acc[nt] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, B2[s][nt], acc[nt], 0, 0, 0);
                }
            }
            __syncthreads();
#pragma unroll
            for (int nt = 0; nt < 2; nt++)
#pragma unroll
                for (int q = 0; q < 16; q++) Hs[(q & 3) + 8 * (q >> 2) + 4 * hi][32 * nt + lo] = acc[nt][q];
            __syncthreads();
            // ---- layer 3: LayerNorm_2 -> ReLU -> dense 64x1 -> 3*tanh
            {
                float raw[ACT_H / 2], sum = 0.f;
#pragma unroll
                for (int s = 0; s < ACT_H / 2; s++) { raw[s] = Hs[lo][2 * s + hi]; sum += raw[s]; }
                const float mean = (sum + __shfl_xor(sum, 32)) / (float)ACT_H;
                float var = 0.f;
#pragma unroll
                for (int s = 0; s < ACT_H / 2; s++) { const float d = raw[s] - mean; var = fmaf(d, d, var); }
                const float rstd = 1.0f / sqrtf((var + __shfl_xor(var, 32)) / (float)ACT_H + 1e-12f);
                float part = 0.f;
#pragma unroll
                for (int s = 0; s < ACT_H / 2; s++) {
                    const float inv = rstd * g2[2 * s + hi];
                    const float y = fmaxf(fmaf(raw[s], inv, be2[2 * s + hi] - mean * inv), 0.f);
                    part = fmaf(y, w3[2 * s + hi], part);
                }
                const float y = part + __shfl_xor(part, 32) + b3;
                const float a = 3.0f * tanhf(y);
                if (hi == 0 && valid) actions[base + slot] = (double)a;
            }
            __syncthreads();                    // Hs is rewritten by the next pass
        }
        __syncthreads();                        // slot_of is rewritten for the next environment
    }
}

#endif  // __HIPCC__
}  // namespace pve
