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

template <int CAP>
__global__ __launch_bounds__(64) void k_actor(const float *__restrict__ W, const double *__restrict__ obs,
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
            const double *row = obs + (base + slot) * OBSW;
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

#endif  // __HIPCC__
}  // namespace pve
