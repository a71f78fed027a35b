// Probe (diagnostics, not part of the library): checks what csrc/pve_actor.h relies on for the 32-wide actor tile:
//   v_mfma_f32_32x32x16_f16:  A[i = lane & 31][k = 8 (lane >> 5) + e],  B[k = 8 (lane >> 5) + e][j = lane & 31],
//                             D[i = 8 (r >> 2) + 4 (lane >> 5) + (r & 3)][j = lane & 31], r = 0..15;  f16 subnormals honoured
//   v_permlane32_swap(a, b):  a.hi32 <-> b.lo32, result {a', b'}
//   v_rsq_f32 / v_exp_f32 / v_rcp_f32 accuracy on the ranges the actor uses
// Build + run on the GPU box:  hipcc --offload-arch=gfx950 tools/mfma32_layout_probe.hip -o /tmp/probe32 && /tmp/probe32
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cmath>
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef float f16v __attribute__((ext_vector_type(16)));
typedef unsigned u2 __attribute__((ext_vector_type(2)));
__global__ void k(const float *A, const float *B, float *D, unsigned *S, float *T)   // A [32][16], B [16][32], D [32][32]
{
    const int l = threadIdx.x, i = l & 31, q = l >> 5;
    h8 a, b;
    for (int e = 0; e < 8; e++) { a[e] = (_Float16)A[i * 16 + 8 * q + e]; b[e] = (_Float16)B[(8 * q + e) * 32 + i]; }
    f16v c;
    for (int r = 0; r < 16; r++) c[r] = 0.f;
    c = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0);
    for (int r = 0; r < 16; r++) D[(8 * (r >> 2) + 4 * q + (r & 3)) * 32 + i] = c[r];
    u2 s = __builtin_amdgcn_permlane32_swap(1000u + l, 2000u + l, false, false);
    S[2 * l] = s[0]; S[2 * l + 1] = s[1];
    const float x = 1e-12f + 0.37f * (float)(l * l * l);          // variance + eps range
    T[3 * l] = __builtin_amdgcn_rsqf(x);
    const float z = -9.f + 0.3f * (float)l;                       // pre-tanh range
    const float t = __expf(2.f * z);
    T[3 * l + 1] = 1.f - 2.f * __builtin_amdgcn_rcpf(t + 1.f);
    T[3 * l + 2] = tanhf(z);
}
int main()
{
    float hA[512], hB[512], hD[1024], hT[192], *dA, *dB, *dD, *dT;
    unsigned hS[128], *dS;
    for (int n = 0; n < 512; n++) { hA[n] = (float)((n * 37 % 61) - 30) / 16.f; hB[n] = (float)((n * 53 % 47) - 23) / 8.f; }
    hipMalloc(&dA, sizeof hA); hipMalloc(&dB, sizeof hB); hipMalloc(&dD, sizeof hD); hipMalloc(&dS, sizeof hS); hipMalloc(&dT, sizeof hT);
    for (int pass = 0; pass < 2; pass++) {
        if (pass == 1) for (int n = 0; n < 512; n++) hA[n] *= 1.0e-6f;      // subnormal f16 range (< 6.1e-5)
        hipMemcpy(dA, hA, sizeof hA, hipMemcpyHostToDevice); hipMemcpy(dB, hB, sizeof hB, hipMemcpyHostToDevice);
        hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, dA, dB, dD, dS, dT);
        hipMemcpy(hD, dD, sizeof hD, hipMemcpyDeviceToHost);
        double worst = 0, scale = 0;
        for (int i = 0; i < 32; i++) for (int j = 0; j < 32; j++) {
            double s = 0;
            for (int kk = 0; kk < 16; kk++) s += (double)(float)(_Float16)hA[i * 16 + kk] * (double)(float)(_Float16)hB[kk * 32 + j];
            worst = fmax(worst, fabs(s - hD[i * 32 + j])); scale = fmax(scale, fabs(s));
        }
        printf("pass %d (%s): max |D - ref| = %.3e, max |ref| = %.3e\n", pass, pass ? "subnormal A" : "normal", worst, scale);
    }
    hipMemcpy(hS, dS, sizeof hS, hipMemcpyDeviceToHost); hipMemcpy(hT, dT, sizeof hT, hipMemcpyDeviceToHost);
    int bad = 0;
    for (int l = 0; l < 64; l++) {
        const unsigned a = l < 32 ? 1000u + l : 2000u + (l - 32), b = l < 32 ? 1000u + (l + 32) : 2000u + l;
        bad += (hS[2 * l] != a) + (hS[2 * l + 1] != b);
    }
    printf("permlane32_swap(a, b): a' = [a.lo | b.lo], b' = [a.hi | b.hi]: %s (lane 0: %u %u, lane 32: %u %u)\n", bad ? "NO" : "yes", hS[0], hS[1], hS[64], hS[65]);
    double er = 0, et = 0;
    for (int l = 0; l < 64; l++) {
        const double x = (double)(1e-12f + 0.37f * (float)(l * l * l));
        er = fmax(er, fabs(hT[3 * l] * sqrt(x) - 1.0));
        et = fmax(et, fabs((double)hT[3 * l + 1] - tanh((double)(-9.f + 0.3f * (float)l))));
    }
    printf("rsq max rel err %.3e; fast tanh max abs err %.3e\n", er, et);
    return 0;
}
