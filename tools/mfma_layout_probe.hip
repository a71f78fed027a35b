// Probe (diagnostics, not part of the library): checks the operand / result layout of v_mfma_f32_16x16x32_f16 that
// csrc/pve_actor.h relies on:  A[i = lane & 15][k = 8 (lane >> 4) + e],  B[k = 8 (lane >> 4) + e][j = lane & 15],
// D[i = 4 (lane >> 4) + r][j = lane & 15];  and whether f16 subnormal inputs are honoured.
// Build + run on the GPU box:  hipcc --offload-arch=gfx950 tools/mfma_layout_probe.hip -o /tmp/probe && /tmp/probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cmath>
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef float f4 __attribute__((ext_vector_type(4)));
__global__ void k(const float *A, const float *B, float *D)   // A [16][32], B [32][16], D [16][16]
{
    const int l = threadIdx.x, i = l & 15, q = l >> 4;
    h8 a, b;
    for (int e = 0; e < 8; e++) { a[e] = (_Float16)A[i * 32 + 8 * q + e]; b[e] = (_Float16)B[(8 * q + e) * 16 + i]; }
    f4 c = {0, 0, 0, 0};
    c = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c, 0, 0, 0);
    for (int r = 0; r < 4; r++) D[(4 * q + r) * 16 + i] = c[r];
}
int main()
{
    float hA[512], hB[512], hD[256], *dA, *dB, *dD;
    for (int n = 0; n < 512; n++) { hA[n] = (float)((n * 37 % 61) - 30) / 16.f; hB[n] = (float)((n * 53 % 47) - 23) / 8.f; }
    hipMalloc(&dA, sizeof hA); hipMalloc(&dB, sizeof hB); hipMalloc(&dD, sizeof hD);
    for (int pass = 0; pass < 2; pass++) {
        if (pass == 1) for (int n = 0; n < 512; n++) hA[n] *= 1.0e-6f;      // subnormal f16 range (< 6.1e-5)
        hipMemcpy(dA, hA, sizeof hA, hipMemcpyHostToDevice); hipMemcpy(dB, hB, sizeof hB, hipMemcpyHostToDevice);
        hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, dA, dB, dD);
        hipMemcpy(hD, dD, sizeof hD, hipMemcpyDeviceToHost);
        double worst = 0, scale = 0;
        for (int i = 0; i < 16; i++) for (int j = 0; j < 16; j++) {
            double s = 0;
            for (int kk = 0; kk < 32; kk++) s += (double)(float)(_Float16)hA[i * 32 + kk] * (double)(float)(_Float16)hB[kk * 16 + j];
            worst = fmax(worst, fabs(s - hD[i * 16 + j])); scale = fmax(scale, fabs(s));
        }
        printf("pass %d (%s): max |D - ref| = %.3e, max |ref| = %.3e\n", pass, pass ? "subnormal A" : "normal", worst, scale);
    }
    return 0;
}
