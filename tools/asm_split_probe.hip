#include <hip/hip_runtime.h>
#include <cstdio>
#include <cmath>
#include <cstring>
typedef _Float16 h2 __attribute__((ext_vector_type(2)));
__global__ void k(const float *x, unsigned *out, int n)
{
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (2 * i + 1 >= n) return;
    float a = x[2 * i], b = x[2 * i + 1];
    unsigned hp, lp;
    asm volatile("v_cvt_pk_f16_f32 %0, %1, %2" : "=v"(hp) : "v"(a), "v"(b));
    asm volatile("v_fma_mixlo_f16 %0, %1, 1.0, -%2 op_sel_hi:[0,0,1]" : "=v"(lp) : "v"(a), "v"(hp));
    asm volatile("v_fma_mixhi_f16 %0, %1, 1.0, -%2 op_sel:[0,0,1] op_sel_hi:[0,0,1]" : "+v"(lp) : "v"(b), "v"(hp));
    _Float16 ha = (_Float16)a, hb = (_Float16)b;
    _Float16 la = (_Float16)(a - (float)ha), lb = (_Float16)(b - (float)hb);
    h2 hr = {ha, hb}, lr = {la, lb};
    out[4 * i] = hp; out[4 * i + 1] = lp; out[4 * i + 2] = __builtin_bit_cast(unsigned, hr); out[4 * i + 3] = __builtin_bit_cast(unsigned, lr);
}
int main()
{
    const int n = 1 << 22;
    float *hx = new float[n]; unsigned *ho = new unsigned[2 * n];
    unsigned s = 12345;
    for (int i = 0; i < n; i++) {
        s = s * 1664525u + 1013904223u;
        unsigned u = s; float f;
        if (i < n / 4) { int e = 100 + (int)((u >> 23) % 40); u = (u & 0x807FFFFFu) | ((unsigned)e << 23); }   // 2^-27 .. 2^12
        else if (i < n / 2) { int e = 118 + (int)((u >> 23) % 16); u = (u & 0x807FFFFFu) | ((unsigned)e << 23); }
        else { int e = 60 + (int)((u >> 23) % 100); u = (u & 0x807FFFFFu) | ((unsigned)e << 23); }
        memcpy(&f, &u, 4); hx[i] = f;
    }
    hx[0] = 0.f; hx[1] = -0.f; hx[2] = 1.f; hx[3] = 65504.f;
    float *dx; unsigned *dout;
    hipMalloc(&dx, n * 4); hipMalloc(&dout, 2 * n * 4);
    hipMemcpy(dx, hx, n * 4, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3(n / 2 / 256), dim3(256), 0, 0, dx, dout, n);
    hipMemcpy(ho, dout, 2 * n * 4, hipMemcpyDeviceToHost);
    long badh = 0, badl = 0;
    for (int i = 0; i < n / 2; i++) {
        if (ho[4 * i] != ho[4 * i + 2]) { if (badh < 5) printf("hi mismatch x=(%g,%g) asm %08x c %08x\n", hx[2*i], hx[2*i+1], ho[4*i], ho[4*i+2]); badh++; }
        if (ho[4 * i + 1] != ho[4 * i + 3]) { if (badl < 8) printf("lo mismatch x=(%.9g,%.9g) asm %08x c %08x (hi %08x)\n", hx[2*i], hx[2*i+1], ho[4*i+1], ho[4*i+3], ho[4*i]); badl++; }
    }
    printf("pairs %d: hi mismatches %ld, lo mismatches %ld\n", n / 2, badh, badl);
    return 0;
}
