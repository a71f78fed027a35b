// On which SIMDs do the two waves of a 128-thread workgroup land?  The resident kernels give wave 0 the dense mapping of the
// controlled vehicles (~75 % of a workgroup's vector instructions); if the dispatcher always puts wave 0 on the same SIMD pair
// the CU's four vector pipes are loaded 3 : 1.  Every wave stamps HW_REG_HW_ID (wave slot, SIMD, CU, SE) and the XCC id and
// spins 50 us (so that the first resident set is all there at once).  Diagnostics, round 6.
//   hipcc --offload-arch=gfx950 -O2 tools/simd_probe.hip -o build/simd_probe && build/simd_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <map>
template <int THREADS, int WPE>
__global__ __launch_bounds__(THREADS) __attribute__((amdgpu_waves_per_eu(WPE, WPE))) void k_where(unsigned *out, int spin_ticks)
{
    extern __shared__ char lds[];
    const unsigned long long t0 = wall_clock64();
    const unsigned hw = __builtin_amdgcn_s_getreg((31 << 11) | (0 << 6) | 4);       // HW_REG_HW_ID[31:0]
    const unsigned xcc = __builtin_amdgcn_s_getreg((3 << 11) | (0 << 6) | 20) & 0xFu;
    if ((threadIdx.x & 63) == 0) {
        out[(blockIdx.x * (THREADS / 64) + threadIdx.x / 64) * 2] = hw;
        out[(blockIdx.x * (THREADS / 64) + threadIdx.x / 64) * 2 + 1] = xcc;
        lds[threadIdx.x] = 1;
    }
    while (wall_clock64() - t0 < (unsigned long long)spin_ticks) __builtin_amdgcn_s_sleep(8);
}
template <int THREADS, int WPE> static void probe(int cus, int lds, int per_cu)
{
    constexpr int NW = THREADS / 64;
    const int grid = cus * per_cu;
    hipFuncSetAttribute((const void *)k_where<THREADS, WPE>, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    unsigned *d; hipMalloc(&d, grid * NW * 8);
    hipLaunchKernelGGL((k_where<THREADS, WPE>), dim3(grid), dim3(THREADS), lds, 0, d, 5000);
    hipDeviceSynchronize();
    std::vector<unsigned> h(grid * NW * 2);
    hipMemcpy(h.data(), d, grid * NW * 8, hipMemcpyDeviceToHost);
    long long hist[NW][4] = {};
    std::map<unsigned, std::vector<int>> per_cu_w0;           // (xcc, se, sh, cu) -> wave-0 count per SIMD
    long long pair[4][4] = {};
    for (int b = 0; b < grid; b++) {
        int simd[NW];
        for (int w = 0; w < NW; w++) {
            const unsigned hw = h[(b * NW + w) * 2], xcc = h[(b * NW + w) * 2 + 1];
            simd[w] = (hw >> 4) & 3;
            hist[w][simd[w]]++;
            if (w == 0) {
                const unsigned key = (xcc << 16) | (hw & 0xFF00u);       // cu_id 11:8, sh_id 12, se_id 15:13
                auto &v = per_cu_w0[key]; if (v.empty()) v.assign(4, 0);
                v[simd[0]]++;
            }
        }
        if (NW == 2) pair[simd[0]][simd[1]]++;
    }
    printf("%d threads, %d waves/SIMD, LDS %d B, %d workgroups (%d per CU), CUs seen %zu\n", THREADS, WPE, lds, grid, per_cu, per_cu_w0.size());
    for (int w = 0; w < NW; w++) printf("  wave %d on SIMD 0..3: %6lld %6lld %6lld %6lld\n", w, hist[w][0], hist[w][1], hist[w][2], hist[w][3]);
    if (NW == 2) {
        printf("  (SIMD of wave 0, SIMD of wave 1) counts:");
        for (int a = 0; a < 4; a++) for (int c = 0; c < 4; c++) if (pair[a][c]) printf(" (%d,%d) %lld", a, c, pair[a][c]);
        printf("\n");
    }
    int worst = 0; std::map<int, int> spread;
    for (auto &kv : per_cu_w0) {
        int mx = 0, mn = 1 << 30; for (int s = 0; s < 4; s++) { mx = std::max(mx, kv.second[s]); mn = std::min(mn, kv.second[s]); }
        spread[mx]++; worst = std::max(worst, mx);
    }
    printf("  wave-0 count of the most loaded SIMD of a CU -> number of CUs:");
    for (auto &kv : spread) printf(" %d: %d", kv.first, kv.second);
    printf("\n  first workgroups: ");
    for (int b = 0; b < 12; b++) {
        printf("[b%d x%u cu%u se%u", b, h[b * NW * 2 + 1], (h[b * NW * 2] >> 8) & 15, (h[b * NW * 2] >> 13) & 7);
        for (int w = 0; w < NW; w++) printf(" w%d:simd%u/slot%u", w, (h[(b * NW + w) * 2] >> 4) & 3, h[(b * NW + w) * 2] & 15);
        printf("] ");
    }
    printf("\n");
    hipFree(d);
}
int main()
{
    hipDeviceProp_t prop; hipGetDeviceProperties(&prop, 0);
    const int cus = prop.multiProcessorCount;
    printf("%s: %d CUs\n", prop.name, cus);
    probe<128, 5>(cus, 15264, 10);      // the HOME kernel's shape: 10 workgroups per CU
    probe<128, 4>(cus, 16896, 8);       // the 8-workgroup register build
    probe<128, 4>(cus, 19728, 8);       // closed loop
    probe<64, 4>(cus, 10240, 16);       // capacity 64
    return 0;
}
