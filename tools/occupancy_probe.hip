// How many 128-thread workgroups with a given LDS block are REALLY resident per CU on this chip?  Every workgroup stamps its
// start (100 MHz constant clock) and spins 50 us; workgroups that start within 20 us of the first one are the first resident
// set.  Diagnostics (round 6: is the 10th workgroup of the 16 384-byte HOME block resident?).
//   hipcc --offload-arch=gfx950 -O2 tools/occupancy_probe.hip -o /tmp/occ_probe && /tmp/occ_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <algorithm>
template <int THREADS, int WPE>
__global__ __launch_bounds__(THREADS) __attribute__((amdgpu_waves_per_eu(WPE, WPE))) void k_spin(unsigned long long *start, int spin_ticks)
{
    extern __shared__ char lds[];
    const unsigned long long t0 = wall_clock64();
    if (threadIdx.x == 0) { start[blockIdx.x] = t0; lds[0] = 1; }
    while (wall_clock64() - t0 < (unsigned long long)spin_ticks) __builtin_amdgcn_s_sleep(8);
}
template <int THREADS, int WPE> static void sweep(int cus, std::initializer_list<int> sizes)
{
    for (int lds : sizes) {
        int nb = 0;
        hipFuncSetAttribute((const void *)k_spin<THREADS, WPE>, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
        hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, k_spin<THREADS, WPE>, THREADS, lds);
        const int grid = cus * 24;
        unsigned long long *d; hipMalloc(&d, grid * 8); hipMemset(d, 0, grid * 8);
        hipLaunchKernelGGL((k_spin<THREADS, WPE>), dim3(grid), dim3(THREADS), lds, 0, d, 5000);
        hipDeviceSynchronize();
        std::vector<unsigned long long> h(grid);
        hipMemcpy(h.data(), d, grid * 8, hipMemcpyDeviceToHost);
        const unsigned long long t0 = *std::min_element(h.begin(), h.end());
        int first = 0; for (auto t : h) first += (t - t0 < 2000);
        printf("%3d threads, %d waves/SIMD, LDS %6d B: occupancy query %2d per CU; started within 20 us: %5d = %.2f per CU\n", THREADS, WPE, lds, nb,
               first, first / (double)cus);
        hipFree(d);
    }
}
int main()
{
    hipDeviceProp_t prop; hipGetDeviceProperties(&prop, 0);
    const int cus = prop.multiProcessorCount;
    printf("%s: %d CUs, LDS per CU %zu B\n", prop.name, cus, (size_t)prop.maxSharedMemoryPerMultiProcessor);
    sweep<128, 5>(cus, {16896, 16384, 16128, 15872, 15744, 15616, 15488, 15360, 14336});
    sweep<128, 4>(cus, {20480, 19968, 19456, 19200, 18944, 18432});
    sweep<64, 4>(cus, {10240, 10216, 9984, 9856, 9728, 9600, 9472, 9216});
    return 0;
}
