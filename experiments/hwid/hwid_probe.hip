// Where do the two waves of each 128-thread workgroup land (XCC, SE, CU, SIMD) when 1000 workgroups with ~37 KB of LDS each
// are resident together?  hipcc --offload-arch=gfx950 -O2 hwid_probe.hip -o hwid_probe && ./hwid_probe [n_blocks] [lds_bytes]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <map>
#include <vector>
__global__ void __launch_bounds__(128) probe(unsigned *out, long long spin_ticks) {
    extern __shared__ double lds[];
    const int wave = threadIdx.x >> 6;
    unsigned hw, xcc;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    if ((threadIdx.x & 63) == 0) {
        out[(blockIdx.x * 2 + wave) * 2] = hw;
        out[(blockIdx.x * 2 + wave) * 2 + 1] = xcc;
    }
    lds[threadIdx.x] = 1.0;
    const long long t0 = wall_clock64();
    while (wall_clock64() - t0 < spin_ticks) __builtin_amdgcn_s_sleep(8);
}
int main(int argc, char **argv) {
    const int n = argc > 1 ? atoi(argv[1]) : 1000;
    const int lds = argc > 2 ? atoi(argv[2]) : 37 * 1024;
    unsigned *d, *h = (unsigned *)malloc(n * 4 * sizeof(unsigned));
    hipMalloc(&d, n * 4 * sizeof(unsigned));
    hipFuncSetAttribute((const void *)probe, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    hipLaunchKernelGGL(probe, dim3(n), dim3(128), lds, 0, d, 20000LL);  // 100 MHz ticks: 200 us
    hipDeviceSynchronize();
    hipMemcpy(h, d, n * 4 * sizeof(unsigned), hipMemcpyDeviceToHost);
    std::map<unsigned, std::vector<int>> per_simd;  // key: xcc, se, cu, simd
    for (int b = 0; b < n; b++)
        for (int wv = 0; wv < 2; wv++) {
            const unsigned hw = h[(b * 2 + wv) * 2], xcc = h[(b * 2 + wv) * 2 + 1] & 0xf;
            const unsigned simd = (hw >> 4) & 3, cu = (hw >> 8) & 0xf, sh = (hw >> 12) & 1, se = (hw >> 13) & 7;
            if (b < 24 || (b % 256) < 2) printf("block %4d wave %d: xcc %u se %u sh %u cu %2u simd %u wave_id %u\n", b, wv, xcc, se, sh, cu, simd, hw & 0xf);
            per_simd[(xcc << 16) | (se << 12) | (sh << 11) | (cu << 4) | simd].push_back(b * 2 + wv);
        }
    // what shares a SIMD: pairs of (block, wave)
    int same_wave_index = 0, mixed = 0, other = 0;
    std::map<int, int> hist;
    for (auto &kv : per_simd) {
        hist[(int)kv.second.size()]++;
        if (kv.second.size() == 2) {
            if ((kv.second[0] & 1) == (kv.second[1] & 1)) same_wave_index++; else mixed++;
        } else other++;
    }
    printf("SIMDs in use %zu; waves per SIMD histogram:", per_simd.size());
    for (auto &kv : hist) printf("  %d waves: %d SIMDs", kv.first, kv.second);
    printf("\nSIMDs with two waves: both the same wave index (0,0 or 1,1) %d, mixed (0,1) %d\n", same_wave_index, mixed);
    int shown = 0;
    for (auto &kv : per_simd) {
        if (shown++ >= 16) break;
        printf("xcc %u se %u cu %2u simd %u:", kv.first >> 16, (kv.first >> 12) & 7, (kv.first >> 4) & 0xf, kv.first & 3);
        for (int x : kv.second) printf("  block %d wave %d", x >> 1, x & 1);
        printf("\n");
    }
    return 0;
}
