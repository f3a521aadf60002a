// How fast does ONE wave get 16-byte write-through (sc1) stores out?  (The publication of a resident launch: 240 chunks of 16 bytes per
// robot and segment, four store instructions per lane — in-kernel stamps showed the issuing wave held for ~2 k clocks.)
// Each workgroup (64 threads) stores `n_instr` x 1 KB contiguous per round, `rounds` times, and times the issue with s_memtime; then the
// same with a drain (s_waitcnt vmcnt(0)) per round.  Forms: plain, sc1, sc0 sc1; one wave per CU (256 workgroups) or four (1024).
// hipcc --offload-arch=gfx950 -O2 store_rate.hip -o store_rate && ./store_rate
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
template <int AUX, bool DRAIN>
__global__ void __launch_bounds__(64) k(unsigned char *area, int n_instr, int rounds, unsigned long long *out) {
    const int lane = threadIdx.x;
    unsigned char *mine = area + (size_t)blockIdx.x * 16384;
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(mine, 0, 16384, 0x00020000);
    unsigned long long t_issue = 0, t_all = 0;
    for (int i = 1; i <= rounds; i++) {
        const unsigned long long t0 = __builtin_readcyclecounter();
        for (int c = 0; c < n_instr; c++) {
            u32x4 v = {(unsigned)i, (unsigned)c, (unsigned)lane, (unsigned)i};
            __builtin_amdgcn_raw_buffer_store_b128(v, rs, 1024 * c + 16 * lane, 0, AUX);
        }
        const unsigned long long t1 = __builtin_readcyclecounter();
        if (DRAIN) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        const unsigned long long t2 = __builtin_readcyclecounter();
        t_issue += t1 - t0;
        t_all += t2 - t0;
        __builtin_amdgcn_s_sleep(20);  // (1280 clocks between rounds, like the rest of an iteration)
    }
    if (lane == 0) { out[2 * blockIdx.x] = t_issue; out[2 * blockIdx.x + 1] = t_all; }
}
int main() {
    const int rounds = 500;
    unsigned char *area; unsigned long long *out;
    hipMalloc(&area, (size_t)1024 * 16384); hipMalloc(&out, 1024 * 16);
    for (int nb : {256, 1024}) {
        for (int n_instr : {1, 4, 8}) {
            for (int form = 0; form < 6; form++) {
                hipMemset(out, 0, 1024 * 16);
                const bool drain = form >= 3;
                switch (form) {
                case 0: hipLaunchKernelGGL((k<0, false>), dim3(nb), dim3(64), 0, 0, area, n_instr, rounds, out); break;
                case 1: hipLaunchKernelGGL((k<16, false>), dim3(nb), dim3(64), 0, 0, area, n_instr, rounds, out); break;
                case 2: hipLaunchKernelGGL((k<17, false>), dim3(nb), dim3(64), 0, 0, area, n_instr, rounds, out); break;
                case 3: hipLaunchKernelGGL((k<0, true>), dim3(nb), dim3(64), 0, 0, area, n_instr, rounds, out); break;
                case 4: hipLaunchKernelGGL((k<16, true>), dim3(nb), dim3(64), 0, 0, area, n_instr, rounds, out); break;
                default: hipLaunchKernelGGL((k<17, true>), dim3(nb), dim3(64), 0, 0, area, n_instr, rounds, out); break;
                }
                hipDeviceSynchronize();
                std::vector<unsigned long long> h(2 * nb);
                hipMemcpy(h.data(), out, sizeof(unsigned long long) * 2 * nb, hipMemcpyDeviceToHost);
                double a = 0, b = 0;
                for (int i = 0; i < nb; i++) { a += h[2 * i]; b += h[2 * i + 1]; }
                printf("%4d workgroups, %d x 1 KB per round, %-8s %-6s: issue %7.0f clocks per round, %s %7.0f\n", nb, n_instr,
                       (form % 3) == 0 ? "plain" : (form % 3) == 1 ? "sc1" : "sc0 sc1", drain ? "drain" : "", a / nb / rounds, drain ? "issue + drain" : "(same)       ", b / nb / rounds);
            }
        }
    }
    return 0;
}
