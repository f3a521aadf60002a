// One-way latency of a relaxed agent-scope (sc1) store -> agent-scope load hand-off between two workgroups, on the same XCD
// (blockIdx a, a + 8) and across XCDs (a, a + 1); MI355X dispatches workgroup b to XCD b % 8.
// hipcc --offload-arch=gfx950 -O2 pingpong.hip -o pingpong && ./pingpong
#include <hip/hip_runtime.h>
#include <cstdio>
template <int SCOPE>
__global__ void __launch_bounds__(64) pingpong(unsigned long long *flags, int stride, int rounds, long long *ticks, unsigned *xcc) {
    // pairs: block b < stride*? ... block b talks to partner b ^ stride (stride a power of two: 1 = other XCD, 8 = same XCD)
    const int b = blockIdx.x, partner = b ^ stride;
    const bool first = (b & stride) == 0;
    unsigned x;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(x));
    if (threadIdx.x == 0) xcc[b] = x & 0xf;
    if (threadIdx.x != 0) return;
    unsigned long long *mine = flags + 32 * b, *theirs = flags + 32 * partner;  // 256 B apart
    const long long t0 = wall_clock64();
    for (int i = 1; i <= rounds; i++) {
        if (first) {
            __hip_atomic_store(mine, (unsigned long long)i, __ATOMIC_RELAXED, SCOPE);
            while (__hip_atomic_load(theirs, __ATOMIC_RELAXED, SCOPE) < (unsigned long long)i) {}
        } else {
            while (__hip_atomic_load(theirs, __ATOMIC_RELAXED, SCOPE) < (unsigned long long)i) {}
            __hip_atomic_store(mine, (unsigned long long)i, __ATOMIC_RELAXED, SCOPE);
        }
    }
    ticks[b] = wall_clock64() - t0;
}
int main() {
    const int nb = 16, rounds = 2000;
    unsigned long long *flags; long long *ticks, h[nb]; unsigned *xcc, hx[nb];
    hipMalloc(&flags, nb * 32 * 8); hipMalloc(&ticks, nb * 8); hipMalloc(&xcc, nb * 4);
    for (int stride : {1, 8, 2, 4}) {
        for (int rep = 0; rep < 2; rep++) {
            hipMemset(flags, 0, nb * 32 * 8);
            hipLaunchKernelGGL(pingpong<__HIP_MEMORY_SCOPE_AGENT>, dim3(nb), dim3(64), 0, 0, flags, stride, rounds, ticks, xcc);
            hipDeviceSynchronize();
        }
        hipMemcpy(h, ticks, sizeof(h), hipMemcpyDeviceToHost);
        hipMemcpy(hx, xcc, sizeof(hx), hipMemcpyDeviceToHost);
        printf("partner = block ^ %d: block 0 on XCC %u, partner on XCC %u: %.0f ns per one-way hand-off (100 MHz clock: %lld ticks / %d round trips)\n",
               stride, hx[0], hx[stride], h[0] * 10.0 / (2.0 * rounds), h[0], rounds);
    }
    // (A workgroup-scope variant of the same-XCD pair — sc0 only, hoping the XCD's shared L2 would answer — never terminates: the
    // polling load keeps hitting the CU's own L1.  Agent scope is the weakest that works, and it costs the same 0.5 - 0.6 us
    // whether or not the two workgroups share an XCD.)
    return 0;
}
