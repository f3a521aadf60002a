// What a hand-off of RECORDS between two workgroups costs, two protocols, on a loaded device (the resident schedule launch hands
// 15 response means of 32 bytes per neighbour over in every iteration):
//   A  "records, then a progress word" (the product's): the sender stores its chunks (sc1), waits for the stores to be acknowledged,
//      stores the word; the receiver polls the word, then loads the chunks — two dependent trips after the stores are through.
//   B  "self-validating chunks": every 16-byte chunk carries its sequence number next to 8 bytes of payload (a 16-byte aligned store
//      is one transaction), the receiver polls the chunks themselves — one trip, twice the bytes.
// Pairs (b, b ^ 1) ping-pong `rounds` times, all pairs at once (1000 workgroups of 64 lanes: every CU busy polling, like the launch).
// hipcc --offload-arch=gfx950 -O2 records.hip -o records && ./records
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
// (s_nop: a store's data registers are read a wait state after issue, and the compiler does not look into the asm)
__device__ __forceinline__ void st16(void *p, u32x4 v) { asm volatile("global_store_dwordx4 %0, %1, off sc1\n\ts_nop 1" ::"v"(p), "v"(v) : "memory"); }
__device__ __forceinline__ u32x4 ld16(const void *p) {
    u32x4 v;
    asm volatile("global_load_dwordx4 %0, %1, off sc1\n\ts_waitcnt vmcnt(0)" : "=v"(v) : "v"(p) : "memory");
    return v;
}
// P chunks of payload per hand-off (one per lane); A: payload bytes 16 per chunk; B: 8 per chunk (so B moves 2 P chunks for the same payload)
template <int PROTO>
__global__ void __launch_bounds__(64) k(unsigned char *area, unsigned long long *flags, int P, int rounds, long long *ticks, unsigned *bad) {
    const int b = blockIdx.x, partner = b ^ 1, lane = threadIdx.x;
    const bool first = (b & 1) == 0;
    const int chunks = PROTO == 0 ? P : 2 * P;
    unsigned char *mine = area + (size_t)b * 2048, *theirs = area + (size_t)partner * 2048;  // 128 chunks of room each
    unsigned long long *fm = flags + 32 * b, *ft = flags + 32 * partner;
    unsigned wrong = 0;
    const long long t0 = wall_clock64();
    for (int i = 1; i <= rounds; i++) {
        for (int half = 0; half < 2; half++) {
            const bool send = (half == 0) == first;
            if (send) {
                for (int c = lane; c < chunks; c += 64) {
                    u32x4 v;
                    if (PROTO == 0) v = u32x4{(unsigned)i, (unsigned)c, (unsigned)b, 7u};
                    else v = u32x4{(unsigned)c, (unsigned)b, (unsigned)i, 0u};  // payload (8 bytes), sequence (8 bytes)
                    st16(mine + 16 * c, v);
                }
                if (PROTO == 0) {
                    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                    __builtin_amdgcn_s_barrier();  // (one wave: all lanes' stores are through)
                    if (lane == 0) __hip_atomic_store(fm, (unsigned long long)i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                }
            } else {
                if (PROTO == 0) {
                    for (unsigned spins = 0; __hip_atomic_load(ft, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < (unsigned long long)i; spins++) {
                        if (spins > (1u << 22)) { wrong += 1000000u; break; }  // (never a hung device)
                        __builtin_amdgcn_s_sleep(1);
                    }
                    for (int c = lane; c < chunks; c += 64) {
                        const u32x4 v = ld16(theirs + 16 * c);
                        wrong += (v.x != (unsigned)i || v.y != (unsigned)c) ? 1u : 0u;
                    }
                } else {
                    for (int c = lane; c < chunks; c += 64) {
                        u32x4 v = ld16(theirs + 16 * c);
                        for (unsigned spins = 0; v.z != (unsigned)i; spins++) {
                            if (spins > (1u << 22)) { wrong += 1000000u; break; }
                            __builtin_amdgcn_s_sleep(1);
                            v = ld16(theirs + 16 * c);
                        }
                        if (v.x != (unsigned)c && bad[1] == 0) { bad[1] = 1; bad[2] = c; bad[3] = v.x; bad[4] = v.y; bad[5] = v.z; bad[6] = v.w; bad[7] = i; }
                        wrong += (v.x != (unsigned)c) ? 1u : 0u;
                    }
                }
            }
        }
    }
    const long long dt = wall_clock64() - t0;
    if (lane == 0) ticks[b] = dt;
    if (wrong) atomicAdd(bad, wrong);
}
int main() {
    const int rounds = 2000;
    for (int nb : {2, 1000}) {
        unsigned char *area; unsigned long long *flags; long long *ticks; unsigned *bad;
        hipMalloc(&area, (size_t)nb * 2048); hipMalloc(&flags, (size_t)nb * 32 * 8); hipMalloc(&ticks, nb * 8); hipMalloc(&bad, 64);
        std::vector<long long> h(nb);
        for (int P : {30, 60}) {
            for (int proto = 0; proto < 2; proto++) {
                double best = 1e30; unsigned hb = 0;
                for (int rep = 0; rep < 3; rep++) {
                    hipMemset(area, 0, (size_t)nb * 2048); hipMemset(flags, 0, (size_t)nb * 32 * 8); hipMemset(bad, 0, 64);
                    if (proto == 0) hipLaunchKernelGGL(k<0>, dim3(nb), dim3(64), 0, 0, area, flags, P, rounds, ticks, bad);
                    else hipLaunchKernelGGL(k<1>, dim3(nb), dim3(64), 0, 0, area, flags, P, rounds, ticks, bad);
                    hipDeviceSynchronize();
                    hipMemcpy(h.data(), ticks, nb * 8, hipMemcpyDeviceToHost);
                    hipMemcpy(&hb, bad, 4, hipMemcpyDeviceToHost);
                    { unsigned dbg[8]; hipMemcpy(dbg, bad, 32, hipMemcpyDeviceToHost); if (dbg[1] && rep == 0) printf("   first wrong: chunk %u holds (%u, %u, %u, %u) in round %u\n", dbg[2], dbg[3], dbg[4], dbg[5], dbg[6], dbg[7]); }
                    double mean = 0; for (long long t : h) mean += (double)t; mean /= nb;
                    best = mean < best ? mean : best;
                }
                printf("%4d workgroups, %2d x 16 bytes of payload, protocol %s: %.0f ns per one-way hand-off (%u wrong)\n", nb, P,
                       proto == 0 ? "A (records + progress word)" : "B (self-validating chunks) ", best * 10.0 / (2.0 * rounds), hb);
            }
        }
        hipFree(area); hipFree(flags); hipFree(ticks); hipFree(bad);
    }
    return 0;
}
