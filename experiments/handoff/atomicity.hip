// Is a 16-byte aligned write-through store ONE transaction for a polling reader on another CU?  (records.hip, form B, assumed so —
// with a payload that did not change from round to round, i.e. it could not have told.)  Pairs (b, b ^ 1) ping-pong; every chunk's
// payload is a function of the round, the last dword is the round itself; the receiver polls until the last dword is the round's and
// then checks the other three.  Forms: buffer b128 sc1 (the product's accesses) and global dwordx4 sc1; and, for comparison, 8-byte
// halves that each carry a 16-bit tag (the form the product uses if 16 bytes tear).
// hipcc --offload-arch=gfx950 -O2 atomicity.hip -o atomicity && ./atomicity
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ __amdgpu_buffer_rsrc_t rsrc(const void *base, unsigned bytes) {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void *>(base), 0, (int)bytes, 0x00020000);
}
__device__ __forceinline__ unsigned f1(unsigned i, unsigned c) { return i * 2654435761u + c; }
__device__ __forceinline__ unsigned f2(unsigned i, unsigned c) { return (i ^ 0x5a5a5a5au) + 7u * c; }
__device__ __forceinline__ unsigned f3(unsigned i, unsigned c) { return ~i - c; }
// FORM 0: buffer b128, tag in dword 3; 1: global dwordx4; 2: buffer b128, 16-bit tags in dwords 1 and 2 (two self-validating halves)
template <int FORM>
__global__ void __launch_bounds__(64) k(unsigned char *area, int chunks, int rounds, unsigned *torn, unsigned *timeouts) {
    const int b = blockIdx.x, partner = b ^ 1, lane = threadIdx.x;
    const bool first = (b & 1) == 0;
    unsigned char *mine = area + (size_t)b * 4096, *theirs = area + (size_t)partner * 4096;
    const __amdgpu_buffer_rsrc_t rm = rsrc(mine, 4096), rt = rsrc(theirs, 4096);
    unsigned bad = 0, to = 0;
    for (int i = 1; i <= rounds; i++) {
        for (int half = 0; half < 2; half++) {
            const bool send = (half == 0) == first;
            for (int c = lane; c < chunks; c += 64) {
                const unsigned ui = (unsigned)i, uc = (unsigned)c;
                if (send) {
                    u32x4 v;
                    if (FORM == 2) {
                        const unsigned mid = f2(ui, uc), tag = ui & 0xffffu;
                        v = u32x4{f1(ui, uc), (tag << 16) | (mid & 0xffffu), (tag << 16) | (mid >> 16), f3(ui, uc)};
                    } else {
                        v = u32x4{f1(ui, uc), f2(ui, uc), f3(ui, uc), ui};
                    }
                    if (FORM == 1) asm volatile("global_store_dwordx4 %0, %1, off sc1\n\ts_nop 1" ::"v"(mine + 16 * c), "v"(v) : "memory");
                    else __builtin_amdgcn_raw_buffer_store_b128(v, rm, 16 * c, 0, 16);
                } else {
                    u32x4 v;
                    for (unsigned spins = 0;; spins++) {
                        if (FORM == 1) asm volatile("global_load_dwordx4 %0, %1, off sc1\n\ts_waitcnt vmcnt(0)" : "=v"(v) : "v"(theirs + 16 * c) : "memory");
                        else v = __builtin_amdgcn_raw_buffer_load_b128(rt, 16 * c, 0, 16);
                        const bool ok = FORM == 2 ? ((v.y >> 16) == (ui & 0xffffu) && (v.z >> 16) == (ui & 0xffffu)) : v.w == ui;
                        if (ok) break;
                        if (spins > (1u << 20)) { to++; break; }
                        __builtin_amdgcn_s_sleep(1);
                    }
                    if (FORM == 2) {
                        const unsigned mid = (v.y & 0xffffu) | (v.z << 16);
                        bad += (v.x != f1(ui, uc) || mid != f2(ui, uc) || v.w != f3(ui, uc)) ? 1u : 0u;
                    } else {
                        bad += (v.x != f1(ui, uc) || v.y != f2(ui, uc) || v.z != f3(ui, uc)) ? 1u : 0u;
                    }
                }
            }
        }
    }
    if (bad) atomicAdd(torn, bad);
    if (to) atomicAdd(timeouts, to);
}
int main() {
    const int rounds = 2000;
    for (int nb : {2, 1000}) {
        unsigned char *area; unsigned *cnt;
        hipMalloc(&area, (size_t)nb * 4096); hipMalloc(&cnt, 8);
        for (int chunks : {64, 240}) {
            for (int form = 0; form < 3; form++) {
                hipMemset(area, 0, (size_t)nb * 4096); hipMemset(cnt, 0, 8);
                if (form == 0) hipLaunchKernelGGL(k<0>, dim3(nb), dim3(64), 0, 0, area, chunks, rounds, cnt, cnt + 1);
                else if (form == 1) hipLaunchKernelGGL(k<1>, dim3(nb), dim3(64), 0, 0, area, chunks, rounds, cnt, cnt + 1);
                else hipLaunchKernelGGL(k<2>, dim3(nb), dim3(64), 0, 0, area, chunks, rounds, cnt, cnt + 1);
                hipDeviceSynchronize();
                unsigned h[2]; hipMemcpy(h, cnt, 8, hipMemcpyDeviceToHost);
                printf("%4d workgroups, %3d chunks, %s: %u of %lld chunk hand-offs with a payload that is not the validated round's, %u timeouts\n", nb, chunks,
                       form == 0 ? "buffer b128 sc1, 32-bit tag in dword 3 " : form == 1 ? "global dwordx4 sc1, 32-bit tag in dword 3" : "buffer b128 sc1, 16-bit tag per 8-byte half",
                       h[0], (long long)nb * chunks * rounds, h[1]);
            }
        }
        hipFree(area); hipFree(cnt);
    }
    return 0;
}
