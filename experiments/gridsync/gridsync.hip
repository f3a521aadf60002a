// Microbenchmark: what does a grid-wide barrier cost on MI355X for 1000 workgroups x 128 threads?
// (a) cooperative groups grid.sync(); (b) hand-rolled barrier on a device-scope atomic counter with
// agent-scope fences; (c) the same with relaxed atomics only (no cache write-back / invalidate).
#include <hip/hip_cooperative_groups.h>
#include <hip/hip_runtime.h>

#include <cstdio>
#include <vector>
namespace cg = cooperative_groups;

__global__ void k_cg(int n, unsigned long long *out) {
    cg::grid_group g = cg::this_grid();
    unsigned long long acc = 0;
    for (int i = 0; i < n; i++) {
        g.sync();
        acc += i;
    }
    if (threadIdx.x == 0 && blockIdx.x == 0) out[0] = acc;
}
template <bool FENCE>
__global__ void k_own(int n, unsigned int *ctr, unsigned long long *out) {
    unsigned int target = 0;
    for (int i = 0; i < n; i++) {
        __syncthreads();
        target += gridDim.x;
        if (threadIdx.x == 0) {
            if (FENCE) __threadfence();
            __hip_atomic_fetch_add(ctr, 1u, FENCE ? __ATOMIC_RELEASE : __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            while (__hip_atomic_load(ctr, FENCE ? __ATOMIC_ACQUIRE : __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target) __builtin_amdgcn_s_sleep(2);
            if (FENCE) __threadfence();
        }
        __syncthreads();
    }
    if (threadIdx.x == 0 && blockIdx.x == 0) out[0] = target;
}
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
int main() {
    const int grid = 1000, block = 128, n = 200;
    unsigned int *ctr; unsigned long long *out;
    CK(hipMalloc(&ctr, 4)); CK(hipMalloc(&out, 8));
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    float ms;
    for (int lds : {0, 28 * 1024}) {
        int nn = n;
        void *args[] = {&nn, &out};
        for (int rep = 0; rep < 2; rep++) {
            CK(hipEventRecord(a));
            CK(hipLaunchCooperativeKernel((void *)k_cg, dim3(grid), dim3(block), args, lds, 0));
            CK(hipEventRecord(b)); CK(hipEventSynchronize(b)); CK(hipEventElapsedTime(&ms, a, b));
        }
        printf("cooperative groups grid.sync, %d B LDS/WG: %.2f us per barrier\n", lds, ms * 1e3 / n);
        void *args2[] = {&nn, &ctr, &out};
        for (int fence = 1; fence >= 0; fence--) {
            for (int rep = 0; rep < 2; rep++) {
                CK(hipMemset(ctr, 0, 4));
                CK(hipEventRecord(a));
                if (fence) CK(hipLaunchCooperativeKernel((void *)k_own<true>, dim3(grid), dim3(block), args2, lds, 0));
                else CK(hipLaunchCooperativeKernel((void *)k_own<false>, dim3(grid), dim3(block), args2, lds, 0));
                CK(hipEventRecord(b)); CK(hipEventSynchronize(b)); CK(hipEventElapsedTime(&ms, a, b));
            }
            printf("own barrier (%s), %d B LDS/WG: %.2f us per barrier\n", fence ? "release/acquire fences" : "relaxed", lds, ms * 1e3 / n);
        }
    }
    return 0;
}
