// Micro-benchmark: cost of a software grid barrier inside a persistent (cooperative) kernel on MI355X, with a
// cross-workgroup visibility check.  Decides whether fusing the decode layer's kernels behind grid barriers can pay.
// Build: hipcc --offload-arch=gfx950 -O3 -o gridsync_bench gridsync_bench.hip
//   V0: release fetch_add + acquire-load spin          V1: fences + relaxed add / relaxed spin (sleep 1)
//   V2: as V1, sleep 20 between polls                  V3: as V2, per-XCD arrival counters, one global add per XCD
#include <hip/hip_runtime.h>
#include <cstdio>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)

// counter[0] = arrivals, counter[1] = abort flag (a barrier that spins too long sets it and every later barrier falls
// through, so the grid always drains), counter[16 + 16*x] = arrivals of XCD x
template <int V>
__device__ __forceinline__ void grid_barrier(unsigned* counter, unsigned target) {
    __syncthreads();
    if (threadIdx.x == 0) {
        if (V == 0) {
            __hip_atomic_fetch_add(counter, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
        } else {
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
            if (V == 3) {
                const unsigned xcd = blockIdx.x & 7, nx = (gridDim.x - xcd + 7) / 8;
                const unsigned old = __hip_atomic_fetch_add(counter + 16 + 16 * xcd, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                if ((old + 1) % nx == 0) __hip_atomic_fetch_add(counter, nx, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            } else {
                __hip_atomic_fetch_add(counter, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
        }
        unsigned spins = 0;
        for (;;) {
            const unsigned seen = V == 0 ? __hip_atomic_load(counter, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT)
                                         : __hip_atomic_load(counter, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (seen >= target) break;
            if (V >= 2) __builtin_amdgcn_s_sleep(20);
            else __builtin_amdgcn_s_sleep(1);
            if (++spins > 2000000u || __hip_atomic_load(counter + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) {
                __hip_atomic_store(counter + 1, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                break;
            }
        }
        if (V != 0) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    }
    __syncthreads();
}

template <int V>
__global__ __launch_bounds__(256) void persist(unsigned* counter, int* slots, int* errors, int iters) {
    const int G = gridDim.x, wg = blockIdx.x;
    int bad = 0;
    for (int it = 0; it < iters; ++it) {
        slots[wg * 256 + threadIdx.x] = it * 7 + wg;                 // "phase output"
        grid_barrier<V>(counter, (unsigned)(2 * it + 1) * G);
        const int src = (wg + 37 + it) % G;
        if (slots[src * 256 + threadIdx.x] != it * 7 + src) ++bad;   // "next phase input", written by another workgroup
        grid_barrier<V>(counter, (unsigned)(2 * it + 2) * G);       // before the slot is overwritten
    }
    if (bad) atomicAdd(errors, bad);
}

int main() {
    int cus = 0;
    CK(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, 0));
    printf("CUs %d\n", cus);
    unsigned* counter;
    int *slots, *errors;
    CK(hipMalloc(&counter, 1024));
    CK(hipMalloc(&slots, 4096 * 256 * 4));
    CK(hipMalloc(&errors, 4));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    const void* fns[4] = {(const void*)persist<0>, (const void*)persist<1>, (const void*)persist<2>, (const void*)persist<3>};
    for (int V = 0; V < 4; ++V)
        for (int G : {256, 512, 1024}) {
            int occ = 0;
            CK(hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, persist<0>, 256, 0));
            if (G > occ * cus) { printf("G %d exceeds co-residency %d\n", G, occ * cus); continue; }
            for (int rep = 0; rep < 2; ++rep) {
                int iters = 300;
                CK(hipMemset(counter, 0, 1024));
                CK(hipMemset(errors, 0, 4));
                CK(hipEventRecord(e0));
                void* args[] = {&counter, &slots, &errors, &iters};
                hipError_t err = hipLaunchCooperativeKernel(fns[V], dim3(G), dim3(256), args, 0, 0);
                CK(hipEventRecord(e1));
                CK(hipEventSynchronize(e1));
                float ms = 0;
                CK(hipEventElapsedTime(&ms, e0, e1));
                int herr = 0;
                unsigned hc[2] = {0, 0};
                CK(hipMemcpy(&herr, errors, 4, hipMemcpyDeviceToHost));
                CK(hipMemcpy(hc, counter, 8, hipMemcpyDeviceToHost));
                printf("V%d G %4d: launch %s, %d iters x 2 barriers: %.3f ms -> %.3f us per barrier, visibility errors %d%s\n", V, G,
                       hipGetErrorString(err), iters, ms, ms * 1e3 / (2 * iters), herr, hc[1] ? "  ABORTED (barrier timed out)" : "");
            }
        }
    return 0;
}
