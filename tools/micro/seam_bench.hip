// Micro-benchmark (round 4): what ONE all-to-all edge of a persistent batch-1 decode layer costs on MI355X, in the two forms the CDNA
// guide prices (MI355X_MICROARCH.md, price list: barrier-xcd 4.1-4.8 us, allgather 2.4-4.2 us), measured here in this project's own
// harness before any persistent layer is built on them (round 1 priced only a flat one-counter barrier: 9-14 us).
//   B: XCD-hierarchical grid barrier -- per-XCD arrival counter, the last arriver of an XCD adds to the top counter, the last XCD
//      publishes the generation word of every XCD; workgroups poll their own XCD's generation (relaxed, s_sleep), one acquire fence.
//   G: data-tagged all-gather -- every workgroup publishes its slice of an N-element vector as 8-byte {epoch, value} granules (ONE sc1
//      store each: the data is the flag); ONE wave per workgroup sweeps all N granules until every tag equals the epoch and writes the
//      values to LDS; no flag, no fence.
// Both loops are bounded (give-up word) so the grid always drains.  Workgroups = CUs (one per CU), 256 threads.
// Build: hipcc --offload-arch=gfx950 -O3 -o seam_bench seam_bench.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)

typedef __attribute__((address_space(1))) unsigned gu32;
typedef __attribute__((address_space(1))) unsigned long long gu64;
#define RLX __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT

struct Sync {            // zeroed before every launch
    unsigned xcnt[8][32];    // per-XCD arrivals (one cache line apart)
    unsigned top[32];
    unsigned gen[8][32];     // per-XCD generation
    unsigned giveup[32];
};

__device__ __forceinline__ bool barrier_xcd(Sync* s, unsigned epoch, int nwg) {
    __syncthreads();
    bool ok = true;
    if (threadIdx.x == 0) {
        const unsigned xcd = blockIdx.x & 7, nx = (nwg - xcd + 7) / 8;
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
        const unsigned old = __hip_atomic_fetch_add((gu32*)&s->xcnt[xcd][0], 1u, RLX);
        if (old + 1 == nx * epoch) {                                   // last arriver of this XCD
            const unsigned o2 = __hip_atomic_fetch_add((gu32*)&s->top[0], 1u, RLX);
            if (o2 + 1 == 8u * epoch)                                  // last XCD: release every XCD
                for (int x = 0; x < 8; ++x) __hip_atomic_store((gu32*)&s->gen[x][0], epoch, RLX);
        }
        unsigned spins = 0;
        while (__hip_atomic_load((gu32*)&s->gen[xcd][0], RLX) < epoch) {
            __builtin_amdgcn_s_sleep(2);
            if (++spins > 300000u || __hip_atomic_load((gu32*)&s->giveup[0], RLX)) { __hip_atomic_store((gu32*)&s->giveup[0], 1u, RLX); ok = false; break; }
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    }
    __syncthreads();
    return ok;
}

__global__ __launch_bounds__(256) void k_barrier(Sync* s, int iters, int* slots, int* errors) {
    const int G = gridDim.x, wg = blockIdx.x;
    int bad = 0;
    for (int it = 0; it < iters; ++it) {
        slots[wg * 64 + (threadIdx.x & 63)] = it * 7 + wg;
        if (!barrier_xcd(s, 2 * it + 1, G)) break;
        const int src = (wg + 37 + it) % G;
        if (slots[src * 64 + (threadIdx.x & 63)] != it * 7 + src) ++bad;
        if (!barrier_xcd(s, 2 * it + 2, G)) break;
    }
    if (bad) atomicAdd(errors, bad);
}

// all-gather of N granules (N = G * per; 8-byte {epoch, payload}): two buffers by iteration parity -- a workgroup can be at most ONE
// iteration ahead of the slowest (it passes iteration it only after every workgroup has published it), so the buffer it overwrites in
// iteration it + 2 has been read by everyone.  (A persistent layer cycles through several distinct edges, each with its own buffer.)
// Sweep: the three consumer waves of a workgroup each take a third of the granules, BATCH loads in flight per lane (asm loads with
// sc1, one s_waitcnt per batch), re-reading a batch until all its tags match.
template <int BATCH>
__device__ __forceinline__ bool sweep(const unsigned long long* g, int n, int first, int count, unsigned epoch, float* vec, int lane, gu32* giveup) {
    // granules [first, first + count) of g; lane l takes first + l, first + l + 64, ...
    for (int b0 = 0; b0 < count; b0 += 64 * BATCH) {
        unsigned spins = 0;
        for (;;) {
            unsigned long long v[BATCH];
#pragma unroll
            for (int j = 0; j < BATCH; ++j) {
                int k = b0 + j * 64 + lane;
                k = first + (k < count ? k : count - 1);
                const unsigned long long* p = g + k;
                asm volatile("global_load_dwordx2 %0, %1, off sc1" : "=v"(v[j]) : "v"(p) : "memory");
            }
            if constexpr (BATCH == 16)
                asm volatile("s_waitcnt vmcnt(0)" : "+v"(v[0]), "+v"(v[1]), "+v"(v[2]), "+v"(v[3]), "+v"(v[4]), "+v"(v[5]), "+v"(v[6]), "+v"(v[7]), "+v"(v[8]),
                             "+v"(v[9]), "+v"(v[10]), "+v"(v[11]), "+v"(v[12]), "+v"(v[13]), "+v"(v[14]), "+v"(v[15]));
            else
                asm volatile("s_waitcnt vmcnt(0)" : "+v"(v[0]), "+v"(v[1]), "+v"(v[2]), "+v"(v[3]), "+v"(v[4]), "+v"(v[5]), "+v"(v[6]), "+v"(v[7]));
            bool ok = true;
#pragma unroll
            for (int j = 0; j < BATCH; ++j) {
                const int k = b0 + j * 64 + lane;
                ok &= (unsigned)(v[j] >> 32) == epoch;
                if (k < count) vec[first + k] = __uint_as_float((unsigned)v[j]);
            }
            if (__all(ok)) break;
            __builtin_amdgcn_s_sleep(1);
            if (++spins > 100000u || __hip_atomic_load(giveup, RLX)) { __hip_atomic_store(giveup, 1u, RLX); return false; }
        }
    }
    return true;
}

template <int STREAM>
__global__ __launch_bounds__(256) void k_allgather(unsigned long long* gran, int N, int iters, Sync* s, int* errors, const uint4* stream, size_t stream_n) {
    extern __shared__ float vec[];
    const int G = gridDim.x, wg = blockIdx.x, per = N / G, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    int bad = 0;
    uint4 sink = make_uint4(0, 0, 0, 0);
    gu32* giveup = (gu32*)&s->giveup[0];
    for (int it = 0; it < iters; ++it) {
        const unsigned epoch = it + 1;
        if (wave >= 1) {
            if (wave == 1 && lane < per) {
                const unsigned val = (unsigned)(it * 131 + wg * per + lane);
                __hip_atomic_store((gu64*)(gran + (size_t)(it & 1) * N + wg * per + lane), ((unsigned long long)epoch << 32) | val, RLX);      // one sc1 store per granule
            }
            const int third = (N + 2) / 3, first = (wave - 1) * third, count = min(third, N - first);
            if (count > 0) sweep<16>(gran + (size_t)(it & 1) * N, N, first, count, epoch, vec, lane, giveup);
        } else if (STREAM) {
            // wave 0 keeps HBM loads in flight (a weight stream beside the gather): 16 x 1 KiB per iteration
            typedef __attribute__((ext_vector_type(4))) unsigned int u32x4;
            const size_t base = ((size_t)wg * 65536 + (size_t)it * 16777216) % (stream_n - 65536);
            for (int j = 0; j < 16; ++j) {
                const u32x4 v = __builtin_nontemporal_load((const u32x4*)(stream + base + j * 64 + lane));
                sink.x ^= v.x; sink.y ^= v.y;
            }
        }
        __syncthreads();
        const int probe = (wg * 7 + it * 13 + tid) % N;
        if (__float_as_uint(vec[probe]) != (unsigned)(it * 131 + probe)) ++bad;
        __syncthreads();
    }
    if (sink.x == 0x12345678u && sink.y == 0x9abcdef0u) atomicAdd(errors, 1 << 20);
    if (bad) atomicAdd(errors, bad);
}

int main() {
    setvbuf(stdout, nullptr, _IONBF, 0);
    int cus = 0;
    CK(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, 0));
    printf("CUs %d\n", cus);
    const int G = cus;
    Sync* s; int *slots, *errors; unsigned long long* gran; uint4* stream;
    const size_t stream_n = (size_t)64 << 20;          // 1 GiB of uint4
    CK(hipMalloc(&s, sizeof(Sync)));
    CK(hipMalloc(&slots, G * 64 * 4));
    CK(hipMalloc(&errors, 4));
    CK(hipMalloc(&gran, 2 * 65536 * 8));
    CK(hipMalloc(&stream, stream_n * 16));
    CK(hipMemset(stream, 1, stream_n * 16));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    CK(hipFuncSetAttribute((const void*)k_allgather<0>, hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024));
    CK(hipFuncSetAttribute((const void*)k_allgather<1>, hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024));
    printf("setup done\n");
    auto report = [&](const char* what, float ms, int n_edges) {
        int herr = 0; Sync hs;
        hipError_t le = hipGetLastError();
        if (le != hipSuccess) { printf("%-58s launch failed: %s\n", what, hipGetErrorString(le)); return; }
        (void)hipMemcpy(&herr, errors, 4, hipMemcpyDeviceToHost);
        (void)hipMemcpy(&hs, s, sizeof(Sync), hipMemcpyDeviceToHost);
        printf("%-58s %8.3f ms -> %6.3f us per edge, errors %d%s\n", what, ms, ms * 1e3 / n_edges, herr, hs.giveup[0] ? "  GAVE UP (spin bound hit)" : "");
    };
    for (int rep = 0; rep < 2; ++rep) {
        const int iters = 500;
        CK(hipMemset(s, 0, sizeof(Sync))); CK(hipMemset(errors, 0, 4));
        CK(hipEventRecord(e0));
        hipLaunchKernelGGL(k_barrier, dim3(G), dim3(256), 0, 0, s, iters, slots, errors);
        CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        float ms = 0; CK(hipEventElapsedTime(&ms, e0, e1));
        report("B  XCD-hierarchical barrier (publish 256 B per WG)", ms, 2 * iters);
    }
    for (int N : {1792, 9472, 3584}) {          // granules of 2 bf16 values: hidden (3584 / 2), inter (18944 / 2); one value per granule: hidden
        const int Nn = N / G * G;
        for (int stream_on = 0; stream_on < 2; ++stream_on)
            for (int rep = 0; rep < 2; ++rep) {
                const int iters = 500;
                CK(hipMemset(s, 0, sizeof(Sync))); CK(hipMemset(errors, 0, 4)); CK(hipMemset(gran, 0, 2 * 65536 * 8));
                CK(hipEventRecord(e0));
                if (stream_on) hipLaunchKernelGGL(k_allgather<1>, dim3(G), dim3(256), Nn * 4, 0, gran, Nn, iters, s, errors, stream, stream_n);
                else hipLaunchKernelGGL(k_allgather<0>, dim3(G), dim3(256), Nn * 4, 0, gran, Nn, iters, s, errors, stream, stream_n);
                CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
                float ms = 0; CK(hipEventElapsedTime(&ms, e0, e1));
                char what[128];
                snprintf(what, sizeof(what), "G  granule all-gather, %5d granules (%3d KB)%s", Nn, Nn * 8 / 1024, stream_on ? ", loader wave streaming" : "");
                report(what, ms, iters);
            }
    }
    return 0;
}
