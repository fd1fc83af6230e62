// Micro-benchmark (round 4): does the HBM -> CU streaming rate depend on WHICH bytes the CUs read at the same time?
// Rounds 2-4 found every kernel in which a workgroup streams its OWN contiguous region of the weights (the M ~ 212 GEMM: a 128-row column
// tile = 0.9 MB per workgroup; the persistent decode layer: 0.1-1 MB per CU) stuck near 2.7-2.9 TB/s, while the decode GEMV, whose
// workgroups take interleaved rows (at any instant the chip reads one contiguous band), reaches 6.2 TB/s.  This bench streams the same
// bytes with one workgroup per CU under both assignments, with LDS-DMA (global_load_lds, 1 KiB per wave instruction) and with register
// loads, for several waves per workgroup and depths, no compute:
//   BLOCKED      workgroup c reads bytes [c * S, (c + 1) * S)                      (S = total / workgroups)
//   INTERLEAVED  chunk i of `gran` bytes belongs to workgroup i % G: at any instant all workgroups read one window of G * gran bytes
// Build: hipcc --offload-arch=gfx950 -O3 -o stream_layout stream_layout.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)
typedef __attribute__((address_space(3))) void* lds_ptr_t;
typedef const __attribute__((address_space(1))) void* gbl_ptr_t;
typedef __attribute__((ext_vector_type(4))) unsigned int u32x4;

template <int N> __device__ __forceinline__ void wait_vm() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }

// every wave is an independent loader: block b (1 KiB) of the wave's list -> its own LDS ring of DEPTH KiB, DEPTH - 8 .. DEPTH blocks in flight
// gran = 0: BLOCKED; else INTERLEAVED at `gran` bytes.  Waves of a workgroup split the workgroup's blocks round-robin.
template <int DEPTH, bool LDSDMA>
__global__ __launch_bounds__(1024) void stream_kernel(const char* base, size_t total, size_t gran, int* sink) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), nw = blockDim.x >> 6;
    const size_t G = gridDim.x, c = blockIdx.x;
    size_t per_wg = total / G;
    if (gran) per_wg = per_wg / gran * gran;                        // whole granules only: chunk (k * G + c) must stay inside the buffer
    const size_t nblk = per_wg / 1024;                               // 1 KiB blocks of this workgroup
    char* ring = smem + (size_t)wave * DEPTH * 1024;
    u32x4 acc = {0, 0, 0, 0};
    auto addr = [&](size_t b) -> const char* {                       // b-th block of this workgroup
        const size_t off = b * 1024;
        if (gran == 0) return base + c * per_wg + off + lane * 16;
        const size_t chunk = off / gran, within = off % gran;        // chunk-th chunk of this workgroup = global chunk chunk * G + c
        return base + (chunk * G + c) * gran + within + lane * 16;
    };
    size_t issued = 0;
    if (!LDSDMA) {
        // register stream: 8 independent 16-byte loads per lane in flight per wave (gemv.hip keeps R x 2 chunks = 8)
        for (size_t b = wave; b < nblk; b += (size_t)nw * 8) {
            u32x4 v[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) v[j] = __builtin_nontemporal_load((const u32x4*)addr(b + (size_t)j * nw < nblk ? b + (size_t)j * nw : b));
#pragma unroll
            for (int j = 0; j < 8; ++j) { acc.x ^= v[j].x; acc.y += v[j].y; }
        }
    }
    for (size_t b = wave; LDSDMA && b < nblk; b += nw, ++issued) {
        if (LDSDMA) {
            __builtin_amdgcn_global_load_lds((gbl_ptr_t)addr(b), (lds_ptr_t)(ring + (issued % DEPTH) * 1024), 16, 0, 2);
            if (issued % 8 == 7 && issued + 1 >= DEPTH) wait_vm<DEPTH - 8>();      // keep DEPTH - 8 .. DEPTH blocks in flight
        } else {
            // register stream: 8 loads in flight per wave per batch (like gemv.hip: R x 2 chunks)
            const u32x4 v = __builtin_nontemporal_load((const u32x4*)addr(b));
            acc.x ^= v.x; acc.y += v.y;
        }
    }
    if (LDSDMA) wait_vm<0>();
    __syncthreads();
    if (acc.x == 0x12345u && acc.y == 77u) sink[0] = 1;
    if (threadIdx.x == 0 && smem[3] == 123 && smem[99] == 45) sink[1] = 1;
}

// MIXED: what the M ~ 212 GEMM does to a CU -- waves 0-3 stream the workgroup's share of a cold 448 MB weight region (LDS-DMA, non-temporal),
// waves 4-7 meanwhile re-read ONE 1.5 MB activation panel (the same for every workgroup: L2-resident) `a_ratio` times as many bytes, default
// cache policy.  Do the two streams share one delivery limit (the weights slow down), or do they add?
template <int DEPTH>
__global__ __launch_bounds__(512) void mixed_kernel(const char* wbase, size_t wtotal, const char* abase, size_t abytes, int a_ratio, int* sink) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const size_t G = gridDim.x, c = blockIdx.x;
    const size_t per_wg = wtotal / G, nblk = per_wg / 1024;
    char* ring = smem + (size_t)wave * DEPTH * 1024;
    size_t issued = 0;
    if (wave < 4) {
        for (size_t b = wave; b < nblk; b += 4, ++issued) {
            __builtin_amdgcn_global_load_lds((gbl_ptr_t)(wbase + c * per_wg + b * 1024 + lane * 16), (lds_ptr_t)(ring + (issued % DEPTH) * 1024), 16, 0, 2);
            if (issued % 8 == 7 && issued + 1 >= DEPTH) wait_vm<DEPTH - 8>();
        }
    } else {
        const size_t ablk = abytes / 1024, n = nblk * (size_t)a_ratio;
        for (size_t b = wave - 4; b < n; b += 4, ++issued) {
            const size_t ab = (b + c * 37) % ablk;                    // every workgroup walks the panel from its own offset
            __builtin_amdgcn_global_load_lds((gbl_ptr_t)(abase + ab * 1024 + lane * 16), (lds_ptr_t)(ring + (issued % DEPTH) * 1024), 16, 0, 0);
            if (issued % 8 == 7 && issued + 1 >= DEPTH) wait_vm<DEPTH - 8>();
        }
    }
    wait_vm<0>();
    __syncthreads();
    if (threadIdx.x == 0 && smem[3] == 123 && smem[99] == 45) sink[1] = 1;
}

int main() {
    setvbuf(stdout, nullptr, _IONBF, 0);
    int cus = 0;
    CK(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, 0));
    const size_t copy = (size_t)448 << 20;                    // ~ one decoder layer of weights
    const int ncopy = 6;                                      // rotate through 2.6 GB: HBM-cold every launch
    char* buf; int* sink;
    CK(hipMalloc(&buf, copy * ncopy));
    CK(hipMemset(buf, 1, copy * ncopy));
    CK(hipMalloc(&sink, 16));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    CK(hipFuncSetAttribute((const void*)stream_kernel<16, true>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    CK(hipFuncSetAttribute((const void*)stream_kernel<32, true>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    CK(hipFuncSetAttribute((const void*)stream_kernel<48, true>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    printf("CUs %d, %zu MB per launch, one workgroup per CU\n", cus, copy >> 20);
    auto run = [&](const char* what, auto kern, int waves, int depth_kb, size_t gran) -> int {
        float best = 1e9f, sum = 0;
        const int reps = 6;
        for (int r = 0; r < reps; ++r) {
            const char* p = buf + (size_t)(r % ncopy) * copy;
            CK(hipEventRecord(e0));
            hipLaunchKernelGGL(kern, dim3(cus), dim3(waves * 64), (size_t)waves * depth_kb * 1024, 0, p, copy, gran, sink);
            CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
            hipError_t le = hipGetLastError();
            if (le != hipSuccess) { printf("%-70s launch failed: %s\n", what, hipGetErrorString(le)); return 0; }
            float ms = 0; CK(hipEventElapsedTime(&ms, e0, e1));
            if (r) { best = ms < best ? ms : best; sum += ms; }
        }
        printf("%-78s avg %7.1f us  best %7.1f us = %5.2f TB/s (%5.1f GB/s per CU)\n", what, sum / (reps - 1) * 1e3, best * 1e3, copy / best / 1e9,
               copy / best / 1e6 / cus);
        return 0;
    };
    char what[160];
    for (size_t gran : {(size_t)0, (size_t)1024, (size_t)16384, (size_t)65536, (size_t)7168}) {
        for (int waves : {1, 2, 4, 8}) {
            const int depth = waves == 1 ? 48 : (waves == 2 ? 48 : (waves == 4 ? 32 : 16));
            snprintf(what, sizeof(what), "LDS-DMA nt, %s%zu B, %d loader wave(s) x %d KB ring", gran ? "INTERLEAVED at " : "BLOCKED ", gran, waves, depth);
            if (depth == 48) run(what, stream_kernel<48, true>, waves, 48, gran);
            else if (depth == 32) run(what, stream_kernel<32, true>, waves, 32, gran);
            else run(what, stream_kernel<16, true>, waves, 16, gran);
        }
        for (int waves : {4, 16}) {
            snprintf(what, sizeof(what), "register loads nt, %s%zu B, %d waves", gran ? "INTERLEAVED at " : "BLOCKED ", gran, waves);
            run(what, stream_kernel<16, false>, waves, 0, gran);
        }
    }
    // mixed streams: weights (HBM) + a shared L2-resident panel
    CK(hipFuncSetAttribute((const void*)mixed_kernel<16>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    char* panel;
    CK(hipMalloc(&panel, (size_t)3 << 20));
    CK(hipMemset(panel, 2, (size_t)3 << 20));
    for (int a_ratio : {0, 1, 2, 4}) {
        float best = 1e9f;
        for (int r = 0; r < 6; ++r) {
            const char* p = buf + (size_t)(r % ncopy) * copy;
            CK(hipEventRecord(e0));
            hipLaunchKernelGGL(mixed_kernel<16>, dim3(cus), dim3(512), (size_t)8 * 16 * 1024, 0, p, copy, panel, (size_t)1536 * 1024, a_ratio, sink);
            CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
            float ms = 0; CK(hipEventElapsedTime(&ms, e0, e1));
            if (r && ms < best) best = ms;
        }
        printf("MIXED 4 weight waves (HBM, nt) + 4 panel waves (1.5 MB, L2-resident), panel bytes = %d x weight bytes: best %7.1f us -> weights %5.2f TB/s (%5.1f GB/s per CU), "
               "panel %5.2f TB/s, together %5.2f TB/s\n", a_ratio, best * 1e3, copy / best / 1e9, copy / best / 1e6 / cus, copy * (double)a_ratio / best / 1e9,
               copy * (1.0 + a_ratio) / best / 1e9);
    }
    return 0;
}
