// Micro-benchmark: HBM -> LDS streaming rate of the GEMM's LDS-DMA weight pipeline (global_load_lds_dwordx4) as a function
// of ring depth, with the GEMM's exact per-stage protocol (counted vmcnt wait, s_barrier, issue) and no compute.
// One 512-thread workgroup per CU streams a [128 x K] bf16 weight panel in 16 KB stages (128 rows x 128 B).
// Build: hipcc --offload-arch=gfx950 -O3 -o glds_stream glds_stream.hip
#include <hip/hip_runtime.h>
#include <cstdio>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)
typedef __attribute__((address_space(3))) void* lds_ptr_t;
typedef const __attribute__((address_space(1))) void* gbl_ptr_t;

template <int NBUF, int ROWS>
__global__ __launch_bounds__(512) void stream_kernel(const char* W, size_t row_bytes, int stages, int* sink) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int STAGE = ROWS * 128, BLK = STAGE / 1024, PER_WAVE = BLK / 8, D = NBUF - 1;
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const char* src[PER_WAVE];
#pragma unroll
    for (int j = 0; j < PER_WAVE; ++j) {
        const int blk = wave + 8 * j, row = blk * 8 + (lane >> 3);
        src[j] = W + ((size_t)blockIdx.x * ROWS + row) * row_bytes + (lane & 7) * 16;
    }
    auto issue = [&](int st, int buf) {
#pragma unroll
        for (int j = 0; j < PER_WAVE; ++j)
            __builtin_amdgcn_global_load_lds((gbl_ptr_t)(src[j] + (size_t)st * 128), (lds_ptr_t)(smem + buf * STAGE + (wave + 8 * j) * 1024), 16, 0, 0);
    };
#pragma unroll
    for (int d = 0; d < D; ++d)
        if (d < stages) issue(d, d);
    int nbuf = D % NBUF;
    for (int i = 0; i < stages; ++i) {
        if (i + D <= stages) asm volatile("s_waitcnt vmcnt(%0)" ::"n"((D - 1) * PER_WAVE) : "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        if (i + D < stages) issue(i + D, nbuf);
        nbuf = nbuf + 1 == NBUF ? 0 : nbuf + 1;
    }
    __syncthreads();
    if (threadIdx.x == 0 && smem[0] == 123 && smem[5] == 77) sink[0] = 1;
}

template <int NBUF, int ROWS> int run(const char* W, int K, int* sink, int n_copies, size_t copy_bytes) {
    constexpr int LDS = NBUF * ROWS * 128;
    CK(hipFuncSetAttribute((const void*)stream_kernel<NBUF, ROWS>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    const int stages = K * 2 / 128, wgs = 256;
    float best = 1e9f, sum = 0;
    const int reps = 6;
    for (int r = 0; r < reps; ++r) {
        const char* Wc = W + (size_t)(r % n_copies) * copy_bytes;          // rotate copies: cold in the infinity cache
        CK(hipEventRecord(e0));
        hipLaunchKernelGGL((stream_kernel<NBUF, ROWS>), dim3(wgs), dim3(512), LDS, 0, Wc, (size_t)K * 2, stages, sink);
        CK(hipEventRecord(e1));
        CK(hipEventSynchronize(e1));
        float ms = 0;
        CK(hipEventElapsedTime(&ms, e0, e1));
        if (r > 0) { sum += ms; best = ms < best ? ms : best; }
    }
    const double bytes = (double)wgs * ROWS * K * 2;
    printf("rows/stage %3d (%2d KB) ring %d (%3d KB LDS, %2d KB in flight): avg %.1f us  best %.1f us  -> %.2f TB/s (best %.2f), %.3f us/stage\n", ROWS, ROWS / 8, NBUF,
           LDS / 1024, (NBUF - 1) * ROWS / 8, sum / (reps - 1) * 1e3, best * 1e3, bytes / (sum / (reps - 1) * 1e-3) / 1e12, bytes / (best * 1e-3) / 1e12,
           sum / (reps - 1) * 1e3 / stages);
    return 0;
}

int main() {
    const int K = 14336;                       // 256 CUs x 128 rows x 28 KB = 940 MB per copy at ROWS=128
    const size_t copy_bytes = (size_t)256 * 256 * K * 2;   // sized for ROWS = 256
    const int n_copies = 2;
    char* W;
    int* sink;
    CK(hipMalloc(&W, copy_bytes * n_copies));
    CK(hipMemset(W, 1, copy_bytes * n_copies));
    CK(hipMalloc(&sink, 4));
    if (run<2, 128>(W, K, sink, n_copies, copy_bytes)) return 1;
    if (run<3, 128>(W, K, sink, n_copies, copy_bytes)) return 1;
    if (run<4, 128>(W, K, sink, n_copies, copy_bytes)) return 1;
    if (run<5, 128>(W, K, sink, n_copies, copy_bytes)) return 1;
    if (run<6, 128>(W, K, sink, n_copies, copy_bytes)) return 1;
    if (run<8, 128>(W, K, sink, n_copies, copy_bytes)) return 1;
    if (run<9, 128>(W, K, sink, n_copies, copy_bytes)) return 1;
    if (run<2, 256>(W, K, sink, n_copies, copy_bytes)) return 1;
    if (run<3, 256>(W, K, sink, n_copies, copy_bytes)) return 1;
    if (run<4, 256>(W, K, sink, n_copies, copy_bytes)) return 1;
    return 0;
}
