// Micro-benchmark: HBM -> LDS streaming rate of the GEMM's LDS-DMA weight pipeline (global_load_lds_dwordx4) as a function
// of ring depth, with the GEMM's exact per-stage protocol (counted vmcnt wait, s_barrier, issue) and no compute.
// One 512-thread workgroup per CU streams a [128 x K] bf16 weight panel in 16 KB stages (128 rows x 128 B).
// Build: hipcc --offload-arch=gfx950 -O3 -o glds_stream glds_stream.hip
#include <hip/hip_runtime.h>
#include <cstdio>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)
typedef __attribute__((address_space(3))) void* lds_ptr_t;
typedef const __attribute__((address_space(1))) void* gbl_ptr_t;

// AROWS > 0: every workgroup also stages the same AROWS x K activation panel (L2-resident) each stage, like the GEMM's A tile
template <int NBUF, int ROWS, int AROWS = 0, int ROT = 0, int ROWB = 128, int MODE = 0>
__global__ __launch_bounds__(512) void stream_kernel(const char* W, size_t row_bytes, int stages, int* sink, const char* A = nullptr) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int STAGE = (ROWS + AROWS) * ROWB, BLK = STAGE / 1024, PER_WAVE = (BLK + 7) / 8, D = NBUF - 1;
    constexpr int RPB = 1024 / ROWB, LPR = ROWB / 16;        // rows per 1 KiB block, lanes per row
    constexpr int BLK_W = ROWS / RPB;
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const char* src[PER_WAVE];
#pragma unroll
    for (int j = 0; j < PER_WAVE; ++j) {
        const int blk = min(wave + 8 * j, BLK - 1), row = blk * RPB + lane / LPR;
        if (blk < BLK_W) src[j] = W + ((size_t)(MODE ? 0 : blockIdx.x) * ROWS + row) * row_bytes + (lane % LPR) * 16;
        else src[j] = A + (size_t)(row - ROWS) * row_bytes + (lane % LPR) * 16;
    }
    // ROT: every workgroup walks K from its own starting stage (wraps), so the workgroups of an XCD do not all hit the same
    // activation lines (same L2 channel) at the same time
    const int rot = ROT ? (int)((blockIdx.x >> 3) * ROT) % stages : 0;
    auto issue = [&](int st, int buf) {
        int sr = st + rot;
        if (sr >= stages) sr -= stages;
        if (MODE == 2) {
            uint4 t[PER_WAVE];
#pragma unroll
            for (int j = 0; j < PER_WAVE; ++j) t[j] = *(const uint4*)(src[j] + (size_t)sr * ROWB);
#pragma unroll
            for (int j = 0; j < PER_WAVE; ++j) *(uint4*)(smem + buf * STAGE + min(wave + 8 * j, BLK - 1) * 1024 + lane * 16) = t[j];
            return;
        }
#pragma unroll
        for (int j = 0; j < PER_WAVE; ++j)
            __builtin_amdgcn_global_load_lds((gbl_ptr_t)(src[j] + (size_t)sr * ROWB), (lds_ptr_t)(smem + buf * STAGE + min(wave + 8 * j, BLK - 1) * 1024), 16, 0, 0);
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

template <int NBUF, int ROWS, int AROWS = 0, int ROT = 0, int ROWB = 128, int MODE = 0> int run(const char* W, int K, int* sink, int n_copies, size_t copy_bytes, int wgs = 256, const char* A = nullptr) {
    constexpr int LDS = NBUF * (ROWS + AROWS) * ROWB;
    CK(hipFuncSetAttribute((const void*)stream_kernel<NBUF, ROWS, AROWS, ROT, ROWB, MODE>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    const int stages = K * 2 / ROWB;
    float best = 1e9f, sum = 0;
    const int reps = 6;
    for (int r = 0; r < reps; ++r) {
        const char* Wc = W + (size_t)(r % n_copies) * copy_bytes;          // rotate copies: cold in the infinity cache
        CK(hipEventRecord(e0));
        hipLaunchKernelGGL((stream_kernel<NBUF, ROWS, AROWS, ROT, ROWB, MODE>), dim3(wgs), dim3(512), LDS, 0, Wc, (size_t)K * 2, stages, sink, A);
        CK(hipEventRecord(e1));
        CK(hipEventSynchronize(e1));
        float ms = 0;
        CK(hipEventElapsedTime(&ms, e0, e1));
        if (r > 0) { sum += ms; best = ms < best ? ms : best; }
    }
    const double bytes = (double)wgs * ROWS * K * 2;      // weight (HBM) bytes only
    printf("mode %d rowB %3d rot %2d wgs %3d A rows %3d | W rows/stage %3d ring %d (%3d KB LDS): avg %.1f us  best %.1f us  -> %.2f TB/s (best %.2f), %.3f us per 128 B of K\n", MODE, ROWB, ROT, wgs, AROWS, ROWS, NBUF,
           LDS / 1024, sum / (reps - 1) * 1e3, best * 1e3, bytes / (sum / (reps - 1) * 1e-3) / 1e12, bytes / (best * 1e-3) / 1e12,
           sum / (reps - 1) * 1e3 / stages * (128 / ROWB));
    return 0;
}

// Role-split variant: waves 0-3 stream only the weight rows (ring NBW deep), waves 4-7 only the shared activation panel (ring NBA
// deep); each wave waits on its own loads (vmcnt is in order per wave), then the workgroup barrier publishes the stage.
template <int NBA, int NBW, int ROWS, int AROWS>
__global__ __launch_bounds__(512) void role_kernel(const char* W, size_t row_bytes, int stages, int* sink, const char* A) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int A_BYTES = AROWS * 128, W_BYTES = ROWS * 128, W_RING = NBA * A_BYTES;
    constexpr int PW = ROWS / 8 / 4, PA = AROWS / 8 / 4, PMAX = PA > PW ? PA : PW, DA = NBA - 1, DW = NBW - 1;
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const bool w_wave = wave < 4;
    const int cnt = w_wave ? PW : PA, dist = w_wave ? DW : DA, nslots = w_wave ? NBW : NBA;
    const char* src[PMAX];
    int loff[PMAX];
#pragma unroll
    for (int j = 0; j < PMAX; ++j) {
        const int blk = min((wave & 3) + 4 * j, (w_wave ? ROWS : AROWS) / 8 - 1), row = blk * 8 + (lane >> 3);
        src[j] = w_wave ? W + ((size_t)blockIdx.x * ROWS + row) * row_bytes + (lane & 7) * 16 : A + (size_t)row * row_bytes + (lane & 7) * 16;
        loff[j] = (w_wave ? W_RING : 0) + blk * 1024;
    }
    const int slot_bytes = w_wave ? W_BYTES : A_BYTES;
    auto issue = [&](int st, int slot) {
#pragma unroll
        for (int j = 0; j < PMAX; ++j) {
            if (j >= cnt) break;
            __builtin_amdgcn_global_load_lds((gbl_ptr_t)(src[j] + (size_t)st * 128), (lds_ptr_t)(smem + loff[j] + slot * slot_bytes), 16, 0, 0);
        }
    };
    for (int d = 0; d < dist; ++d)
        if (d < stages) issue(d, d);
    int islot = dist % nslots;
    for (int i = 0; i < stages; ++i) {
        if (w_wave) {
            if (i + DW <= stages) asm volatile("s_waitcnt vmcnt(%0)" ::"n"((DW - 1) * PW) : "memory");
            else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        } else {
            if (i + DA <= stages) asm volatile("s_waitcnt vmcnt(%0)" ::"n"((DA - 1) * PA) : "memory");
            else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        __builtin_amdgcn_s_barrier();
        if (i + dist < stages) issue(i + dist, islot);
        islot = islot + 1 == nslots ? 0 : islot + 1;
    }
    __syncthreads();
    if (threadIdx.x == 0 && smem[0] == 123 && smem[5] == 77) sink[0] = 1;
}

template <int NBA, int NBW, int ROWS, int AROWS> int run_role(const char* W, int K, int* sink, int n_copies, size_t copy_bytes, int wgs, const char* A) {
    constexpr int LDS = NBA * AROWS * 128 + NBW * ROWS * 128;
    CK(hipFuncSetAttribute((const void*)role_kernel<NBA, NBW, ROWS, AROWS>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    const int stages = K * 2 / 128;
    float best = 1e9f, sum = 0;
    const int reps = 6;
    for (int r = 0; r < reps; ++r) {
        const char* Wc = W + (size_t)(r % n_copies) * copy_bytes;
        CK(hipEventRecord(e0));
        hipLaunchKernelGGL((role_kernel<NBA, NBW, ROWS, AROWS>), dim3(wgs), dim3(512), LDS, 0, Wc, (size_t)K * 2, stages, sink, A);
        CK(hipEventRecord(e1));
        CK(hipEventSynchronize(e1));
        float ms = 0;
        CK(hipEventElapsedTime(&ms, e0, e1));
        if (r > 0) { sum += ms; best = ms < best ? ms : best; }
    }
    const double bytes = (double)wgs * ROWS * K * 2;
    printf("role-split wgs %3d A rows %3d ring %d | W rows %3d ring %d (%3d KB LDS): avg %.1f us best %.1f us -> %.2f TB/s, %.3f us per stage\n", wgs, AROWS, NBA, ROWS, NBW,
           LDS / 1024, sum / (reps - 1) * 1e3, best * 1e3, bytes / (sum / (reps - 1) * 1e-3) / 1e12, sum / (reps - 1) * 1e3 / stages);
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
    if (run<3, 128>(W, K, sink, n_copies, copy_bytes)) return 1;
    // with the shared activation panel (256 rows x K, 7 MB: L2 / infinity-cache resident)
    char* A;
    CK(hipMalloc(&A, (size_t)256 * K * 2));
    CK(hipMemset(A, 2, (size_t)256 * K * 2));
    if (run<3, 128, 256>(W, K, sink, n_copies, copy_bytes, 256, A)) return 1;           // the 256x128-tile GEMM's traffic, unified ring
    if (run_role<2, 5, 128, 256>(W, K, sink, n_copies, copy_bytes, 256, A)) return 1;
    if (run_role<2, 6, 128, 256>(W, K, sink, n_copies, copy_bytes, 256, A)) return 1;
    if (run_role<3, 4, 128, 256>(W, K, sink, n_copies, copy_bytes, 256, A)) return 1;
    if (run_role<3, 3, 128, 256>(W, K, sink, n_copies, copy_bytes, 256, A)) return 1;
    if (run_role<2, 3, 192, 256>(W, K, sink, n_copies, copy_bytes, 198, A)) return 1;
    if (run_role<2, 4, 192, 256>(W, K, sink, n_copies, copy_bytes, 198, A)) return 1;
    return 0;
}
