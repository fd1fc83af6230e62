// Micro-benchmark behind DESIGN.md 4.1 (round 4): the M <= 224 products of a steady prefill (gate/up: 212 x 37888 x 3584) with the WEIGHTS
// streamed straight into the MFMA B-operand registers (no LDS stage for them) and only the activation panel staged through LDS.
//   workgroup = 8 waves, wave w owns 32 weight rows (one MFMA column tile) against all 7 row tiles of the panel (acc 7 x f32x16);
//   per 64-wide K slab: panel rows 256 x 128 B by LDS-DMA into a ring, the wave's 4 weight fragments by global loads PF slabs ahead.
// Two weight layouts:  ROWMAJOR  lane (r, h) loads W[n0 + r][16 s + 8 h ..]   (32 rows x 32 B per wave-instruction)
//                      PACKED    fragment (n tile, k16 step) = 1 KiB contiguous, lane-major (a load-time repack; fully coalesced)
// Build:  hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/micro/skinny_gemm.hip -o tools/micro/skinny_gemm
#include <hip/hip_runtime.h>
#include <hip/hip_bf16.h>
#include <cstdio>
#include <cstdlib>
#include <cstdint>
#include <cmath>
#include <cstring>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

typedef __attribute__((address_space(3))) void* lds_ptr_t;
typedef const __attribute__((address_space(1))) void* gbl_ptr_t;
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

constexpr int WAVES = 8, NT = WAVES * 64, MT = 7, ROWS_LDS = 256, STAGE = ROWS_LDS * 128;

static __device__ __forceinline__ int acc_row(int reg, int lane) { return (reg & 3) + 8 * (reg >> 2) + 4 * (lane >> 5); }

template <bool PACKED, int D, int NST, int NSL = 0>      // NSL > 0: the slab count per workgroup is this constant and the loop is fully unrolled
__global__ __launch_bounds__(NT, 1) void skinny_kernel(const uint16_t* __restrict__ A, int lda, const uint16_t* __restrict__ W, int ldw, float* __restrict__ C,
                                                       int M, int N, int K, int nsplit, const char* __restrict__ zeros) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int r = lane & 31, h = lane >> 5;
    const int ntile_wg = blockIdx.x / nsplit, ks = blockIdx.x - ntile_wg * nsplit;
    const int nt = ntile_wg * WAVES + wave;                 // this wave's 32-row weight tile
    const int slabs_total = K / 64, per = (slabs_total + nsplit - 1) / nsplit;
    const int sl0 = ks * per, sl1 = min(slabs_total, sl0 + per), n_sl = NSL > 0 ? NSL : sl1 - sl0;

    // panel staging: wave w issues the 1-KiB blocks w, w + 8, w + 16, w + 24 of a stage (8 rows each)
    const char* asrc[4];
    int aoff[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int blk = wave + WAVES * j, row = blk * 8 + (lane >> 3);
        const int c = (lane & 7) ^ ((row >> 1) & 7);
        asrc[j] = (const char*)(A + (size_t)min(row, M - 1) * lda) + c * 16 + (size_t)sl0 * 128;
        aoff[j] = blk * 1024;
    }
    // weight stream of this wave
    const char* wsrc;
    size_t wstep;       // bytes per k16 step
    if (PACKED) { wsrc = (const char*)W + ((size_t)nt * (K / 16) + (size_t)sl0 * 4) * 1024 + lane * 16; wstep = 1024; }
    else { wsrc = (const char*)(W + (size_t)min(nt * 32 + r, N - 1) * ldw) + (size_t)sl0 * 128 + h * 16; wstep = 32; }

    f32x16 acc[MT];
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[i][e] = 0.0f;

    u32x4 wr[D][4];
    auto issue_a = [&](int sl, int buf) {
        char* base = smem + buf * STAGE;
#pragma unroll
        for (int j = 0; j < 4; ++j) __builtin_amdgcn_global_load_lds((gbl_ptr_t)(asrc[j] + (size_t)sl * 128), (lds_ptr_t)(base + aoff[j]), 16, 0, 0);
    };
    auto issue_w = [&](int sl, u32x4 (&dst)[4]) {
#pragma unroll
        for (int s = 0; s < 4; ++s) {       // past the last slab: zeros (the tail of a slab count that is not a multiple of D multiplies to nothing)
            const char* g = sl < n_sl ? wsrc + ((size_t)sl * 4 + s) * wstep : zeros + lane * 16;
            dst[s] = __builtin_nontemporal_load((const u32x4*)g);
        }
    };
    const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) char*)smem;
    unsigned sw[4];
#pragma unroll
    for (int s4 = 0; s4 < 4; ++s4) sw[s4] = r * 128 + (((2 * s4 + h) ^ ((r >> 1) & 7)) << 4);
    u32x4 fa[2][MT];
    auto read7 = [&](unsigned addr, u32x4 (&f)[MT]) {
        asm volatile("ds_read_b128 %0, %1" : "=v"(f[0]) : "v"(addr));
        asm volatile("ds_read_b128 %0, %1 offset:4096" : "=v"(f[1]) : "v"(addr));
        asm volatile("ds_read_b128 %0, %1 offset:8192" : "=v"(f[2]) : "v"(addr));
        asm volatile("ds_read_b128 %0, %1 offset:12288" : "=v"(f[3]) : "v"(addr));
        asm volatile("ds_read_b128 %0, %1 offset:16384" : "=v"(f[4]) : "v"(addr));
        asm volatile("ds_read_b128 %0, %1 offset:20480" : "=v"(f[5]) : "v"(addr));
        asm volatile("ds_read_b128 %0, %1 offset:24576" : "=v"(f[6]) : "v"(addr));
    };
    // The first pass of the loop (t0 = -D) only issues slabs 0 .. D-1 (panel block then weight fragments, slab by slab: the order the counted
    // waits rely on; a separate prologue gets its weight loads sunk below its DMA by hipcc, and the loop header then needs a vmcnt(0)).
#pragma unroll (NSL > 0 ? (NSL + D - 1) / D + 1 : 1)
    for (int t0 = -D; t0 < n_sl; t0 += D) {
        const bool run = t0 >= 0;
#pragma unroll
        for (int d = 0; d < D; ++d) {
            const int t = t0 + d;          // (slabs past n_sl run on zero weights)
            if (!run) {
                issue_a(min(t + D, n_sl - 1), (t + D) % NST);
                __builtin_amdgcn_sched_barrier(0);
                issue_w(t + D, wr[d]);
                __builtin_amdgcn_sched_barrier(0);
                continue;
            }
            // slab t landed when at most the D - 1 younger slabs (8 loads each) are still outstanding (as the builtin, so that hipcc's own wait
            // insertion knows the slab's weight registers have landed)
            __builtin_amdgcn_s_waitcnt(((((D - 1) * 8) & 0x30) << 10) | (((D - 1) * 8) & 0xF) | 0x0F70);
            __builtin_amdgcn_s_barrier();
            const int buf = t % NST;
            issue_a(min(t + D, n_sl - 1), (t + D) % NST);       // into the stage slab t - 1 used: every wave is past it (the tail re-loads the last slab: uniform counts)
            // fragment reads as inline asm (hipcc would order every ds_read it can see behind ALL outstanding LDS-DMA: vmcnt(0)), double-buffered
            // over the 4 k16 steps: step s + 1's seven reads are issued before step s's MFMAs
            const unsigned sb = lds0 + buf * STAGE;
            read7(sb + sw[0], fa[0]);
#pragma unroll
            for (int s = 0; s < 4; ++s) {
                if (s < 3) {
                    read7(sb + sw[s + 1], fa[(s + 1) & 1]);
                    asm volatile("s_waitcnt lgkmcnt(7)" : "+v"(fa[s & 1][0]), "+v"(fa[s & 1][1]), "+v"(fa[s & 1][2]), "+v"(fa[s & 1][3]), "+v"(fa[s & 1][4]), "+v"(fa[s & 1][5]), "+v"(fa[s & 1][6]));
                } else {
                    asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(fa[s & 1][0]), "+v"(fa[s & 1][1]), "+v"(fa[s & 1][2]), "+v"(fa[s & 1][3]), "+v"(fa[s & 1][4]), "+v"(fa[s & 1][5]), "+v"(fa[s & 1][6]));
                }
#pragma unroll
                for (int i = 0; i < MT; ++i)
                    acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, fa[s & 1][i]), __builtin_bit_cast(bf16x8, wr[d][s]), acc[i], 0, 0, 0);
            }
            issue_w(t + D, wr[d]);
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    float* slab = C + (size_t)ks * M * N;
    const int n = nt * 32 + r;
    if (n < N) {
#pragma unroll
        for (int i = 0; i < MT; ++i)
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const int m = i * 32 + acc_row(e, lane);
                if (m < M) slab[(size_t)m * N + n] = acc[i][e];
            }
    }
}

static uint16_t f2bf(float f) { uint32_t u; memcpy(&u, &f, 4); u += 0x7FFF + ((u >> 16) & 1); return (uint16_t)(u >> 16); }
static float bf2f(uint16_t b) { uint32_t u = (uint32_t)b << 16; float f; memcpy(&f, &u, 4); return f; }

template <bool PACKED, int D, int NST, int NSL = 0> void run(const char* name, const uint16_t* dA, const std::vector<uint16_t*>& dW, float* dC, int M, int N, int K, int nsplit,
                                               const std::vector<uint16_t>& hA, const std::vector<uint16_t>& hW) {
    const size_t lds = (size_t)NST * STAGE;
    CK(hipFuncSetAttribute((const void*)skinny_kernel<PACKED, D, NST, NSL>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    const int grid = (N / 32 / WAVES) * nsplit;
    if ((K / 64) % nsplit != 0 || N % 256 != 0 || (NSL > 0 && K / 64 / nsplit != NSL)) { printf("%s: shape not supported by this benchmark\n", name); return; }
    static char* zeros = nullptr;
    if (!zeros) { CK(hipMalloc(&zeros, 1024)); CK(hipMemset(zeros, 0, 1024)); }
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int i = 0; i < 3; ++i) hipLaunchKernelGGL((skinny_kernel<PACKED, D, NST, NSL>), dim3(grid), dim3(NT), lds, 0, dA, K, dW[i % dW.size()], K, dC, M, N, K, nsplit, zeros);
    CK(hipDeviceSynchronize());
    const int reps = 20;
    CK(hipEventRecord(e0));
    for (int i = 0; i < reps; ++i) hipLaunchKernelGGL((skinny_kernel<PACKED, D, NST, NSL>), dim3(grid), dim3(NT), lds, 0, dA, K, dW[i % dW.size()], K, dC, M, N, K, nsplit, zeros);
    CK(hipEventRecord(e1));
    CK(hipDeviceSynchronize());
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    const double us = ms * 1e3 / reps;
    // check a sample of outputs of the LAST launch (weights copy (reps - 1) % n: every copy holds the same values)
    std::vector<float> hC((size_t)nsplit * M * N);
    CK(hipMemcpy(hC.data(), dC, hC.size() * 4, hipMemcpyDeviceToHost));
    double worst = 0;
    for (int q = 0; q < 400; ++q) {
        const int m = (q * 37) % M, n = (int)(((size_t)q * 7919 + 13) % N);
        double ref = 0;
        for (int k = 0; k < K; ++k) ref += (double)bf2f(hA[(size_t)m * K + k]) * bf2f(hW[(size_t)n * K + k]);
        double got = 0;
        for (int s = 0; s < nsplit; ++s) got += hC[(size_t)s * M * N + (size_t)m * N + n];
        worst = fmax(worst, fabs(got - ref) / (fabs(ref) + 1e-2));
    }
    printf("%-34s split %d grid %4d  %7.2f us  W %.2f TB/s  %.0f TF/s  max rel err %.2e\n", name, nsplit, grid, us, (double)N * K * 2 / us * 1e-6,
           2.0 * M * N * K / us * 1e-6, worst);
    fflush(stdout);
}

int main(int argc, char** argv) {
    setvbuf(stdout, nullptr, _IONBF, 0);
    const int M = 212, N = argc > 1 ? atoi(argv[1]) : 37888, K = argc > 2 ? atoi(argv[2]) : 3584;
    std::vector<uint16_t> hA((size_t)M * K), hW((size_t)N * K), hP((size_t)N * K);
    uint64_t s = 0x9E3779B97F4A7C15ull;
    auto rnd = [&]() { s ^= s << 13; s ^= s >> 7; s ^= s << 17; return (float)((s >> 40) & 0xFFFF) / 65536.0f - 0.5f; };
    for (auto& v : hA) v = f2bf(rnd());
    for (auto& v : hW) v = f2bf(rnd() * 0.1f);
    // packed: fragment (nt, k16) lane (r, h) <- W[nt*32 + r][16 k16 + 8 h .. + 7]
    for (int nt = 0; nt < N / 32; ++nt)
        for (int k16 = 0; k16 < K / 16; ++k16)
            for (int lane = 0; lane < 64; ++lane)
                memcpy(&hP[(((size_t)nt * (K / 16) + k16) * 64 + lane) * 8], &hW[(size_t)(nt * 32 + (lane & 31)) * K + 16 * k16 + 8 * (lane >> 5)], 16);
    const int copies = (int)fmax(1.0, ceil(800e6 / ((double)N * K * 2)));
    uint16_t* dA; float* dC;
    CK(hipMalloc(&dA, hA.size() * 2)); CK(hipMemcpy(dA, hA.data(), hA.size() * 2, hipMemcpyHostToDevice));
    CK(hipMalloc(&dC, (size_t)8 * M * N * 4));
    std::vector<uint16_t*> dW(copies), dP(copies);
    for (int c = 0; c < copies; ++c) {
        CK(hipMalloc(&dW[c], hW.size() * 2)); CK(hipMemcpy(dW[c], hW.data(), hW.size() * 2, hipMemcpyHostToDevice));
        CK(hipMalloc(&dP[c], hP.size() * 2)); CK(hipMemcpy(dP[c], hP.data(), hP.size() * 2, hipMemcpyHostToDevice));
    }
    printf("M %d N %d K %d, %d weight copies (HBM-cold)\n", M, N, K, copies);
    for (int nsplit : {1, 2, 4}) {
        run<false, 2, 3>("row-major W, 2 slabs ahead", dA, dW, dC, M, N, K, nsplit, hA, hW);
        run<false, 3, 4>("row-major W, 3 slabs ahead", dA, dW, dC, M, N, K, nsplit, hA, hW);
        run<false, 4, 5>("row-major W, 4 slabs ahead", dA, dW, dC, M, N, K, nsplit, hA, hW);
        run<true, 2, 3>("packed W,    2 slabs ahead", dA, dP, dC, M, N, K, nsplit, hA, hW);
        run<true, 3, 4>("packed W,    3 slabs ahead", dA, dP, dC, M, N, K, nsplit, hA, hW);
        run<true, 4, 5>("packed W,    4 slabs ahead", dA, dP, dC, M, N, K, nsplit, hA, hW);
        run<true, 2, 3, 28>("packed W, 2 ahead, unrolled 28", dA, dP, dC, M, N, K, nsplit, hA, hW);
        run<true, 3, 4, 28>("packed W, 3 ahead, unrolled 28", dA, dP, dC, M, N, K, nsplit, hA, hW);
        run<true, 3, 4, 14>("packed W, 3 ahead, unrolled 14", dA, dP, dC, M, N, K, nsplit, hA, hW);
    }
    return 0;
}
