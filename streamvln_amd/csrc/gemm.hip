// MFMA GEMM  C[M,N] = epi(A[M,K] . W[N,K]^T + bias) + res   for the dense QKV / MLP / projector /
// patch-embed products (prefill and vision; decode uses gemv.hip).
//
// Structure (gfx950): 128x128 output tile per 256-thread workgroup (4 waves as 2x2, each wave a
// 64x64 sub-tile = 2x2 MFMA 32x32 accumulators), K consumed 128 bytes per row per stage
// (64 bf16 / 32 fp32), two LDS stages, register-staged prefetch of stage t+1 issued before the
// MFMAs of stage t and written after them (one barrier per stage).  LDS rows are 128 B; the
// 16-byte chunk index is XOR-swizzled with (row>>1)&7 so the 16 lanes of every ds_read_b128
// group hit 16 distinct 16-byte slots of the 256-byte bank row (conflict-free).
// Workgroup ids are remapped so that the 8 XCDs each own a contiguous band of tiles (operand
// panels shared through the XCD's private L2).
//
// Roofline: MFMA-bound for M >= ~512; at M ~ 212 (steady prefill turn) the weight stream
// (HBM) and MFMA times are comparable.  Algorithmic flops = 2*M*N*K.
#include "common.h"
#include "kernels.h"

namespace svln {

namespace {

constexpr int BM = 128, BN = 128, ROWB = 128;              // ROWB = bytes of K per row per stage
constexpr int STAGE_BYTES = (BM + BN) * ROWB;              // 32 KiB

SVLN_DEV int swz(int row, int c) { return (c ^ ((row >> 1) & 7)) << 4; }

template <typename T, int EPI>
__global__ __launch_bounds__(256) void gemm_nt_kernel(GemmArgs p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int EPC = Elt<T>::PER_CHUNK;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wr = wave >> 1, wc = wave & 1;
    const int r32 = lane & 31, h = lane >> 5;

    const int tiles_n = (p.N + BN - 1) / BN;
    const int tiles_m = (p.M + BM - 1) / BM;
    const int nwg = tiles_m * tiles_n;
    // XCD-aware bijective remap (blocks b, b+8, ... share an XCD): give each XCD a contiguous band
    int bid = blockIdx.x;
    {
        const int xcd = bid & 7, q = nwg >> 3, r = nwg & 7;
        bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (bid >> 3);
    }
    // walk tiles column-major inside the band so consecutive blocks share the W panel
    const int bn = bid / tiles_m, bm = bid % tiles_m;
    const int row0 = bm * BM, col0 = bn * BN;
    const int kchunks = p.K / EPC;
    const int nkt = (kchunks + 7) >> 3;

    const T* A = (const T*)p.A;
    const T* W = (const T*)p.W;

    uint4 ra[4], rw[4];
    auto load_stage = [&](int kt) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int q = tid + 256 * i, r = q >> 3, kc = kt * 8 + (q & 7);
            const int gr = row0 + r, gc = col0 + r;
            ra[i] = (gr < p.M && kc < kchunks) ? *(const uint4*)(A + (size_t)gr * p.lda + (size_t)kc * EPC) : zero_chunk();
            rw[i] = (gc < p.N && kc < kchunks) ? *(const uint4*)(W + (size_t)gc * p.ldw + (size_t)kc * EPC) : zero_chunk();
        }
    };
    auto store_stage = [&](int buf) {
        char* sa = smem + buf * STAGE_BYTES;
        char* sw = sa + BM * ROWB;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int q = tid + 256 * i, r = q >> 3, c = q & 7;
            const int off = r * ROWB + swz(r, c);
            *(uint4*)(sa + off) = ra[i];
            *(uint4*)(sw + off) = rw[i];
        }
    };

    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.0f;

    load_stage(0);
    store_stage(0);
    __syncthreads();
    for (int kt = 0; kt < nkt; ++kt) {
        const bool more = kt + 1 < nkt;
        if (more) load_stage(kt + 1);
        const char* sa = smem + (kt & 1) * STAGE_BYTES;
        const char* sw = sa + BM * ROWB;
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            uint4 a[2], b[2];
            const int c = 2 * s + h;
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const int r = wr * 64 + i * 32 + r32;
                a[i] = *(const uint4*)(sa + r * ROWB + swz(r, c));
            }
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const int r = wc * 64 + j * 32 + r32;
                b[j] = *(const uint4*)(sw + r * ROWB + swz(r, c));
            }
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j) mma_chunk<T>(a[i], b[j], acc[i][j]);
        }
        if (more) store_stage((kt + 1) & 1);
        __syncthreads();
    }

    // epilogue: D[row = A row][col = W row]; lanes 0..31 hold 32 consecutive columns
    T* C = (T*)p.C;
    const T* bias = (const T*)p.bias;
    const T* res = (const T*)p.res;
    if (EPI == EPI_SWIGLU) {
        const int n_out = ((col0 + wc * 64) >> 1) + r32;
        const bool ok_n = (col0 + wc * 64 + 32 + r32) < p.N;
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int m = row0 + wr * 64 + i * 32 + acc_row(r, lane);
                if (m < p.M && ok_n) C[(size_t)m * p.ldc + n_out] = from_f32<T>(silu_f(acc[i][0][r]) * acc[i][1][r]);
            }
        return;
    }
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int n = col0 + wc * 64 + j * 32 + r32;
            if (n >= p.N) continue;
            const float bv = bias ? to_f32(bias[n]) : 0.0f;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int m = row0 + wr * 64 + i * 32 + acc_row(r, lane);
                if (m >= p.M) continue;
                float v = acc[i][j][r] + bv;
                if (EPI == EPI_GELU_TANH) v = gelu_tanh_f(v);
                if (EPI == EPI_GELU_ERF) v = gelu_erf_f(v);
                if (res) {
                    const int rr = p.res_mod > 0 ? m % p.res_mod : m;
                    v += to_f32(res[(size_t)rr * p.ldr + n]);
                }
                C[(size_t)m * p.ldc + n] = from_f32<T>(v);
            }
        }
}

}  // namespace

template <typename T> void launch_gemm(hipStream_t s, const GemmArgs& a) {
    const int tiles = ((a.M + BM - 1) / BM) * ((a.N + BN - 1) / BN);
    if (tiles <= 0) return;
    const size_t lds = 2 * STAGE_BYTES;
    dim3 grid(tiles), block(256);
    switch (a.epi) {
        case EPI_NONE: hipLaunchKernelGGL((gemm_nt_kernel<T, EPI_NONE>), grid, block, lds, s, a); break;
        case EPI_GELU_TANH: hipLaunchKernelGGL((gemm_nt_kernel<T, EPI_GELU_TANH>), grid, block, lds, s, a); break;
        case EPI_GELU_ERF: hipLaunchKernelGGL((gemm_nt_kernel<T, EPI_GELU_ERF>), grid, block, lds, s, a); break;
        case EPI_SWIGLU: hipLaunchKernelGGL((gemm_nt_kernel<T, EPI_SWIGLU>), grid, block, lds, s, a); break;
        default: break;
    }
}
template <typename T, int EPI> static void gemm_attr() {
    (void)hipFuncSetAttribute((const void*)gemm_nt_kernel<T, EPI>, hipFuncAttributeMaxDynamicSharedMemorySize, 2 * STAGE_BYTES);
}
void gemm_init_attrs() {
    gemm_attr<bf16, EPI_NONE>(); gemm_attr<bf16, EPI_GELU_TANH>(); gemm_attr<bf16, EPI_GELU_ERF>(); gemm_attr<bf16, EPI_SWIGLU>();
    gemm_attr<float, EPI_NONE>(); gemm_attr<float, EPI_GELU_TANH>(); gemm_attr<float, EPI_GELU_ERF>(); gemm_attr<float, EPI_SWIGLU>();
}
template void launch_gemm<bf16>(hipStream_t, const GemmArgs&);
template void launch_gemm<float>(hipStream_t, const GemmArgs&);

}  // namespace svln
