// Decode-path GEMV  y[N] = epi(W[N,K] . x'[K] + bias) + res  -- the HBM-bound weight stream that
// dominates a batch-1 action-token decode (14.1 GB of bf16 weights per token).
//
// Structure (gfx950): 256-thread workgroups; the activation vector is staged ONCE per workgroup
// into LDS as fp32 (optionally through a fused RMSNorm: sum of squares, rsqrt, gain), then every
// wave streams whole weight rows straight HBM -> VGPR with 16-byte loads (lane i takes chunks
// i, i+64, ... of the row: each wave instruction reads 1 KiB contiguous), R rows per wave in
// flight for memory-level parallelism, fp32 FMA accumulate, wave-shuffle reduction, fused epilogue
// (bias / residual / SwiGLU / arg-max).  No LDS round trip for weights (each byte is used once).
//
// Roofline: HBM.  Algorithmic bytes per launch = N*K*sizeof(T) (+ x, y: negligible).
#include "common.h"
#include "kernels.h"

namespace svln {

namespace {

constexpr int GEMV_THREADS = 256;
constexpr int GEMV_WAVES = GEMV_THREADS / 64;

// x in LDS, split in 16-byte planes so that consecutive lanes read consecutive 16 B (conflict-free):
// floats of chunk ci, part p (4 floats each) live at xs[p * nch * 4 + ci * 4 ...].
template <typename T>
SVLN_DEV void stage_x(float* xs, const GemvArgs& p, int nch) {
    constexpr int EPC = Elt<T>::PER_CHUNK;
    constexpr int PARTS = EPC / 4;
    const T* x = (const T*)p.x;
    const int tid = threadIdx.x;
    float scale = 1.0f;
    if (p.norm_w) {
        __shared__ float red[GEMV_WAVES];
        float ss = 0.0f;
        for (int ci = tid; ci < nch; ci += GEMV_THREADS) {
            float f[EPC];
            chunk_to_f32<T>(*(const uint4*)(x + (size_t)ci * EPC), f);
#pragma unroll
            for (int e = 0; e < EPC; ++e) ss += f[e] * f[e];
        }
        ss = wave_sum(ss);
        if ((tid & 63) == 0) red[tid >> 6] = ss;
        __syncthreads();
        float tot = 0.0f;
#pragma unroll
        for (int w = 0; w < GEMV_WAVES; ++w) tot += red[w];
        scale = rsqrtf(tot / (float)p.K + p.eps);
    }
    const T* g = (const T*)p.norm_w;
    for (int ci = tid; ci < nch; ci += GEMV_THREADS) {
        float f[EPC];
        chunk_to_f32<T>(*(const uint4*)(x + (size_t)ci * EPC), f);
        if (g) {
            float gf[EPC];
            chunk_to_f32<T>(*(const uint4*)(g + (size_t)ci * EPC), gf);
#pragma unroll
            for (int e = 0; e < EPC; ++e) f[e] = gf[e] * (f[e] * scale);       // Qwen2RMSNorm: weight * (x * rsqrt)
        }
#pragma unroll
        for (int q = 0; q < PARTS; ++q)
            *(float4*)(xs + (size_t)q * nch * 4 + (size_t)ci * 4) = make_float4(f[4 * q], f[4 * q + 1], f[4 * q + 2], f[4 * q + 3]);
    }
    __syncthreads();
}

template <typename T>
SVLN_DEV void load_x(const float* xs, int nch, int ci, float* f) {
    constexpr int PARTS = Elt<T>::PER_CHUNK / 4;
#pragma unroll
    for (int q = 0; q < PARTS; ++q) {
        const float4 v = *(const float4*)(xs + (size_t)q * nch * 4 + (size_t)ci * 4);
        f[4 * q] = v.x; f[4 * q + 1] = v.y; f[4 * q + 2] = v.z; f[4 * q + 3] = v.w;
    }
}

// R dot products of consecutive "logical" rows against x (rows given by pointer)
template <typename T, int R>
SVLN_DEV void dot_rows(const T* const (&rows)[R], const float* xs, int nch, int lane, float (&acc)[R]) {
    constexpr int EPC = Elt<T>::PER_CHUNK;
#pragma unroll
    for (int r = 0; r < R; ++r) acc[r] = 0.0f;
    int ci = lane;
    for (; ci + 64 < nch; ci += 128) {                 // two chunks per row in flight: 2R x 1 KiB per wave
        uint4 w0[R], w1[R];
#pragma unroll
        for (int r = 0; r < R; ++r) {
            w0[r] = load_nt(rows[r] + (size_t)ci * EPC);
            w1[r] = load_nt(rows[r] + (size_t)(ci + 64) * EPC);
        }
        float x0[EPC], x1[EPC];
        load_x<T>(xs, nch, ci, x0);
        load_x<T>(xs, nch, ci + 64, x1);
#pragma unroll
        for (int r = 0; r < R; ++r) {
            float f[EPC];
            chunk_to_f32<T>(w0[r], f);
#pragma unroll
            for (int e = 0; e < EPC; ++e) acc[r] = fmaf(f[e], x0[e], acc[r]);
            chunk_to_f32<T>(w1[r], f);
#pragma unroll
            for (int e = 0; e < EPC; ++e) acc[r] = fmaf(f[e], x1[e], acc[r]);
        }
    }
    for (; ci < nch; ci += 64) {
        float x0[EPC];
        load_x<T>(xs, nch, ci, x0);
#pragma unroll
        for (int r = 0; r < R; ++r) {
            float f[EPC];
            chunk_to_f32<T>(load_nt(rows[r] + (size_t)ci * EPC), f);
#pragma unroll
            for (int e = 0; e < EPC; ++e) acc[r] = fmaf(f[e], x0[e], acc[r]);
        }
    }
#pragma unroll
    for (int r = 0; r < R; ++r) acc[r] = wave_sum(acc[r]);
}

template <typename T, int EPI>
__global__ __launch_bounds__(GEMV_THREADS) void gemv_kernel(GemvArgs p) {
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    float* xs = (float*)smem_raw;
    constexpr int EPC = Elt<T>::PER_CHUNK;
    constexpr int R = 4;
    const int nch = p.K / EPC;
    stage_x<T>(xs, p, nch);

    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int gw = blockIdx.x * GEMV_WAVES + wave, nw = gridDim.x * GEMV_WAVES;
    const T* W = (const T*)p.W;

    if (EPI == EPI_SWIGLU) {
        // packed rows: 64-row blocks = [32 gate | 32 up]; one group = 2 outputs (2 gate + 2 up rows)
        const int n_out = p.N >> 1;
        T* y = (T*)p.y;
        for (int j0 = gw * 2; j0 < n_out; j0 += nw * 2) {
            const T* rows[R];
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                const int j = min(j0 + u, n_out - 1);
                const size_t gr = (size_t)(j >> 5) * 64 + (j & 31);
                rows[2 * u] = W + gr * p.ldw;
                rows[2 * u + 1] = W + (gr + 32) * p.ldw;
            }
            float acc[R];
            dot_rows<T, R>(rows, xs, nch, lane, acc);
            if (lane < 2 && j0 + lane < n_out) {
                const float gt = lane == 0 ? acc[0] : acc[2], up = lane == 0 ? acc[1] : acc[3];
                y[j0 + lane] = from_f32<T>(silu_f(gt) * up);
            }
        }
        return;
    }

    float best = -INFINITY;
    int best_i = 0x7FFFFFFF;
    for (int n0 = gw * R; n0 < p.N; n0 += nw * R) {
        const T* rows[R];
#pragma unroll
        for (int r = 0; r < R; ++r) rows[r] = W + (size_t)min(n0 + r, p.N - 1) * p.ldw;
        float acc[R];
        dot_rows<T, R>(rows, xs, nch, lane, acc);
        if (EPI == EPI_ARGMAX) {
#pragma unroll
            for (int r = 0; r < R; ++r)
                if (n0 + r < p.N && acc[r] > best) { best = acc[r]; best_i = n0 + r; }     // rows ascend: first max wins
        } else if (lane < R && n0 + lane < p.N) {
            const int n = n0 + lane;
            float v = lane == 0 ? acc[0] : lane == 1 ? acc[1] : lane == 2 ? acc[2] : acc[3];
            if (p.bias) v += to_f32(((const T*)p.bias)[n]);
            if (p.res) v += to_f32(((const T*)p.res)[n]);
            ((T*)p.y)[n] = from_f32<T>(v);
        }
    }
    if (EPI == EPI_ARGMAX) {
        __shared__ float bv[GEMV_WAVES];
        __shared__ int bi[GEMV_WAVES];
        if (lane == 0) { bv[wave] = best; bi[wave] = best_i; }
        __syncthreads();
        if (threadIdx.x == 0) {
            float v = bv[0]; int i = bi[0];
#pragma unroll
            for (int w = 1; w < GEMV_WAVES; ++w)
                if (bv[w] > v || (bv[w] == v && bi[w] < i)) { v = bv[w]; i = bi[w]; }
            p.part_val[blockIdx.x] = v;
            p.part_idx[blockIdx.x] = i;
        }
    }
}

// final arg-max over per-workgroup partials: greatest value, lowest index on ties (torch.argmax on CPU)
__global__ __launch_bounds__(256) void argmax_final_kernel(const float* pv, const int* pi, int n, int* out_token, float* out_top) {
    __shared__ float sv[256];
    __shared__ int si[256];
    __shared__ float s2[256];
    float v = -INFINITY, v2 = -INFINITY;
    int i = 0x7FFFFFFF;
    for (int k = threadIdx.x; k < n; k += 256) {
        const float c = pv[k];
        const int ci = pi[k];
        if (c > v || (c == v && ci < i)) { v2 = v; v = c; i = ci; } else if (c > v2) v2 = c;
    }
    sv[threadIdx.x] = v; si[threadIdx.x] = i; s2[threadIdx.x] = v2;
    __syncthreads();
    for (int s = 128; s > 0; s >>= 1) {
        if (threadIdx.x < s) {
            const float a = sv[threadIdx.x], b = sv[threadIdx.x + s];
            const int ai = si[threadIdx.x], bi = si[threadIdx.x + s];
            const float a2 = s2[threadIdx.x], b2 = s2[threadIdx.x + s];
            if (b > a || (b == a && bi < ai)) { sv[threadIdx.x] = b; si[threadIdx.x] = bi; s2[threadIdx.x] = fmaxf(a, b2); }
            else s2[threadIdx.x] = fmaxf(a2, b);
        }
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        *out_token = si[0];
        if (out_top) { out_top[0] = sv[0]; out_top[1] = s2[0]; }
    }
}

}  // namespace

int gemv_grid(int N) {
    // 4 rows per wave-iteration, 4 waves per workgroup; ~4 workgroups per CU resident, then grid-stride
    const int groups = (N + 15) / 16;
    return groups < 1024 ? (groups < 1 ? 1 : groups) : 1024;
}

template <typename T> void launch_gemv(hipStream_t s, const GemvArgs& a) {
    const int grid = gemv_grid(a.epi == EPI_SWIGLU ? a.N / 2 * 2 : a.N);
    const size_t lds = (size_t)a.K * sizeof(float);
    dim3 g(grid), b(GEMV_THREADS);
    switch (a.epi) {
        case EPI_NONE: hipLaunchKernelGGL((gemv_kernel<T, EPI_NONE>), g, b, lds, s, a); break;
        case EPI_SWIGLU: hipLaunchKernelGGL((gemv_kernel<T, EPI_SWIGLU>), g, b, lds, s, a); break;
        case EPI_ARGMAX: hipLaunchKernelGGL((gemv_kernel<T, EPI_ARGMAX>), g, b, lds, s, a); break;
        default: break;
    }
}
template <typename T, int EPI> static void gemv_attr() {
    (void)hipFuncSetAttribute((const void*)gemv_kernel<T, EPI>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 256);
}
void gemv_init_attrs() {
    gemv_attr<bf16, EPI_NONE>(); gemv_attr<bf16, EPI_SWIGLU>(); gemv_attr<bf16, EPI_ARGMAX>();
    gemv_attr<float, EPI_NONE>(); gemv_attr<float, EPI_SWIGLU>(); gemv_attr<float, EPI_ARGMAX>();
}
template void launch_gemv<bf16>(hipStream_t, const GemvArgs&);
template void launch_gemv<float>(hipStream_t, const GemvArgs&);

void launch_argmax_final(hipStream_t s, const float* pv, const int* pi, int n, int* out_token, float* out_top) {
    hipLaunchKernelGGL(argmax_final_kernel, dim3(1), dim3(256), 0, s, pv, pi, n, out_token, out_top);
}

}  // namespace svln
