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
#include <hip/hip_ext.h>

#include <cstdlib>

#include "common.h"
#include "kernels.h"

namespace svln {

namespace {

typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2_t;
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

// R dot products against x over the chunks ci = c0 + lane + 64*k*cstep (k = 0, 1, ...) below nch.
// XLDS: x comes from the workgroup's LDS copy (fp32, possibly RMS-normalised); otherwise each lane reads the
// x chunk it needs straight from global memory (L2-resident, 7-37 KB) next to its weight chunks.
template <typename T, int R, bool XLDS>
SVLN_DEV void dot_accum(const T* const (&rows)[R], const float* xs, const T* xg, int nch, int c0, int cstep, int lane, float (&acc)[R]) {
    constexpr int EPC = Elt<T>::PER_CHUNK;
    const int stride = 64 * cstep;
    int ci = c0 + lane;
    for (; ci + stride < nch; ci += 2 * stride) {      // two chunks per row in flight: 2R x 1 KiB per wave
        uint4 w0[R], w1[R];
#pragma unroll
        for (int r = 0; r < R; ++r) {
            w0[r] = load_nt(rows[r] + (size_t)ci * EPC);
            w1[r] = load_nt(rows[r] + (size_t)(ci + stride) * EPC);
        }
        float x0[EPC], x1[EPC];
        if (XLDS) {
            load_x<T>(xs, nch, ci, x0);
            load_x<T>(xs, nch, ci + stride, x1);
        } else {
            chunk_to_f32<T>(*(const uint4*)(xg + (size_t)ci * EPC), x0);
            chunk_to_f32<T>(*(const uint4*)(xg + (size_t)(ci + stride) * EPC), x1);
        }
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
    for (; ci < nch; ci += stride) {
        float x0[EPC];
        if (XLDS) load_x<T>(xs, nch, ci, x0);
        else chunk_to_f32<T>(*(const uint4*)(xg + (size_t)ci * EPC), x0);
#pragma unroll
        for (int r = 0; r < R; ++r) {
            float f[EPC];
            chunk_to_f32<T>(load_nt(rows[r] + (size_t)ci * EPC), f);
#pragma unroll
            for (int e = 0; e < EPC; ++e) acc[r] = fmaf(f[e], x0[e], acc[r]);
        }
    }
}
template <typename T, int R, bool XLDS>
SVLN_DEV void dot_rows(const T* const (&rows)[R], const float* xs, const T* xg, int nch, int c0, int cstep, int lane, float (&acc)[R]) {
#pragma unroll
    for (int r = 0; r < R; ++r) acc[r] = 0.0f;
    dot_accum<T, R, XLDS>(rows, xs, xg, nch, c0, cstep, lane, acc);
#pragma unroll
    for (int r = 0; r < R; ++r) acc[r] = wave_sum(acc[r]);
}

// Small-N variant (qkv / o / down projections of a decode step: N <= 8192 rows is only 224-288 workgroups of the
// wave-per-rows kernel, i.e. < 1 per CU).  Here a workgroup owns 4 rows and its 4 waves split K (interleaved
// 1 KiB blocks), so N/4 workgroups exist (3.5-4.5 per CU) and their prologues overlap other workgroups' streams.
template <typename T, bool NORM, int R>
__global__ __launch_bounds__(GEMV_THREADS) void gemv_ksplit_kernel(GemvArgs p) {
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    __shared__ float part[GEMV_WAVES][R + 1];
    constexpr int EPC = Elt<T>::PER_CHUNK;
    const int nch = p.K / EPC;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const T* W = (const T*)p.W;
    const T* xg = (const T*)p.x;
    const T* gg = (const T*)p.norm_w;
    constexpr int STRIDE = 64 * GEMV_WAVES;
    for (int n0 = blockIdx.x * R; n0 < p.N; n0 += gridDim.x * R) {
        const T* rows[R];
#pragma unroll
        for (int r = 0; r < R; ++r) rows[r] = W + (size_t)min(n0 + r, p.N - 1) * p.ldw;
        float acc[R + 1];                      // acc[R] = this wave's share of sum(x^2) when NORM
#pragma unroll
        for (int r = 0; r <= R; ++r) acc[r] = 0.0f;
        if (NORM) {
            // RMSNorm folded into the product, no prologue:  y = rsqrt(mean(x^2) + eps) * sum_i W[n][i] * (g[i] * x[i]);
            // every wave reads its K share of x and g next to its weight chunks (L2-resident, 7 KB each)
            for (int ci = wave * 64 + lane; ci < nch; ci += STRIDE) {
                uint4 w[R];
#pragma unroll
                for (int r = 0; r < R; ++r) w[r] = load_nt(rows[r] + (size_t)ci * EPC);
                float xf[EPC], gf[EPC];
                chunk_to_f32<T>(*(const uint4*)(xg + (size_t)ci * EPC), xf);
                chunk_to_f32<T>(*(const uint4*)(gg + (size_t)ci * EPC), gf);
#pragma unroll
                for (int e = 0; e < EPC; ++e) { acc[R] = fmaf(xf[e], xf[e], acc[R]); xf[e] *= gf[e]; }
#pragma unroll
                for (int r = 0; r < R; ++r) {
                    float f[EPC];
                    chunk_to_f32<T>(w[r], f);
#pragma unroll
                    for (int e = 0; e < EPC; ++e) acc[r] = fmaf(f[e], xf[e], acc[r]);
                }
            }
            acc[R] = wave_sum(acc[R]);
        } else {
            float a4[R];
#pragma unroll
            for (int r = 0; r < R; ++r) a4[r] = 0.0f;
            dot_accum<T, R, false>(rows, nullptr, xg, nch, wave * 64, GEMV_WAVES, lane, a4);
#pragma unroll
            for (int r = 0; r < R; ++r) acc[r] = a4[r];
        }
#pragma unroll
        for (int r = 0; r < R; ++r) acc[r] = wave_sum(acc[r]);
        if (lane == 0) {
#pragma unroll
            for (int r = 0; r <= R; ++r) part[wave][r] = acc[r];
        }
        __syncthreads();
        if (threadIdx.x < R && n0 + threadIdx.x < p.N) {
            const int n = n0 + threadIdx.x;
            float v = part[0][threadIdx.x] + part[1][threadIdx.x] + part[2][threadIdx.x] + part[3][threadIdx.x];
            if (NORM) v *= rsqrtf((part[0][R] + part[1][R] + part[2][R] + part[3][R]) / (float)p.K + p.eps);
            if (p.bias) v += to_f32(((const T*)p.bias)[n]);
            if (p.res) v += to_f32(((const T*)p.res)[n]);
            ((T*)p.y)[n] = from_f32<T>(v);
        }
        __syncthreads();
    }
}

// Batched decode GEMV for B environments decoded in lockstep (SURVEY.md 8f-1 / BASELINE configs[4]):
//   Y[b][n] = epi(W[n,:] . x'_b + bias[n]) + res[b][n],   x'_b = x_b or rmsnorm(x_b) * g  (folded, no prologue)
// Every weight byte is streamed from HBM once for all B activations.  Workgroup = 4 rows (SwiGLU: 2 outputs), its 4 waves
// split K; per chunk position a lane loads 4 weight chunks (non-temporal) + B activation chunks (L2-resident) and does
// 4 * B * 8 FMAs.  Partial sums (and the per-env sum of squares) are reduced across waves through LDS.
template <typename T, int EPI, bool NORM, int B>
__global__ __launch_bounds__(GEMV_THREADS) void gemv_batched_kernel(GemvBatchArgs p) {
    constexpr int EPC = Elt<T>::PER_CHUNK, R = 4, STRIDE = 64 * GEMV_WAVES;       // (R = 8 measured no better at B = 8: dot2-issue bound)
    __shared__ float part[GEMV_WAVES][R * B + B];
    const int nch = p.K / EPC;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, tid = threadIdx.x;
    const T* W = (const T*)p.W;
    const T* xg = (const T*)p.x;
    const T* gg = (const T*)p.norm_w;
    const int n_units = EPI == EPI_SWIGLU ? p.N / R : (p.N + R - 1) / R;      // one unit = R weight rows
    float best = -INFINITY;                                                   // EPI_ARGMAX: thread b < B tracks env b
    int best_i = 0x7FFFFFFF;
    for (int u = blockIdx.x; u < n_units; u += gridDim.x) {
        const T* rows[R];
        int n0;
        if (EPI == EPI_SWIGLU) {          // outputs j0 .. j0+R/2-1: rows (gate j, up j) pairs of the [gate 32 | up 32] packing
            const int j0 = u * (R / 2);
            n0 = j0;
#pragma unroll
            for (int o = 0; o < R / 2; ++o) {
                const size_t gr = (size_t)((j0 + o) >> 5) * 64 + ((j0 + o) & 31);
                rows[2 * o] = W + gr * p.ldw;
                rows[2 * o + 1] = W + (gr + 32) * p.ldw;
            }
        } else {
            n0 = u * R;
#pragma unroll
            for (int r = 0; r < R; ++r) rows[r] = W + (size_t)min(n0 + r, p.N - 1) * p.ldw;
        }
        float acc[R][B], ss[B];
#pragma unroll
        for (int b = 0; b < B; ++b) {
            ss[b] = 0.0f;
#pragma unroll
            for (int r = 0; r < R; ++r) acc[r][b] = 0.0f;
        }
        for (int ci = wave * 64 + lane; ci < nch; ci += STRIDE) {
            uint4 w[R], xr[B];
#pragma unroll
            for (int r = 0; r < R; ++r) w[r] = load_nt(rows[r] + (size_t)ci * EPC);
#pragma unroll
            for (int b = 0; b < B; ++b) xr[b] = *(const uint4*)(xg + (size_t)b * p.ldx + (size_t)ci * EPC);
            if (sizeof(T) == 2 && !NORM) {
                // bf16: packed dot products straight on the bf16 pairs (v_dot2c_f32_bf16), no conversions
#pragma unroll
                for (int b = 0; b < B; ++b) {
                    const unsigned xw[4] = {xr[b].x, xr[b].y, xr[b].z, xr[b].w};
#pragma unroll
                    for (int r = 0; r < R; ++r) {
                        const unsigned ww[4] = {w[r].x, w[r].y, w[r].z, w[r].w};
#pragma unroll
                        for (int q = 0; q < 4; ++q)
                            acc[r][b] = __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(bf16x2_t, ww[q]), __builtin_bit_cast(bf16x2_t, xw[q]),
                                                                        acc[r][b], false);
                    }
                }
            } else {
                float gf[EPC];
                if (NORM) chunk_to_f32<T>(*(const uint4*)(gg + (size_t)ci * EPC), gf);
                float wf[R][EPC];
#pragma unroll
                for (int r = 0; r < R; ++r) chunk_to_f32<T>(w[r], wf[r]);
#pragma unroll
                for (int b = 0; b < B; ++b) {
                    float xf[EPC];
                    chunk_to_f32<T>(xr[b], xf);
                    if (NORM) {
#pragma unroll
                        for (int e = 0; e < EPC; ++e) { ss[b] = fmaf(xf[e], xf[e], ss[b]); xf[e] *= gf[e]; }
                    }
#pragma unroll
                    for (int r = 0; r < R; ++r)
#pragma unroll
                        for (int e = 0; e < EPC; ++e) acc[r][b] = fmaf(wf[r][e], xf[e], acc[r][b]);
                }
            }
        }
#pragma unroll
        for (int b = 0; b < B; ++b) {
            if (NORM) ss[b] = wave_sum(ss[b]);
#pragma unroll
            for (int r = 0; r < R; ++r) acc[r][b] = wave_sum(acc[r][b]);
        }
        if (lane == 0) {
#pragma unroll
            for (int b = 0; b < B; ++b) {
                part[wave][R * B + b] = ss[b];
#pragma unroll
                for (int r = 0; r < R; ++r) part[wave][r * B + b] = acc[r][b];
            }
        }
        __syncthreads();
        auto total = [&](int k) { return part[0][k] + part[1][k] + part[2][k] + part[3][k]; };
        if (EPI == EPI_SWIGLU) {
            if (tid < (R / 2) * B) {
                const int o = tid / B, b = tid % B;
                const float sc = NORM ? rsqrtf(total(R * B + b) / (float)p.K + p.eps) : 1.0f;
                const float gt = total((2 * o) * B + b) * sc, up = total((2 * o + 1) * B + b) * sc;
                ((T*)p.y)[(size_t)b * p.ldy + n0 + o] = from_f32<T>(silu_f(gt) * up);
            }
        } else if (EPI == EPI_ARGMAX) {
            if (tid < B) {
#pragma unroll
                for (int r = 0; r < R; ++r) {
                    const float v = total(r * B + tid);
                    if (n0 + r < p.N && v > best) { best = v; best_i = n0 + r; }      // units ascend per workgroup: first max wins
                }
            }
        } else if (tid < R * B) {
            const int r = tid / B, b = tid % B, n = n0 + r;
            if (n < p.N) {
                float v = total(r * B + b);
                if (NORM) v *= rsqrtf(total(R * B + b) / (float)p.K + p.eps);
                if (p.bias) v += to_f32(((const T*)p.bias)[n]);
                if (p.res) v += to_f32(((const T*)p.res)[(size_t)b * p.ldr + n]);
                ((T*)p.y)[(size_t)b * p.ldy + n] = from_f32<T>(v);
            }
        }
        __syncthreads();
    }
    if (EPI == EPI_ARGMAX && tid < B) {
        p.part_val[(size_t)tid * gridDim.x + blockIdx.x] = best;
        p.part_idx[(size_t)tid * gridDim.x + blockIdx.x] = best_i;
    }
}

// final arg-max of env b = blockIdx.x over its per-workgroup partials
__global__ __launch_bounds__(256) void argmax_final_batched_kernel(const float* pv, const int* pi, int n, int* out_tokens) {
    __shared__ float sv[256];
    __shared__ int si[256];
    const float* v0 = pv + (size_t)blockIdx.x * n;
    const int* i0 = pi + (size_t)blockIdx.x * n;
    float v = -INFINITY;
    int i = 0x7FFFFFFF;
    for (int k = threadIdx.x; k < n; k += 256) {
        const float c = v0[k];
        const int ci = i0[k];
        if (c > v || (c == v && ci < i)) { v = c; i = ci; }
    }
    sv[threadIdx.x] = v; si[threadIdx.x] = i;
    __syncthreads();
    for (int s2 = 128; s2 > 0; s2 >>= 1) {
        if (threadIdx.x < s2) {
            const float b = sv[threadIdx.x + s2];
            const int bi = si[threadIdx.x + s2];
            if (b > sv[threadIdx.x] || (b == sv[threadIdx.x] && bi < si[threadIdx.x])) { sv[threadIdx.x] = b; si[threadIdx.x] = bi; }
        }
        __syncthreads();
    }
    if (threadIdx.x == 0) out_tokens[blockIdx.x] = si[0];
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
            dot_rows<T, R, true>(rows, xs, nullptr, nch, 0, 1, lane, acc);
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
        dot_rows<T, R, true>(rows, xs, nullptr, nch, 0, 1, lane, acc);
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
    // wave-per-4-rows kernel: 4 row groups per workgroup iteration.  Prefer a grid (<= 1280 workgroups, ~5 per CU)
    // that divides the row groups evenly so no wave runs an extra iteration (9472 groups -> 1184 workgroups x 2).
    const int groups = (N + 3) / 4;
    const int wg_groups = (groups + GEMV_WAVES - 1) / GEMV_WAVES;       // workgroup-iterations needed
    if (wg_groups <= 1280) return wg_groups < 1 ? 1 : wg_groups;
    if (wg_groups > 4 * 1280) return 1024;          // many iterations per wave (lm_head): imbalance is negligible, 1024 measured best
    if (groups % GEMV_WAVES == 0)
        for (int g = 1280; g >= 640; --g)
            if (wg_groups % g == 0) return g;
    return 1024;
}

template <typename T> void launch_gemv(hipStream_t s, const GemvArgs& a) { launch_gemv_timed<T>(s, a, nullptr, nullptr); }

// start/stop (optional) receive the kernel's own begin/end timestamps (hipExtLaunchKernelGGL)
#define SVLN_LAUNCH(kern, grid, block, lds)                                                        \
    do {                                                                                          \
        if (start || stop) hipExtLaunchKernelGGL(kern, grid, block, lds, s, start, stop, 0, a);   \
        else hipLaunchKernelGGL(kern, grid, block, lds, s, a);                                    \
    } while (0)
template <typename T> void launch_gemv_timed(hipStream_t s, const GemvArgs& a, hipEvent_t start, hipEvent_t stop) {
    dim3 b(GEMV_THREADS);
    if (a.epi == EPI_NONE && a.N <= 8192) {
        static const int r_env = getenv("SVLN_GEMV_R") ? atoi(getenv("SVLN_GEMV_R")) : 0;      // tuning experiments
        const int R = r_env ? r_env : 2;
        int grid = (a.N + R - 1) / R;
        if (grid > 2048) grid = 2048;
        if (R == 2) { if (a.norm_w) SVLN_LAUNCH((gemv_ksplit_kernel<T, true, 2>), dim3(grid), b, 0); else SVLN_LAUNCH((gemv_ksplit_kernel<T, false, 2>), dim3(grid), b, 0); }
        else if (R == 8) { if (a.norm_w) SVLN_LAUNCH((gemv_ksplit_kernel<T, true, 8>), dim3(grid), b, 0); else SVLN_LAUNCH((gemv_ksplit_kernel<T, false, 8>), dim3(grid), b, 0); }
        else { if (a.norm_w) SVLN_LAUNCH((gemv_ksplit_kernel<T, true, 4>), dim3(grid), b, 0); else SVLN_LAUNCH((gemv_ksplit_kernel<T, false, 4>), dim3(grid), b, 0); }
        return;
    }
    const int grid = gemv_grid(a.N);
    const size_t lds = (size_t)a.K * sizeof(float);
    dim3 g(grid);
    switch (a.epi) {
        case EPI_NONE: SVLN_LAUNCH((gemv_kernel<T, EPI_NONE>), g, b, lds); break;
        case EPI_SWIGLU: SVLN_LAUNCH((gemv_kernel<T, EPI_SWIGLU>), g, b, lds); break;
        case EPI_ARGMAX: SVLN_LAUNCH((gemv_kernel<T, EPI_ARGMAX>), g, b, lds); break;
        default: break;
    }
}
#undef SVLN_LAUNCH
int gemv_batched_grid(int N, int epi, int B) {
    const int R = 4;
    (void)B;
    const int units = epi == EPI_SWIGLU ? N / R : (N + R - 1) / R;
    return units < 2048 ? (units < 1 ? 1 : units) : 2048;
}
template <typename T, int EPI, bool NORM> static void launch_gb(hipStream_t s, const GemvBatchArgs& a) {
    dim3 g(gemv_batched_grid(a.N, EPI, a.B)), b(GEMV_THREADS);
    switch (a.B) {
        case 1: hipLaunchKernelGGL((gemv_batched_kernel<T, EPI, NORM, 1>), g, b, 0, s, a); break;
        case 2: hipLaunchKernelGGL((gemv_batched_kernel<T, EPI, NORM, 2>), g, b, 0, s, a); break;
        case 4: hipLaunchKernelGGL((gemv_batched_kernel<T, EPI, NORM, 4>), g, b, 0, s, a); break;
        case 8: hipLaunchKernelGGL((gemv_batched_kernel<T, EPI, NORM, 8>), g, b, 0, s, a); break;
        default: break;
    }
}
template <typename T> void launch_gemv_batched(hipStream_t s, const GemvBatchArgs& a) {
    const bool norm = a.norm_w != nullptr;
    switch (a.epi) {
        case EPI_NONE: if (norm) launch_gb<T, EPI_NONE, true>(s, a); else launch_gb<T, EPI_NONE, false>(s, a); break;
        case EPI_SWIGLU: if (norm) launch_gb<T, EPI_SWIGLU, true>(s, a); else launch_gb<T, EPI_SWIGLU, false>(s, a); break;
        case EPI_ARGMAX: if (norm) launch_gb<T, EPI_ARGMAX, true>(s, a); else launch_gb<T, EPI_ARGMAX, false>(s, a); break;
        default: break;
    }
}
template void launch_gemv_batched<bf16>(hipStream_t, const GemvBatchArgs&);
template void launch_gemv_batched<float>(hipStream_t, const GemvBatchArgs&);
void launch_argmax_final_batched(hipStream_t s, const float* pv, const int* pi, int n, int B, int* out_tokens) {
    hipLaunchKernelGGL(argmax_final_batched_kernel, dim3(B), dim3(256), 0, s, pv, pi, n, out_tokens);
}
template void launch_gemv_timed<bf16>(hipStream_t, const GemvArgs&, hipEvent_t, hipEvent_t);
template void launch_gemv_timed<float>(hipStream_t, const GemvArgs&, hipEvent_t, hipEvent_t);
template <typename T, int EPI> static void gemv_attr() {
    (void)hipFuncSetAttribute((const void*)gemv_kernel<T, EPI>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 256);
}
template <typename T> static void gemv_ksplit_attr() {}
void gemv_init_attrs() {
    gemv_ksplit_attr<bf16>(); gemv_ksplit_attr<float>();
    gemv_attr<bf16, EPI_NONE>(); gemv_attr<bf16, EPI_SWIGLU>(); gemv_attr<bf16, EPI_ARGMAX>();
    gemv_attr<float, EPI_NONE>(); gemv_attr<float, EPI_SWIGLU>(); gemv_attr<float, EPI_ARGMAX>();
}
template void launch_gemv<bf16>(hipStream_t, const GemvArgs&);
template void launch_gemv<float>(hipStream_t, const GemvArgs&);

void launch_argmax_final(hipStream_t s, const float* pv, const int* pi, int n, int* out_token, float* out_top) {
    hipLaunchKernelGGL(argmax_final_kernel, dim3(1), dim3(256), 0, s, pv, pi, n, out_token, out_top);
}

}  // namespace svln
