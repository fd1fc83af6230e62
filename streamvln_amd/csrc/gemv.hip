// Decode-path GEMV  y[N] = epi(W[N,K] . x'[K] + bias) + res  -- the HBM-bound weight stream that
// dominates a batch-1 action-token decode (14.1 GB of bf16 weights per token).
//
// Structure (gfx950): 256-thread workgroups; the activation vector is staged ONCE per workgroup
// into LDS as fp32 (with a fused RMSNorm the copy holds g * x and rsqrt(mean x^2 + eps) is applied
// once per output in the epilogue: one pass over x, one barrier), then every
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
// With a fused RMSNorm the LDS copy holds g * x (one pass over x, one barrier) and the function returns rsqrt(mean(x^2) + eps):
// y = rstd * (W . (g * x)) -- the scale is applied once per output in the epilogue (same folding as gemv_ksplit_kernel).
template <typename T>
SVLN_DEV float stage_x(float* xs, const GemvArgs& p, int nch) {
    constexpr int EPC = Elt<T>::PER_CHUNK;
    constexpr int PARTS = EPC / 4;
    __shared__ float red[GEMV_WAVES];
    const T* x = (const T*)p.x;
    const T* g = (const T*)p.norm_w;
    const int tid = threadIdx.x;
    float ss = 0.0f;
    for (int ci = tid; ci < nch; ci += GEMV_THREADS) {
        float f[EPC];
        chunk_to_f32<T>(*(const uint4*)(x + (size_t)ci * EPC), f);
        if (g) {
            float gf[EPC];
            chunk_to_f32<T>(*(const uint4*)(g + (size_t)ci * EPC), gf);
#pragma unroll
            for (int e = 0; e < EPC; ++e) { ss = fmaf(f[e], f[e], ss); f[e] *= gf[e]; }
        }
#pragma unroll
        for (int q = 0; q < PARTS; ++q)
            *(float4*)(xs + (size_t)q * nch * 4 + (size_t)ci * 4) = make_float4(f[4 * q], f[4 * q + 1], f[4 * q + 2], f[4 * q + 3]);
    }
    if (g) {
        ss = wave_sum(ss);
        if ((tid & 63) == 0) red[tid >> 6] = ss;
    }
    __syncthreads();
    if (!g) return 1.0f;
    float tot = 0.0f;
#pragma unroll
    for (int w = 0; w < GEMV_WAVES; ++w) tot += red[w];
    return rsqrtf(tot / (float)p.K + p.eps);
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
    if (p.skip && *p.skip) return;
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
                const uint8_t* fl = p.pen_flags ? p.pen_flags + (size_t)p.pen_rows[tid] * p.N : nullptr;
#pragma unroll
                for (int r = 0; r < R; ++r) {
                    float v = total(r * B + tid);
                    if (fl && n0 + r < p.N && fl[n0 + r]) v = v < 0.0f ? v * p.pen : v / p.pen;
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
    if (threadIdx.x == 0) out_tokens[blockIdx.x] = si[0] == 0x7FFFFFFF ? -1 : si[0];    // no finite logit: -1 (in-range for the next gather, an error on the host)
}

template <typename T, int EPI>
__global__ __launch_bounds__(GEMV_THREADS) void gemv_kernel(GemvArgs p) {
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    float* xs = (float*)smem_raw;
    const int skip = p.skip ? *p.skip : 0;        // checked after the activation staging, so the flag's load latency hides behind it
    constexpr int EPC = Elt<T>::PER_CHUNK;
    constexpr int R = 4;
    const int nch = p.K / EPC;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int gw = blockIdx.x * GEMV_WAVES + wave, nw = gridDim.x * GEMV_WAVES;
    const T* W = (const T*)p.W;

    if (EPI == EPI_SWIGLU) {
        // packed rows: 64-row blocks = [32 gate | 32 up]; one group = 2 outputs (2 gate + 2 up rows)
        const int n_out = p.N >> 1;
        T* y = (T*)p.y;
        auto group_rows = [&](int j0, const T* (&rows)[R]) {
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                const int j = min(j0 + u, n_out - 1);
                const size_t gr = (size_t)(j >> 5) * 64 + (j & 31);
                rows[2 * u] = W + gr * p.ldw;
                rows[2 * u + 1] = W + (gr + 32) * p.ldw;
            }
        };
        const float xscale = stage_x<T>(xs, p, nch);
        if (skip) return;
        auto finish = [&](int j0, const float (&acc)[R]) {
            if (lane < 2 && j0 + lane < n_out) {
                const float gt = (lane == 0 ? acc[0] : acc[2]) * xscale, up = (lane == 0 ? acc[1] : acc[3]) * xscale;
                y[j0 + lane] = from_f32<T>(silu_f(gt) * up);
            }
        };
        for (int j0 = gw * 2; j0 < n_out; j0 += nw * 2) {
            const T* rows[R];
            group_rows(j0, rows);
            float acc[R];
            dot_rows<T, R, true>(rows, xs, nullptr, nch, 0, 1, lane, acc);
            finish(j0, acc);
        }
        return;
    }
    const float xscale = stage_x<T>(xs, p, nch);
    if (skip) return;

    float best = -INFINITY;
    int best_i = 0x7FFFFFFF;
    for (int n0 = gw * R; n0 < p.N; n0 += nw * R) {
        const T* rows[R];
#pragma unroll
        for (int r = 0; r < R; ++r) rows[r] = W + (size_t)min(n0 + r, p.N - 1) * p.ldw;
        float acc[R];
        dot_rows<T, R, true>(rows, xs, nullptr, nch, 0, 1, lane, acc);
#pragma unroll
        for (int r = 0; r < R; ++r) acc[r] *= xscale;
        if (EPI == EPI_ARGMAX) {
            if (p.pen_flags) {
#pragma unroll
                for (int r = 0; r < R; ++r)
                    if (n0 + r < p.N && p.pen_flags[n0 + r]) acc[r] = acc[r] < 0.0f ? acc[r] * p.pen : acc[r] / p.pen;
            }
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


// ------------------------------------------------------------------------------------------------ fp8 weight-only variants
// Opt-in decode mode: the weights are stored as OCP e4m3 bytes with one fp32 scale per output row, the
// activations stay bf16/fp32.  Same structure as the bf16 kernels above with 16 weights per 16-byte chunk
// (v_cvt_pk_f32_fp8: 2 weights per instruction), the row scale applied once after the wave reduction.  Halves the HBM bytes of
// a decode step; VALU work per byte doubles but stays far below the issue limit.
typedef float f32x2 __attribute__((ext_vector_type(2)));
SVLN_DEV void fp8x16_to_f32(const uint4& w, float* f) {
    const unsigned d[4] = {w.x, w.y, w.z, w.w};
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const auto lo = __builtin_amdgcn_cvt_pk_f32_fp8((int)d[q], false);
        const auto hi = __builtin_amdgcn_cvt_pk_f32_fp8((int)d[q], true);
        f[4 * q] = lo[0]; f[4 * q + 1] = lo[1]; f[4 * q + 2] = hi[0]; f[4 * q + 3] = hi[1];
    }
}
// acc (two partial sums) += 16 e4m3 weights . 16 activations, as packed fp32 FMAs (v_pk_fma_f32: the conversion delivers pairs)
SVLN_DEV void fp8x16_dot(const uint4& w, const f32x2* x2, f32x2& acc) {
    const unsigned d[4] = {w.x, w.y, w.z, w.w};
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const f32x2 lo = __builtin_amdgcn_cvt_pk_f32_fp8((int)d[q], false);
        const f32x2 hi = __builtin_amdgcn_cvt_pk_f32_fp8((int)d[q], true);
        acc = __builtin_elementwise_fma(lo, x2[2 * q], acc);
        acc = __builtin_elementwise_fma(hi, x2[2 * q + 1], acc);
    }
}
// x (bf16 in global memory) -> LDS fp32 in four 16-byte planes per 16-element chunk: element 16*cj + 4*p + e lives at
// xs[p * nch8 * 4 + cj * 4 + e], so consecutive lanes read consecutive 16 B in every plane.
SVLN_DEV float stage_x8(float* xs, const GemvArgs& p, int nch8) {
    __shared__ float red8[GEMV_WAVES];
    const bf16* x = (const bf16*)p.x;
    const bf16* g = (const bf16*)p.norm_w;
    const int tid = threadIdx.x, nch = p.K / 8;
    float ss = 0.0f;
    for (int ci = tid; ci < nch; ci += GEMV_THREADS) {
        float f[8];
        chunk_to_f32<bf16>(*(const uint4*)(x + (size_t)ci * 8), f);
        if (g) {
            float gf[8];
            chunk_to_f32<bf16>(*(const uint4*)(g + (size_t)ci * 8), gf);
#pragma unroll
            for (int e = 0; e < 8; ++e) { ss = fmaf(f[e], f[e], ss); f[e] *= gf[e]; }
        }
        const int cj = ci >> 1, p0 = (ci & 1) * 2;
        *(float4*)(xs + (size_t)p0 * nch8 * 4 + (size_t)cj * 4) = make_float4(f[0], f[1], f[2], f[3]);
        *(float4*)(xs + (size_t)(p0 + 1) * nch8 * 4 + (size_t)cj * 4) = make_float4(f[4], f[5], f[6], f[7]);
    }
    if (g) {
        ss = wave_sum(ss);
        if ((tid & 63) == 0) red8[tid >> 6] = ss;
    }
    __syncthreads();
    if (!g) return 1.0f;
    float tot = 0.0f;
#pragma unroll
    for (int w = 0; w < GEMV_WAVES; ++w) tot += red8[w];
    return rsqrtf(tot / (float)p.K + p.eps);     // applied in the epilogue: y = rstd * scale[n] * (Wq . (g * x))
}
SVLN_DEV void load_x8(const float* xs, int nch8, int cj, f32x2* x2) {
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const float4 v = *(const float4*)(xs + (size_t)q * nch8 * 4 + (size_t)cj * 4);
        x2[2 * q] = f32x2{v.x, v.y};
        x2[2 * q + 1] = f32x2{v.z, v.w};
    }
}
// R dot products of fp8 rows against the LDS copy of x; two chunks per row in flight
template <int R>
SVLN_DEV void dot8_rows(const uint8_t* const (&rows)[R], const float* xs, int nch8, int lane, float (&acc)[R]) {
    f32x2 a2[R];
#pragma unroll
    for (int r = 0; r < R; ++r) a2[r] = f32x2{0.0f, 0.0f};
    int ci = lane;
    for (; ci + 64 < nch8; ci += 128) {
        uint4 w0[R], w1[R];
#pragma unroll
        for (int r = 0; r < R; ++r) {
            w0[r] = load_nt(rows[r] + (size_t)ci * 16);
            w1[r] = load_nt(rows[r] + (size_t)(ci + 64) * 16);
        }
        f32x2 x0[8], x1[8];
        load_x8(xs, nch8, ci, x0);
        load_x8(xs, nch8, ci + 64, x1);
#pragma unroll
        for (int r = 0; r < R; ++r) {
            fp8x16_dot(w0[r], x0, a2[r]);
            fp8x16_dot(w1[r], x1, a2[r]);
        }
    }
    for (; ci < nch8; ci += 64) {
        f32x2 x0[8];
        load_x8(xs, nch8, ci, x0);
#pragma unroll
        for (int r = 0; r < R; ++r) fp8x16_dot(load_nt(rows[r] + (size_t)ci * 16), x0, a2[r]);
    }
#pragma unroll
    for (int r = 0; r < R; ++r) acc[r] = wave_sum(a2[r][0] + a2[r][1]);
}

template <int EPI>
__global__ __launch_bounds__(GEMV_THREADS) void gemv8_kernel(GemvArgs p) {
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    float* xs = (float*)smem_raw;
    const int skip = p.skip ? *p.skip : 0;
    constexpr int R = 4;
    const int nch8 = p.K / 16;
    const float xscale = stage_x8(xs, p, nch8);
    if (skip) return;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int gw = blockIdx.x * GEMV_WAVES + wave, nw = gridDim.x * GEMV_WAVES;
    const uint8_t* W = (const uint8_t*)p.w8;
    if (EPI == EPI_SWIGLU) {
        const int n_out = p.N >> 1;
        bf16* y = (bf16*)p.y;
        for (int j0 = gw * 2; j0 < n_out; j0 += nw * 2) {
            const uint8_t* rows[R];
            size_t gr[2];
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                const int j = min(j0 + u, n_out - 1);
                gr[u] = (size_t)(j >> 5) * 64 + (j & 31);
                rows[2 * u] = W + gr[u] * p.ldw;
                rows[2 * u + 1] = W + (gr[u] + 32) * p.ldw;
            }
            float acc[R];
            dot8_rows<R>(rows, xs, nch8, lane, acc);
            if (lane < 2 && j0 + lane < n_out) {
                const size_t g0 = gr[lane];
                const float gt = (lane == 0 ? acc[0] : acc[2]) * (p.scale[g0] * xscale), up = (lane == 0 ? acc[1] : acc[3]) * (p.scale[g0 + 32] * xscale);
                y[j0 + lane] = from_f32<bf16>(silu_f(gt) * up);
            }
        }
        return;
    }
    float best = -INFINITY;
    int best_i = 0x7FFFFFFF;
    for (int n0 = gw * R; n0 < p.N; n0 += nw * R) {
        const uint8_t* rows[R];
#pragma unroll
        for (int r = 0; r < R; ++r) rows[r] = W + (size_t)min(n0 + r, p.N - 1) * p.ldw;
        float acc[R];
        dot8_rows<R>(rows, xs, nch8, lane, acc);
#pragma unroll
        for (int r = 0; r < R; ++r) acc[r] *= p.scale[min(n0 + r, p.N - 1)] * xscale;
        if (EPI == EPI_ARGMAX) {
            if (p.pen_flags) {
#pragma unroll
                for (int r = 0; r < R; ++r)
                    if (n0 + r < p.N && p.pen_flags[n0 + r]) acc[r] = acc[r] < 0.0f ? acc[r] * p.pen : acc[r] / p.pen;
            }
#pragma unroll
            for (int r = 0; r < R; ++r)
                if (n0 + r < p.N && acc[r] > best) { best = acc[r]; best_i = n0 + r; }
        } else if (lane < R && n0 + lane < p.N) {
            const int n = n0 + lane;
            float v = lane == 0 ? acc[0] : lane == 1 ? acc[1] : lane == 2 ? acc[2] : acc[3];
            if (p.bias) v += to_f32(((const bf16*)p.bias)[n]);
            if (p.res) v += to_f32(((const bf16*)p.res)[n]);
            ((bf16*)p.y)[n] = from_f32<bf16>(v);
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

// small-N fp8 variant: workgroup = R rows, its 4 waves split K; RMSNorm folded in without a prologue (as gemv_ksplit_kernel)
template <bool NORM, int R>
__global__ __launch_bounds__(GEMV_THREADS) void gemv8_ksplit_kernel(GemvArgs p) {
    __shared__ float part[GEMV_WAVES][R + 1];
    if (p.skip && *p.skip) return;
    const int nch8 = p.K / 16;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const uint8_t* W = (const uint8_t*)p.w8;
    const bf16* xg = (const bf16*)p.x;
    const bf16* gg = (const bf16*)p.norm_w;
    for (int n0 = blockIdx.x * R; n0 < p.N; n0 += gridDim.x * R) {
        const uint8_t* rows[R];
#pragma unroll
        for (int r = 0; r < R; ++r) rows[r] = W + (size_t)min(n0 + r, p.N - 1) * p.ldw;
        float acc[R + 1];
        f32x2 a2[R];
#pragma unroll
        for (int r = 0; r <= R; ++r) acc[r] = 0.0f;
#pragma unroll
        for (int r = 0; r < R; ++r) a2[r] = f32x2{0.0f, 0.0f};
        for (int ci = wave * 64 + lane; ci < nch8; ci += 64 * GEMV_WAVES) {
            uint4 w[R];
#pragma unroll
            for (int r = 0; r < R; ++r) w[r] = load_nt(rows[r] + (size_t)ci * 16);
            float xf[16];
            chunk_to_f32<bf16>(*(const uint4*)(xg + (size_t)ci * 16), xf);
            chunk_to_f32<bf16>(*(const uint4*)(xg + (size_t)ci * 16 + 8), xf + 8);
            if (NORM) {
                float gf[16];
                chunk_to_f32<bf16>(*(const uint4*)(gg + (size_t)ci * 16), gf);
                chunk_to_f32<bf16>(*(const uint4*)(gg + (size_t)ci * 16 + 8), gf + 8);
#pragma unroll
                for (int e = 0; e < 16; ++e) { acc[R] = fmaf(xf[e], xf[e], acc[R]); xf[e] *= gf[e]; }
            }
            f32x2 x2[8];
#pragma unroll
            for (int e = 0; e < 8; ++e) x2[e] = f32x2{xf[2 * e], xf[2 * e + 1]};
#pragma unroll
            for (int r = 0; r < R; ++r) fp8x16_dot(w[r], x2, a2[r]);
        }
#pragma unroll
        for (int r = 0; r < R; ++r) acc[r] = a2[r][0] + a2[r][1];
#pragma unroll
        for (int r = 0; r <= R; ++r) acc[r] = wave_sum(acc[r]);
        if (lane == 0) {
#pragma unroll
            for (int r = 0; r <= R; ++r) part[wave][r] = acc[r];
        }
        __syncthreads();
        if (threadIdx.x < R && n0 + threadIdx.x < p.N) {
            const int n = n0 + threadIdx.x;
            float v = (part[0][threadIdx.x] + part[1][threadIdx.x] + part[2][threadIdx.x] + part[3][threadIdx.x]) * p.scale[n];
            if (NORM) v *= rsqrtf((part[0][R] + part[1][R] + part[2][R] + part[3][R]) / (float)p.K + p.eps);
            if (p.bias) v += to_f32(((const bf16*)p.bias)[n]);
            if (p.res) v += to_f32(((const bf16*)p.res)[n]);
            ((bf16*)p.y)[n] = from_f32<bf16>(v);
        }
        __syncthreads();
    }
}

// per-row e4m3 quantisation: one workgroup per row, scale = max|w| / 448 (1 for an all-zero row), round-to-nearest-even
__global__ __launch_bounds__(256) void quant_fp8_rows_kernel(const bf16* w, int ld, uint8_t* q, float* scale, int cols) {
    __shared__ float red[4];
    const size_t row = blockIdx.x;
    const bf16* wr = w + row * ld;
    float amax = 0.0f;
    for (int ci = threadIdx.x; ci < cols / 8; ci += 256) {
        float f[8];
        chunk_to_f32<bf16>(*(const uint4*)(wr + (size_t)ci * 8), f);
#pragma unroll
        for (int e = 0; e < 8; ++e) amax = fmaxf(amax, fabsf(f[e]));
    }
    amax = wave_max(amax);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = amax;
    __syncthreads();
    amax = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
    const float sc = amax > 0.0f ? amax / 448.0f : 1.0f;
    if (threadIdx.x == 0) scale[row] = sc;
    const float inv = 1.0f / sc;
    for (int ci = threadIdx.x; ci < cols / 8; ci += 256) {
        float f[8];
        chunk_to_f32<bf16>(*(const uint4*)(wr + (size_t)ci * 8), f);
#pragma unroll
        for (int e = 0; e < 8; ++e) f[e] = fminf(fmaxf(f[e] * inv, -448.0f), 448.0f);
        int lo = 0, hi = 0;
        lo = __builtin_amdgcn_cvt_pk_fp8_f32(f[0], f[1], lo, false);
        lo = __builtin_amdgcn_cvt_pk_fp8_f32(f[2], f[3], lo, true);
        hi = __builtin_amdgcn_cvt_pk_fp8_f32(f[4], f[5], hi, false);
        hi = __builtin_amdgcn_cvt_pk_fp8_f32(f[6], f[7], hi, true);
        *(uint2*)(q + row * cols + (size_t)ci * 8) = make_uint2((unsigned)lo, (unsigned)hi);
    }
}

// final arg-max over per-workgroup partials: greatest value, lowest index on ties (torch.argmax on CPU)
// With `ctl` it is also one step of the greedy loop (GenerationMixin._sample: append, stop on EOS / max_new_tokens): see GenCtl.
__global__ __launch_bounds__(256) void argmax_final_kernel(const float* pv, const int* pi, int n, int* out_token, float* out_top, GenCtl* ctl,
                                                           const int* eos, int* out_ids, uint8_t* pen_flags) {
    __shared__ float sv[256];
    if (ctl && ctl->done) return;
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
    const int tok = si[0] == 0x7FFFFFFF ? -1 : si[0];      // no finite logit (NaN / -inf everywhere): -1, which the next embedding gather
                                                           // reads as frame-feature row 0 (in range) and the host reports as an error
    if (threadIdx.x == 0) {
        *out_token = tok;
        if (out_top) { out_top[0] = sv[0]; out_top[1] = s2[0]; }
    }
    if (ctl) {
        int hit = 0;
        for (int k = threadIdx.x; k < ctl->n_eos; k += 256) hit |= eos[k] == tok;
        hit = __syncthreads_or(hit);
        if (threadIdx.x == 0) {
            const int c = ctl->count;
            out_ids[c] = tok;
            if (pen_flags && tok >= 0) pen_flags[tok] = 1;
            ctl->count = c + 1;
            if (hit || tok < 0 || c + 1 >= ctl->max_new) ctl->done = 1;      // EOS is appended, never fed
            else { ctl->pos += 1; ctl->kv_len += 1; }
        }
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
    if (a.w8) {                                   // fp8 weights (bf16 engine only; the engine refuses to enable it otherwise)
        if (a.epi == EPI_NONE && a.N <= 8192) {
            constexpr int R = 4;
            int grid = (a.N + R - 1) / R;
            if (grid > 2048) grid = 2048;
            if (a.norm_w) SVLN_LAUNCH((gemv8_ksplit_kernel<true, R>), dim3(grid), b, 0);
            else SVLN_LAUNCH((gemv8_ksplit_kernel<false, R>), dim3(grid), b, 0);
            return;
        }
        const size_t lds8 = (size_t)a.K * sizeof(float);
        dim3 g8(gemv_grid(a.N));
        switch (a.epi) {
            case EPI_NONE: SVLN_LAUNCH((gemv8_kernel<EPI_NONE>), g8, b, lds8); break;
            case EPI_SWIGLU: SVLN_LAUNCH((gemv8_kernel<EPI_SWIGLU>), g8, b, lds8); break;
            case EPI_ARGMAX: SVLN_LAUNCH((gemv8_kernel<EPI_ARGMAX>), g8, b, lds8); break;
            default: break;
        }
        return;
    }
    if (a.epi == EPI_NONE && a.N <= 8192) {
        constexpr int R = 2;                       // rows per workgroup (R = 4 / 8 measured slower at N <= 8192)
        int grid = (a.N + R - 1) / R;
        if (grid > 2048) grid = 2048;
        if (a.norm_w) SVLN_LAUNCH((gemv_ksplit_kernel<T, true, R>), dim3(grid), b, 0);
        else SVLN_LAUNCH((gemv_ksplit_kernel<T, false, R>), dim3(grid), b, 0);
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
    set_max_lds((const void*)gemv_kernel<T, EPI>, 160 * 1024 - 256);
}
void launch_quant_fp8_rows(hipStream_t s, const void* w_bf16, int ld, void* w8, float* scale, int64_t rows, int cols) {
    hipLaunchKernelGGL(quant_fp8_rows_kernel, dim3((unsigned)rows), dim3(256), 0, s, (const bf16*)w_bf16, ld, (uint8_t*)w8, scale, cols);
}
void gemv_init_attrs() {
    // (every kernel that may ask for more than 64 KiB of dynamic LDS goes through set_max_lds: a refusal is reported at engine creation)
    set_max_lds((const void*)gemv8_kernel<EPI_NONE>, 160 * 1024 - 256);
    set_max_lds((const void*)gemv8_kernel<EPI_SWIGLU>, 160 * 1024 - 256);
    set_max_lds((const void*)gemv8_kernel<EPI_ARGMAX>, 160 * 1024 - 256);
    gemv_attr<bf16, EPI_NONE>(); gemv_attr<bf16, EPI_SWIGLU>(); gemv_attr<bf16, EPI_ARGMAX>();
    gemv_attr<float, EPI_NONE>(); gemv_attr<float, EPI_SWIGLU>(); gemv_attr<float, EPI_ARGMAX>();
}
template void launch_gemv<bf16>(hipStream_t, const GemvArgs&);
template void launch_gemv<float>(hipStream_t, const GemvArgs&);

void launch_argmax_final(hipStream_t s, const float* pv, const int* pi, int n, int* out_token, float* out_top) {
    hipLaunchKernelGGL(argmax_final_kernel, dim3(1), dim3(256), 0, s, pv, pi, n, out_token, out_top, (GenCtl*)nullptr, (const int*)nullptr, (int*)nullptr,
                       (uint8_t*)nullptr);
}
void launch_argmax_step(hipStream_t s, const float* pv, const int* pi, int n, int* out_token, float* out_top, GenCtl* ctl, const int* eos,
                        int* out_ids, uint8_t* pen_flags) {
    hipLaunchKernelGGL(argmax_final_kernel, dim3(1), dim3(256), 0, s, pv, pi, n, out_token, out_top, ctl, eos, out_ids, pen_flags);
}
namespace {
__global__ __launch_bounds__(256) void set_flags_kernel(uint8_t* flags, const int* ids, const int* count, int n_host, int value) {
    const int n = count ? *count : n_host;
    for (int k = blockIdx.x * 256 + threadIdx.x; k < n; k += gridDim.x * 256)
        if (ids[k] >= 0) flags[ids[k]] = (uint8_t)value;
}
}  // namespace
void launch_set_flags(hipStream_t s, uint8_t* flags, const int* ids, const int* count, int n_host, int value) {
    hipLaunchKernelGGL(set_flags_kernel, dim3(count ? 16 : (n_host + 255) / 256 > 0 ? (n_host + 255) / 256 : 1), dim3(256), 0, s, flags, ids, count, n_host, value);
}

}  // namespace svln
