// Bandwidth-bound helper kernels of the path (norms, RoPE + KV append, ViT K/V packing,
// patch extraction, bilinear token pooling, embedding splice, weight synthesis / conversion).
// All of them are HBM/L2-bound byte movers: 16-byte vector accesses, one wave per row where a
// row reduction is needed, no LDS.
#include "common.h"
#include "kernels.h"

namespace svln {

namespace {

// ------------------------------------------------------------------------------------------ norms
// one wave per row; three cached passes (the row stays in L1/L2): statistics in fp32.
// y2 (optional): a second copy of the normalised rows starting at row *y2_row (device scalar, clamped to y2_cap - 1) of y2 -- the hidden tap
// of a decode step whose index only the device knows (GenCtl.count), so the launch can sit in a captured graph
template <typename T>
__global__ __launch_bounds__(256) void rmsnorm_kernel(const T* x, const T* g, T* y, int rows, int n, float eps, const int* skip, T* y2,
                                                      const int* y2_row, int y2_cap) {
    constexpr int EPC = Elt<T>::PER_CHUNK;
    if (skip && *skip) return;
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    const T* xr = x + (size_t)row * n;
    T* yr = y + (size_t)row * n;
    const int nch = n / EPC;
    float ss = 0.0f;
    for (int ci = lane; ci < nch; ci += 64) {
        float f[EPC];
        chunk_to_f32<T>(*(const uint4*)(xr + (size_t)ci * EPC), f);
#pragma unroll
        for (int e = 0; e < EPC; ++e) ss += f[e] * f[e];
    }
    ss = wave_sum(ss);
    const float sc = rsqrtf(ss / (float)n + eps);
    for (int ci = lane; ci < nch; ci += 64) {
        float f[EPC], gf[EPC];
        chunk_to_f32<T>(*(const uint4*)(xr + (size_t)ci * EPC), f);
        chunk_to_f32<T>(*(const uint4*)(g + (size_t)ci * EPC), gf);
#pragma unroll
        for (int e = 0; e < EPC; ++e) f[e] = gf[e] * (f[e] * sc);
        const uint4 o = f32_to_chunk<T>(f);
        *(uint4*)(yr + (size_t)ci * EPC) = o;
        if (y2) {
            const int r2 = min(*y2_row + row, y2_cap - 1);
            *(uint4*)(y2 + (size_t)r2 * n + (size_t)ci * EPC) = o;
        }
    }
}

template <typename T>
__global__ __launch_bounds__(256) void layernorm_kernel(const T* x, const T* g, const T* b, T* y, int rows, int n, float eps) {
    constexpr int EPC = Elt<T>::PER_CHUNK;
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    const T* xr = x + (size_t)row * n;
    T* yr = y + (size_t)row * n;
    const int nch = n / EPC;
    float s = 0.0f;
    for (int ci = lane; ci < nch; ci += 64) {
        float f[EPC];
        chunk_to_f32<T>(*(const uint4*)(xr + (size_t)ci * EPC), f);
#pragma unroll
        for (int e = 0; e < EPC; ++e) s += f[e];
    }
    const float mu = wave_sum(s) / (float)n;
    float v = 0.0f;
    for (int ci = lane; ci < nch; ci += 64) {
        float f[EPC];
        chunk_to_f32<T>(*(const uint4*)(xr + (size_t)ci * EPC), f);
#pragma unroll
        for (int e = 0; e < EPC; ++e) v += (f[e] - mu) * (f[e] - mu);
    }
    const float sc = rsqrtf(wave_sum(v) / (float)n + eps);
    for (int ci = lane; ci < nch; ci += 64) {
        float f[EPC], gf[EPC], bf[EPC];
        chunk_to_f32<T>(*(const uint4*)(xr + (size_t)ci * EPC), f);
        chunk_to_f32<T>(*(const uint4*)(g + (size_t)ci * EPC), gf);
        chunk_to_f32<T>(*(const uint4*)(b + (size_t)ci * EPC), bf);
#pragma unroll
        for (int e = 0; e < EPC; ++e) f[e] = (f[e] - mu) * sc * gf[e] + bf[e];
        *(uint4*)(yr + (size_t)ci * EPC) = f32_to_chunk<T>(f);
    }
}

// --------------------------------------------------------------------------------- RoPE + KV append
// grid (T, nq + nkv + nkv), 64 threads: lane d pairs (d, d + 64) of a 128-wide head.
// q heads: rotate in place.  k heads: rotate, write K page row.  v heads: write transposed Vt page.
// A wave per (row, head), four heads per workgroup (70 000 single-wave workgroups at T = 1952 are dispatch-bound).
template <typename T>
__global__ __launch_bounds__(256) void rope_kv_kernel(RopeKvArgs p) {
    const int i = blockIdx.x, hd = blockIdx.y * 4 + (threadIdx.x >> 6), d = threadIdx.x & 63;
    if (hd >= p.nq + 2 * p.nkv) return;
    const int P = p.dyn_pos ? *p.dyn_pos : p.P;
    const int pos = P + i;
    T* row = (T*)p.qkv + (size_t)i * p.ld + (size_t)hd * 128;
    const float x1 = to_f32(row[d]), x2 = to_f32(row[d + 64]);
    const int page = p.page_table[pos >> 6], off = pos & 63;
    if (hd < p.nq + p.nkv) {
        const float c = p.rope_tab[(size_t)pos * 128 + d], s = p.rope_tab[(size_t)pos * 128 + 64 + d];
        const float o1 = x1 * c - x2 * s, o2 = x2 * c + x1 * s;     // q*cos + rotate_half(q)*sin
        if (hd < p.nq) {
            row[d] = from_f32<T>(o1);
            row[d + 64] = from_f32<T>(o2);
        } else {
            const int kh = hd - p.nq;
            T* kr = (T*)p.Kpool + (((size_t)page * p.nkv + kh) * 64 + off) * 128;
            kr[d] = from_f32<T>(o1);
            kr[d + 64] = from_f32<T>(o2);
        }
    } else {
        const int kh = hd - p.nq - p.nkv;
        T* vt = (T*)p.Vpool + ((size_t)page * p.nkv + kh) * 128 * 64;
        vt[(size_t)d * 64 + off] = from_f32<T>(x1);
        vt[(size_t)(d + 64) * 64 + off] = from_f32<T>(x2);
    }
}

// cos/sin of pos * inv_freq[d] in fp32 (Qwen2RotaryEmbedding, modeling_qwen2.py:119-131), tabulated once
__global__ __launch_bounds__(64) void rope_table_kernel(float* tab, const float* inv_freq, int positions) {
    const int pos = blockIdx.x, d = threadIdx.x;
    const float ang = (float)pos * inv_freq[d];
    tab[(size_t)pos * 128 + d] = cosf(ang);
    tab[(size_t)pos * 128 + 64 + d] = sinf(ang);
}

// 128-bit content key of each frame: two position-dependent, order-independent 64-bit sums over the pixel words
// (64-bit atomic adds of per-thread partial sums), used to memoise pooled frame features (engine feature cache).
__global__ __launch_bounds__(256) void frame_hash_kernel(const uint32_t* pix, size_t words_per_frame, unsigned long long* out) {
    const int f = blockIdx.y;
    const uint32_t* p = pix + (size_t)f * words_per_frame;
    unsigned long long a = 0, b = 0;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < words_per_frame; i += (size_t)gridDim.x * 256) {
        const uint64_t v = (uint64_t)p[i] | ((uint64_t)i << 32);
        a += splitmix64(v ^ 0x243F6A8885A308D3ull);
        b += splitmix64(v * 0x9E3779B97F4A7C15ull + 0x13198A2E03707344ull);
    }
    atomicAdd(out + 2 * f, a);
    atomicAdd(out + 2 * f + 1, b);
}

// ViT K/V packing: qkv [F*S][3*Hv] -> K pages [tile][F*heads][64][HDP] (zero padded), Vt [tile][F*heads][VROWS][64].
// One workgroup per (64-key tile, frame*head).  K rows are copied as 16-byte chunks; V goes through an LDS tile
// [64 keys][HD] (16-byte chunk loads, row pitch padded by one chunk against bank conflicts) and leaves transposed as
// 16-byte chunks of 8 (4 for fp32) consecutive keys per head dim.  Requires HD % chunk == 0 and ld % chunk == 0.
template <typename T>
__global__ __launch_bounds__(256) void vit_kv_pack_kernel(const T* qkv, int ld, T* Kpool, T* Vpool, int F, int S, int heads, int HD,
                                                          int HDP, int VROWS) {
    extern __shared__ __attribute__((aligned(16))) char pack_smem[];
    constexpr int EPC = Elt<T>::PER_CHUNK;
    T* vt = (T*)pack_smem;
    const int tile = blockIdx.x, kh = blockIdx.y;          // kh = f*heads + head
    const int f = kh / heads, head = kh % heads, Hv = heads * HD;
    const int nkv = F * heads;
    const int hc = HD / EPC, hpc = HDP / EPC, pitch = HD + EPC;
    T* kp = Kpool + ((size_t)tile * nkv + kh) * 64 * HDP;
    T* vp = Vpool + ((size_t)tile * nkv + kh) * VROWS * 64;
    const uint4 zero = make_uint4(0, 0, 0, 0);
    for (int e = threadIdx.x; e < 64 * hpc; e += 256) {
        const int key = e / hpc, cchunk = e % hpc, s = tile * 64 + key;
        uint4 v = zero;
        if (s < S && cchunk < hc) v = *(const uint4*)(qkv + (size_t)(f * S + s) * ld + Hv + head * HD + cchunk * EPC);
        *(uint4*)(kp + (size_t)key * HDP + cchunk * EPC) = v;
    }
    for (int e = threadIdx.x; e < 64 * hc; e += 256) {
        const int key = e / hc, cchunk = e % hc, s = tile * 64 + key;
        uint4 v = zero;
        if (s < S) v = *(const uint4*)(qkv + (size_t)(f * S + s) * ld + 2 * Hv + head * HD + cchunk * EPC);
        *(uint4*)(vt + key * pitch + cchunk * EPC) = v;
    }
    __syncthreads();
    constexpr int KC = 64 / EPC;                          // key chunks per Vt row
    for (int e = threadIdx.x; e < VROWS * KC; e += 256) {
        const int d = e / KC, kc = e % KC;
        T out[EPC];
#pragma unroll
        for (int j = 0; j < EPC; ++j) out[j] = d < HD ? vt[(kc * EPC + j) * pitch + d] : from_f32<T>(0.0f);
        *(uint4*)(vp + (size_t)d * 64 + kc * EPC) = *(const uint4*)out;
    }
}

// ------------------------------------------------------------------------------------------ vision
template <typename T>
__global__ __launch_bounds__(256) void patchify_kernel(const float* pix, T* out, int F, int image, int patch, int kp) {
    const int side = image / patch;
    const int prow = blockIdx.x;                          // f*side*side + py*side + px
    const int f = prow / (side * side), pp = prow % (side * side), py = pp / side, px = pp % side;
    const int kreal = 3 * patch * patch;
    for (int k = threadIdx.x; k < kp; k += 256) {
        float v = 0.0f;
        if (k < kreal) {
            const int c = k / (patch * patch), rem = k % (patch * patch), ky = rem / patch, kx = rem % patch;
            v = pix[(((size_t)f * 3 + c) * image + (py * patch + ky)) * image + px * patch + kx];
        }
        out[(size_t)prow * kp + k] = from_f32<T>(v);
    }
}

// bilinear pooling, ATen upsample_bilinear2d order: w0y*(w0x*a + w1x*b) + w1y*(w0x*c + w1x*d)
template <typename T>
__global__ __launch_bounds__(256) void pool_kernel(const T* in, T* out, const int* tap_idx, const float* tap_w, int side, int oside, int C) {
    constexpr int EPC = Elt<T>::PER_CHUNK;
    const int o = blockIdx.x;                              // f*oside*oside + oy*oside + ox
    const int f = o / (oside * oside), oo = o % (oside * oside), oy = oo / oside, ox = oo % oside;
    const int y0 = tap_idx[2 * oy], y1 = tap_idx[2 * oy + 1], x0 = tap_idx[2 * ox], x1 = tap_idx[2 * ox + 1];
    const float wy0 = tap_w[2 * oy], wy1 = tap_w[2 * oy + 1], wx0 = tap_w[2 * ox], wx1 = tap_w[2 * ox + 1];
    const T* base = in + (size_t)f * side * side * C;
    const T* a = base + (size_t)(y0 * side + x0) * C;
    const T* b = base + (size_t)(y0 * side + x1) * C;
    const T* c = base + (size_t)(y1 * side + x0) * C;
    const T* d = base + (size_t)(y1 * side + x1) * C;
    T* orow = out + (size_t)o * C;
    for (int ci = threadIdx.x; ci < C / EPC; ci += 256) {
        float fa[EPC], fb[EPC], fc[EPC], fd[EPC], r[EPC];
        chunk_to_f32<T>(*(const uint4*)(a + (size_t)ci * EPC), fa);
        chunk_to_f32<T>(*(const uint4*)(b + (size_t)ci * EPC), fb);
        chunk_to_f32<T>(*(const uint4*)(c + (size_t)ci * EPC), fc);
        chunk_to_f32<T>(*(const uint4*)(d + (size_t)ci * EPC), fd);
#pragma unroll
        for (int e = 0; e < EPC; ++e) r[e] = wy0 * (wx0 * fa[e] + wx1 * fb[e]) + wy1 * (wx0 * fc[e] + wx1 * fd[e]);
        *(uint4*)(orow + (size_t)ci * EPC) = f32_to_chunk<T>(r);
    }
}

template <typename T>
__global__ __launch_bounds__(256) void gather_rows_kernel(const int* src, const T* embed, const T* feats, T* out, int n, const int* skip) {
    constexpr int EPC = Elt<T>::PER_CHUNK;
    if (skip && *skip) return;
    const int row = blockIdx.x;
    const int sidx = src[row];
    const T* s = sidx >= 0 ? embed + (size_t)sidx * n : feats + (size_t)(-(sidx + 1)) * n;
    T* o = out + (size_t)row * n;
    for (int ci = threadIdx.x; ci < n / EPC; ci += 256) *(uint4*)(o + (size_t)ci * EPC) = *(const uint4*)(s + (size_t)ci * EPC);
}


// --------------------------------------------------------------------------- slow-memory token pruning
// OPT-IN EXTENSION (BASELINE configs[3]; SURVEY.md a-13: the reference has no counterpart, parity is pinned only by this
// project's own CPU restatement oracle/streamvln_oracle.py: prune_memory_tokens).  Rule: score_i = cos(M_i, mean_j M_j) over the
// N memory tokens; the `keep` tokens with the SMALLEST score (the least like the average token; ties: lower index) survive,
// in their original order.  Four launches: column partial sums (64-row blocks, fixed order -> deterministic), mean, per-token
// score (one wave per token), rank by counting (one wave per token), compaction of the kept indices (one workgroup, ballot scan).
template <typename T>
__global__ __launch_bounds__(256) void mem_colsum_kernel(const T* m, int n_rows, int H, float* partial) {
    const int col = blockIdx.x * 256 + threadIdx.x, r0 = blockIdx.y * 64;
    if (col >= H) return;
    float acc = 0.0f;
    for (int r = r0; r < min(r0 + 64, n_rows); ++r) acc += to_f32(m[(size_t)r * H + col]);
    partial[(size_t)blockIdx.y * H + col] = acc;
}
__global__ __launch_bounds__(256) void mem_mean_kernel(const float* partial, int n_blocks, int n_rows, int H, float* mean) {
    const int col = blockIdx.x * 256 + threadIdx.x;
    if (col >= H) return;
    float mu = 0.0f;
    for (int b = 0; b < n_blocks; ++b) mu += partial[(size_t)b * H + col];
    mean[col] = mu / (float)n_rows;
}
template <typename T>
__global__ __launch_bounds__(256) void mem_score_kernel(const T* m, int n_rows, int H, const float* mean, float* score) {
    constexpr int EPC = Elt<T>::PER_CHUNK;
    const int lane = threadIdx.x & 63, row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= n_rows) return;
    float dot = 0.0f, nn = 0.0f, mm = 0.0f;
    for (int ci = lane; ci < H / EPC; ci += 64) {
        float f[EPC];
        chunk_to_f32<T>(*(const uint4*)(m + (size_t)row * H + (size_t)ci * EPC), f);
#pragma unroll
        for (int e = 0; e < EPC; ++e) {
            const float mu = mean[ci * EPC + e];
            dot = fmaf(f[e], mu, dot); nn = fmaf(f[e], f[e], nn); mm = fmaf(mu, mu, mm);
        }
    }
    dot = wave_sum(dot); nn = wave_sum(nn); mm = wave_sum(mm);
    if (lane == 0) score[row] = dot / fmaxf(sqrtf(nn) * sqrtf(mm), 1e-20f);
}
// rank by counting, one wave per token: flag[i] = (number of tokens that sort before i) < keep.  n / 4 workgroups instead of the single
// workgroup of round 2 (162 us at n = 1568).
__global__ __launch_bounds__(256) void mem_rank_kernel(const float* score, int n, int keep, int* flag) {
    const int lane = threadIdx.x & 63, i = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (i >= n) return;
    const float si = score[i];
    int rank = 0;
    for (int j = lane; j < n; j += 64) {
        const float sj = score[j];
        rank += (sj < si || (sj == si && j < i)) ? 1 : 0;
    }
    rank = (int)wave_sum((float)rank);        // counts <= 2^24 are exact in fp32
    if (lane == 0) flag[i] = rank < keep ? 1 : 0;
}
// sel = ascending indices of the flagged tokens: one workgroup, block-wide exclusive scan of the flags (wave ballots + wave totals)
__global__ __launch_bounds__(1024) void mem_compact_kernel(const int* flag, int n, int* sel) {
    __shared__ int wtot[16];
    __shared__ int base;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    if (tid == 0) base = 0;
    __syncthreads();
    for (int i0 = 0; i0 < n; i0 += 1024) {
        const int i = i0 + tid;
        const int f = i < n ? flag[i] : 0;
        const unsigned long long b = __ballot(f != 0);
        const int before = __popcll(b & ((1ull << lane) - 1ull));
        if (lane == 0) wtot[wave] = __popcll(b);
        __syncthreads();
        int off = base;
        for (int w = 0; w < wave; ++w) off += wtot[w];
        if (f) sel[off + before] = i;
        __syncthreads();
        if (tid == 0) { int t = 0; for (int w = 0; w < 16; ++w) t += wtot[w]; base += t; }
        __syncthreads();
    }
}

// ----------------------------------------------------------------------------------------- weights
SVLN_DEV int64_t map_row(int64_t r, RowMap m) { return (r / m.blk) * (int64_t)m.blk * m.nint + (int64_t)m.phase * m.blk + r % m.blk; }

template <typename T>
__global__ __launch_bounds__(256) void synth_kernel(T* dst, int dst_ld, int64_t rows, int cols, RowMap m, uint64_t seed_t, float step,
                                                    float base) {
    const int64_t total = rows * cols;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
        const int64_t r = i / cols;
        const int c = (int)(i - r * cols);
        // weights are defined as bf16 values (like the real checkpoint); the fp32 engine stores them widened
        dst[map_row(r, m) * dst_ld + c] = from_f32<T>(to_f32((bf16)synth_value(seed_t, (uint64_t)i, step, base)));
    }
}

template <typename T>
__global__ __launch_bounds__(256) void convert_kernel(T* dst, int dst_ld, int64_t rows, int cols, RowMap m, const void* src, int src_is_f32) {
    const int64_t total = rows * cols;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
        const int64_t r = i / cols;
        const int c = (int)(i - r * cols);
        const float v = src_is_f32 ? ((const float*)src)[i] : to_f32(((const bf16*)src)[i]);
        dst[map_row(r, m) * dst_ld + c] = from_f32<T>(v);
    }
}

template <typename T> __global__ void to_f32_kernel(const T* s, float* d, int64_t n) {
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) d[i] = to_f32(s[i]);
}
template <typename T> __global__ void from_f32_kernel(const float* s, T* d, int64_t n) {
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) d[i] = from_f32<T>(s[i]);
}

int grid_for(int64_t n) {
    int64_t g = (n + 255) / 256;
    return (int)(g < 1 ? 1 : (g > 8192 ? 8192 : g));
}

}  // namespace



template <typename T> void launch_rmsnorm(hipStream_t s, const void* x, const void* g, void* y, int rows, int n, float eps, const int* skip,
                                          void* y2, const int* y2_row, int y2_cap) {
    if (rows <= 0) return;
    hipLaunchKernelGGL((rmsnorm_kernel<T>), dim3((rows + 3) / 4), dim3(256), 0, s, (const T*)x, (const T*)g, (T*)y, rows, n, eps, skip, (T*)y2,
                       y2_row, y2_cap);
}
template <typename T> void launch_layernorm(hipStream_t s, const void* x, const void* g, const void* b, void* y, int rows, int n, float eps) {
    if (rows <= 0) return;
    hipLaunchKernelGGL((layernorm_kernel<T>), dim3((rows + 3) / 4), dim3(256), 0, s, (const T*)x, (const T*)g, (const T*)b, (T*)y, rows, n, eps);
}
template <typename T> void launch_rope_kv(hipStream_t s, const RopeKvArgs& a) {
    hipLaunchKernelGGL((rope_kv_kernel<T>), dim3(a.T, (a.nq + 2 * a.nkv + 3) / 4), dim3(256), 0, s, a);
}
void launch_frame_hash(hipStream_t s, const float* pix, int F, size_t words_per_frame, unsigned long long* out) {
    hipLaunchKernelGGL(frame_hash_kernel, dim3(64, F), dim3(256), 0, s, (const uint32_t*)pix, words_per_frame, out);
}
void launch_rope_table(hipStream_t s, float* tab, const float* inv_freq, int positions) {
    hipLaunchKernelGGL(rope_table_kernel, dim3(positions), dim3(64), 0, s, tab, inv_freq, positions);
}
template <typename T> void launch_vit_kv_pack(hipStream_t s, const void* qkv, int ld, void* Kpool, void* Vpool, int F, int S, int heads,
                                             int head_dim) {
    constexpr int EPC = Elt<T>::PER_CHUNK;
    const int hdc = (((head_dim + EPC - 1) / EPC) + 1) & ~1;
    const int vrows = ((head_dim + 31) / 32) * 32;
    const size_t lds = (size_t)64 * (head_dim + EPC) * sizeof(T);
    hipLaunchKernelGGL((vit_kv_pack_kernel<T>), dim3((S + 63) / 64, F * heads), dim3(256), lds, s, (const T*)qkv, ld, (T*)Kpool, (T*)Vpool, F, S,
                       heads, head_dim, hdc * EPC, vrows);
}
template <typename T> void launch_patchify(hipStream_t s, const float* pix, void* out, int F, int image, int patch, int kp) {
    const int side = image / patch;
    hipLaunchKernelGGL((patchify_kernel<T>), dim3(F * side * side), dim3(256), 0, s, pix, (T*)out, F, image, patch, kp);
}
template <typename T> void launch_pool(hipStream_t s, const void* in, void* out, const int* tap_idx, const float* tap_w, int F, int side,
                                       int out_side, int C) {
    hipLaunchKernelGGL((pool_kernel<T>), dim3(F * out_side * out_side), dim3(256), 0, s, (const T*)in, (T*)out, tap_idx, tap_w, side, out_side, C);
}
template <typename T> void launch_gather_rows(hipStream_t s, const int* src, const void* embed, const void* feats, void* out, int rows, int n,
                                              const int* skip) {
    if (rows <= 0) return;
    hipLaunchKernelGGL((gather_rows_kernel<T>), dim3(rows), dim3(256), 0, s, src, (const T*)embed, (const T*)feats, (T*)out, n, skip);
}
// scratch: partial [ceil(n/64)][H] floats, mean [H], score [n]; sel [keep] ints (ascending row indices)
template <typename T> void launch_memory_prune(hipStream_t s, const void* m, int n_rows, int H, int keep, float* partial, float* mean, float* score,
                                               int* sel) {
    const int nb = (n_rows + 63) / 64;
    hipLaunchKernelGGL((mem_colsum_kernel<T>), dim3((H + 255) / 256, nb), dim3(256), 0, s, (const T*)m, n_rows, H, partial);
    hipLaunchKernelGGL(mem_mean_kernel, dim3((H + 255) / 256), dim3(256), 0, s, partial, nb, n_rows, H, mean);
    hipLaunchKernelGGL((mem_score_kernel<T>), dim3((n_rows + 3) / 4), dim3(256), 0, s, (const T*)m, n_rows, H, mean, score);
    int* flag = (int*)partial;              // the column partial sums are consumed by now: their buffer (>= ceil(n / 64) * H floats) holds the n flags
    hipLaunchKernelGGL(mem_rank_kernel, dim3((n_rows + 3) / 4), dim3(256), 0, s, score, n_rows, keep, flag);
    hipLaunchKernelGGL(mem_compact_kernel, dim3(1), dim3(1024), 0, s, flag, n_rows, sel);
}
template <typename T> void launch_synth(hipStream_t s, void* dst, int dst_ld, int64_t rows, int cols, RowMap m, uint64_t seed_t,
                                        float half_width, float base) {
    const float step = half_width / 8388608.0f;
    hipLaunchKernelGGL((synth_kernel<T>), dim3(grid_for(rows * cols)), dim3(256), 0, s, (T*)dst, dst_ld, rows, cols, m, seed_t, step, base);
}
template <typename T> void launch_convert(hipStream_t s, void* dst, int dst_ld, int64_t rows, int cols, RowMap m, const void* src, int src_is_f32) {
    hipLaunchKernelGGL((convert_kernel<T>), dim3(grid_for(rows * cols)), dim3(256), 0, s, (T*)dst, dst_ld, rows, cols, m, src, src_is_f32);
}
template <typename T> void launch_to_f32(hipStream_t s, const void* src, float* dst, int64_t n) {
    if (n <= 0) return;
    hipLaunchKernelGGL((to_f32_kernel<T>), dim3(grid_for(n)), dim3(256), 0, s, (const T*)src, dst, n);
}
template <typename T> void launch_from_f32(hipStream_t s, const float* src, void* dst, int64_t n) {
    if (n <= 0) return;
    hipLaunchKernelGGL((from_f32_kernel<T>), dim3(grid_for(n)), dim3(256), 0, s, src, (T*)dst, n);
}

#define SVLN_INST(T)                                                                                                              \
    template void launch_rmsnorm<T>(hipStream_t, const void*, const void*, void*, int, int, float, const int*, void*, const int*, int);                              \
    template void launch_layernorm<T>(hipStream_t, const void*, const void*, const void*, void*, int, int, float);               \
    template void launch_rope_kv<T>(hipStream_t, const RopeKvArgs&);                                                              \
    template void launch_vit_kv_pack<T>(hipStream_t, const void*, int, void*, void*, int, int, int, int);                         \
    template void launch_patchify<T>(hipStream_t, const float*, void*, int, int, int, int);                                       \
    template void launch_pool<T>(hipStream_t, const void*, void*, const int*, const float*, int, int, int, int);                  \
    template void launch_gather_rows<T>(hipStream_t, const int*, const void*, const void*, void*, int, int, const int*);                      \
    template void launch_memory_prune<T>(hipStream_t, const void*, int, int, int, float*, float*, float*, int*);                   \
    template void launch_synth<T>(hipStream_t, void*, int, int64_t, int, RowMap, uint64_t, float, float);                          \
    template void launch_convert<T>(hipStream_t, void*, int, int64_t, int, RowMap, const void*, int);                              \
    template void launch_to_f32<T>(hipStream_t, const void*, float*, int64_t);                                                    \
    template void launch_from_f32<T>(hipStream_t, const float*, void*, int64_t);
SVLN_INST(bf16)
SVLN_INST(float)

}  // namespace svln
