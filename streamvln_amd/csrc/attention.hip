// Flash-style attention over a paged K / V^T cache -- one kernel for the three uses on the path:
//   * LLM prefill  (causal, bottom-right aligned over cached context + new tokens, GQA 7:1, hd 128)
//   * LLM decode   (one query position = 7 q-head rows per kv head, split-KV over key tiles)
//   * SigLIP ViT   (non-causal, 729 keys, hd 72 padded to an even number of 16-byte chunks)
//
// Layout (this engine's own; reference keeps a contiguous torch.cat cache):
//   K  pages [page][kvh][64 keys][HDP]     row = one key, HD contiguous  -> A operand of S^T = K.Q^T
//   Vt pages [page][kvh][DT*32][64 keys]   row = one head-dim channel, keys contiguous -> A operand of
//                                          O^T = Vt.P^T, so no transpose is ever done on chip
// A workgroup = WAVES waves, each wave owns 32 query rows of one kv head (row rho = i*G + g, so the
// 7 q-heads of a GQA group sit next to each other and share every K/V tile read).  Per 64-key tile:
//   coalesced 16-byte loads of the page -> XOR-swizzled LDS tile (shared by all waves)
//   S^T = K.Q^T on MFMA with the key on the accumulator ROW and the query on the LANE, so the
//   online-softmax max/sum over keys is an in-lane reduction + ONE cross-half wavefront shuffle
//   (__shfl_xor 32); P stays in registers and feeds the second MFMA directly as its B operand.
// Split-KV (decode): grid.z splits write (m, l, unnormalised O) partials; a combine kernel merges.
//
// Roofline: decode = HBM (KV bytes 2*nkv*hd*len*sizeof(T) per layer); prefill/ViT = MFMA.
#include "common.h"
#include "kernels.h"

namespace svln {

namespace {

template <typename T, int HD> struct AttnGeom {
    static constexpr int EPC = Elt<T>::PER_CHUNK;
    static constexpr int HDC = (((HD + EPC - 1) / EPC) + 1) & ~1;     // 16-byte chunks per K row (even)
    static constexpr int HDP = HDC * EPC;                              // padded head dim (elements)
    static constexpr int DT = (HD + 31) / 32;                          // 32-row output tiles along d
    static constexpr int VROWS = DT * 32;
    static constexpr int VC = 64 / EPC;                                // chunks per Vt row (64 keys)
    static constexpr int K_TILE_BYTES = 64 * HDC * 16;
    static constexpr int V_TILE_BYTES = VROWS * VC * 16;
    static constexpr bool KSWZ = (HDC % 16) == 0;
};

// K tile: chunk c of key row `row`
template <typename G> SVLN_DEV int k_off(int row, int c) {
    return row * (G::HDC * 16) + ((G::KSWZ ? (c ^ (row & 15)) : c) << 4);
}
// Vt tile, float: 16 chunks (256 B) per row, ds_read_b128 -> XOR chunk with row & 15
SVLN_DEV int v_off_f32(int row, int c) { return row * 256 + ((c ^ (row & 15)) << 4); }
// Vt tile, bf16: 16 units of 8 B (128 B) per row, ds_read_b64 by 32-lane halves -> XOR unit with (row>>1)&15
SVLN_DEV int v_off_bf16(int row, int unit) { return row * 128 + ((unit ^ ((row >> 1) & 15)) << 3); }

template <typename T, int HD, int WAVES>
__global__ __launch_bounds__(WAVES * 64) void attn_kernel(AttnArgs p) {
    using G = AttnGeom<T, HD>;
    constexpr int NT = WAVES * 64;
    constexpr int EPC = G::EPC;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* sK = smem;
    char* sV = smem + G::K_TILE_BYTES;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int r = lane & 31, h = lane >> 5;
    const int kh = blockIdx.y;
    const int frame = kh / p.hpf, head0 = (kh % p.hpf) * p.G;
    // first page id is independent of kv_len: fetch both scalars together (one round trip instead of two)
    const int kt0 = blockIdx.z * p.tiles_per_split;
    const int page0 = p.page_table ? p.page_table[kt0] : kt0;
    const int kv_len = p.dyn_kv_len ? *p.dyn_kv_len : p.kv_len;
    const int P = p.dyn_kv_len ? kv_len - p.T : p.P;
    const int rows_total = p.T * p.G;
    const int rho = (blockIdx.x * WAVES + wave) * 32 + r;
    const bool valid = rho < rows_total;
    const int qi = valid ? rho / p.G : 0;
    const int qg = valid ? rho - qi * p.G : 0;
    const int qpos = P + qi;

    // Q fragments (B operand of S^T): lane (r,h) holds chunk 2s+h of query row rho
    uint4 qf[G::HDC / 2];
    {
        const T* qrow = (const T*)p.Q + (size_t)(frame * p.T + qi) * p.q_stride + (size_t)(head0 + qg) * HD;
#pragma unroll
        for (int s = 0; s < G::HDC / 2; ++s) {
            const int e0 = (2 * s + h) * EPC;
            qf[s] = (valid && e0 < HD) ? *(const uint4*)(qrow + e0) : zero_chunk();
        }
    }

    // key-tile range of this workgroup
    int tiles = (kv_len + 63) >> 6;
    if (p.causal) {
        const int last_row = min((int)(blockIdx.x + 1) * WAVES * 32, rows_total) - 1;
        const int last_pos = P + last_row / p.G;
        tiles = min(tiles, (last_pos >> 6) + 1);
    }
    const int kt_begin = blockIdx.z * p.tiles_per_split;
    const int kt_end = min(tiles, kt_begin + p.tiles_per_split);

    f32x16 O[G::DT];
#pragma unroll
    for (int d = 0; d < G::DT; ++d)
#pragma unroll
        for (int e = 0; e < 16; ++e) O[d][e] = 0.0f;
    float m = -INFINITY, l = 0.0f;

    const size_t k_page_stride = (size_t)p.n_kv_total * 64 * G::HDP * sizeof(T);
    const size_t v_page_stride = (size_t)p.n_kv_total * G::VROWS * 64 * sizeof(T);

    for (int kt = kt_begin; kt < kt_end; ++kt) {
        const int page = kt == kt0 ? page0 : (p.page_table ? p.page_table[kt] : kt);
        const char* gK = (const char*)p.Kpool + (size_t)page * k_page_stride + (size_t)kh * 64 * G::HDP * sizeof(T);
        const char* gV = (const char*)p.Vpool + (size_t)page * v_page_stride + (size_t)kh * G::VROWS * 64 * sizeof(T);
        // ---- stage K and Vt tiles: coalesced 16-byte loads issued in batches (all in flight before the first
        //      dependent LDS write), swizzled LDS writes
        {
            constexpr int KTOT = 64 * G::HDC, VTOT = G::VROWS * G::VC, B = WAVES == 1 ? 4 : 8;
            constexpr int KL = (KTOT + NT - 1) / NT, VL = (VTOT + NT - 1) / NT;
#pragma unroll
            for (int b0 = 0; b0 < KL; b0 += B) {
                uint4 t[B];
#pragma unroll
                for (int u = 0; u < B; ++u) {
                    const int q = tid + (b0 + u) * NT;
                    t[u] = (b0 + u < KL && q < KTOT) ? *(const uint4*)(gK + (size_t)q * 16) : zero_chunk();
                }
#pragma unroll
                for (int u = 0; u < B; ++u) {
                    const int q = tid + (b0 + u) * NT;
                    if (b0 + u < KL && q < KTOT) {
                        const int row = q / G::HDC, c = q - row * G::HDC;
                        *(uint4*)(sK + k_off<G>(row, c)) = t[u];
                    }
                }
            }
#pragma unroll
            for (int b0 = 0; b0 < VL; b0 += B) {
                uint4 t[B];
#pragma unroll
                for (int u = 0; u < B; ++u) {
                    const int q = tid + (b0 + u) * NT;
                    t[u] = (b0 + u < VL && q < VTOT) ? *(const uint4*)(gV + (size_t)q * 16) : zero_chunk();
                }
#pragma unroll
                for (int u = 0; u < B; ++u) {
                    const int q = tid + (b0 + u) * NT;
                    if (b0 + u < VL && q < VTOT) {
                        const int row = q / G::VC, c = q - row * G::VC;
                        if (sizeof(T) == 4) {
                            *(uint4*)(sV + v_off_f32(row, c)) = t[u];
                        } else {
                            *(uint2*)(sV + v_off_bf16(row, 2 * c)) = make_uint2(t[u].x, t[u].y);
                            *(uint2*)(sV + v_off_bf16(row, 2 * c + 1)) = make_uint2(t[u].z, t[u].w);
                        }
                    }
                }
            }
        }
        __syncthreads();

        // ---- S^T[j] = K_tile[j*32 .. j*32+31] . Q^T   (rows = keys, col = this lane's query)
        f32x16 S[2];
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) S[j][e] = 0.0f;
#pragma unroll
        for (int s = 0; s < G::HDC / 2; ++s) {
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const uint4 a = *(const uint4*)(sK + k_off<G>(j * 32 + r, 2 * s + h));
                mma_chunk<T>(a, qf[s], S[j]);
            }
        }

        // ---- mask, online softmax (per lane = per query row; other half of the keys is in lane ^ 32)
        float mloc = -INFINITY;
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const int key = kt * 64 + j * 32 + acc_row(e, lane);
                const bool ok = key < kv_len && (!p.causal || key <= qpos);
                const float sv = ok ? S[j][e] * p.scale : -INFINITY;
                S[j][e] = sv;
                mloc = fmaxf(mloc, sv);
            }
        mloc = fmaxf(mloc, __shfl_xor(mloc, 32, 64));
        const float mnew = fmaxf(m, mloc);
        float alpha = 1.0f, psum = 0.0f;
        if (mnew == -INFINITY) {
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int e = 0; e < 16; ++e) S[j][e] = 0.0f;
        } else {
            alpha = expf(m - mnew);          // m = -inf -> 0
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int e = 0; e < 16; ++e) {
                    const float pv = expf(S[j][e] - mnew);
                    S[j][e] = pv;
                    psum += pv;
                }
        }
        l = l * alpha + psum;
        m = mnew;
#pragma unroll
        for (int d = 0; d < G::DT; ++d)
#pragma unroll
            for (int e = 0; e < 16; ++e) O[d][e] *= alpha;

        // ---- O^T += Vt_tile . P^T   (A = Vt rows (d), B = P from the S accumulators of this lane)
        if (sizeof(T) == 2) {
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) {
                const int j = ks >> 1, b0 = 8 * (ks & 1);
                // element jj of lane half h  <->  key 16*ks + 8*(jj>>2) + 4*h + (jj&3)   (regs b0 .. b0+7)
                const uint4 pf = make_uint4(pack_bf16x2(S[j][b0 + 0], S[j][b0 + 1]), pack_bf16x2(S[j][b0 + 2], S[j][b0 + 3]),
                                            pack_bf16x2(S[j][b0 + 4], S[j][b0 + 5]), pack_bf16x2(S[j][b0 + 6], S[j][b0 + 7]));
#pragma unroll
                for (int d = 0; d < G::DT; ++d) {
                    const int row = d * 32 + r;
                    const uint2 lo = *(const uint2*)(sV + v_off_bf16(row, 4 * ks + h));
                    const uint2 hi = *(const uint2*)(sV + v_off_bf16(row, 4 * ks + 2 + h));
                    O[d] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, make_uint4(lo.x, lo.y, hi.x, hi.y)),
                                                                   __builtin_bit_cast(bf16x8, pf), O[d], 0, 0, 0);
                }
            }
        } else {
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int g4 = 0; g4 < 4; ++g4) {
                    // MFMA with reg 4*g4+i: k = h  <->  key j*32 + 8*g4 + 4*h + i
#pragma unroll
                    for (int d = 0; d < G::DT; ++d) {
                        const uint4 a = *(const uint4*)(sV + v_off_f32(d * 32 + r, j * 8 + 2 * g4 + h));
                        O[d] = __builtin_amdgcn_mfma_f32_32x32x2f32(__uint_as_float(a.x), S[j][4 * g4 + 0], O[d], 0, 0, 0);
                        O[d] = __builtin_amdgcn_mfma_f32_32x32x2f32(__uint_as_float(a.y), S[j][4 * g4 + 1], O[d], 0, 0, 0);
                        O[d] = __builtin_amdgcn_mfma_f32_32x32x2f32(__uint_as_float(a.z), S[j][4 * g4 + 2], O[d], 0, 0, 0);
                        O[d] = __builtin_amdgcn_mfma_f32_32x32x2f32(__uint_as_float(a.w), S[j][4 * g4 + 3], O[d], 0, 0, 0);
                    }
                }
        }
        __syncthreads();
    }

    l += __shfl_xor(l, 32, 64);
    if (!valid) return;

    if (p.nsplit > 1) {
        // partial: [split][kh][rho][HD + 2] = O (unnormalised), m, l
        float* dst = p.part + (((size_t)blockIdx.z * p.n_kv_total + kh) * p.rows_pad + rho) * (HD + 2);
#pragma unroll
        for (int d = 0; d < G::DT; ++d)
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const int dd = d * 32 + acc_row(e, lane);
                if (dd < HD) dst[dd] = O[d][e];
            }
        if (h == 0) { dst[HD] = m; dst[HD + 1] = l; }
        return;
    }

    const float inv = l > 0.0f ? 1.0f / l : 0.0f;
    T* orow = (T*)p.O + (size_t)(frame * p.T + qi) * p.o_stride + (size_t)(head0 + qg) * HD;
#pragma unroll
    for (int d = 0; d < G::DT; ++d)
#pragma unroll
        for (int e4 = 0; e4 < 4; ++e4) {
            const int dd = d * 32 + acc_row(e4 * 4, lane);     // 4 consecutive d: regs 4*e4 .. 4*e4+3
            if (dd < HD) {
                if (sizeof(T) == 2) {
                    *(uint2*)(orow + dd) = make_uint2(pack_bf16x2(O[d][4 * e4] * inv, O[d][4 * e4 + 1] * inv),
                                                      pack_bf16x2(O[d][4 * e4 + 2] * inv, O[d][4 * e4 + 3] * inv));
                } else {
                    *(float4*)(orow + dd) = make_float4(O[d][4 * e4] * inv, O[d][4 * e4 + 1] * inv, O[d][4 * e4 + 2] * inv,
                                                        O[d][4 * e4 + 3] * inv);
                }
            }
        }
}

// merge split-KV partials: one wave per (kh, rho).  Pass 1: lane z owns split z (m_z, l_z) -> wave max / weights;
// pass 2: lane d owns output channels d and d + 64 and sums the weighted partial rows (independent loads).
template <typename T, int HD>
__global__ __launch_bounds__(64) void attn_combine_kernel(AttnArgs p) {
    __shared__ float wsh[64];
    const int rho = blockIdx.x, kh = blockIdx.y, lane = threadIdx.x;
    const int kv_len = p.dyn_kv_len ? *p.dyn_kv_len : p.kv_len;
    const int tiles = (kv_len + 63) >> 6;
    const int nsplit = min(min(p.nsplit, 64), (tiles + p.tiles_per_split - 1) / p.tiles_per_split);
    const size_t split_stride = (size_t)p.n_kv_total * p.rows_pad * (HD + 2);
    const float* base = p.part + ((size_t)kh * p.rows_pad + rho) * (HD + 2);
    float mz = -INFINITY, lz = 0.0f;
    if (lane < nsplit) { mz = base[lane * split_stride + HD]; lz = base[lane * split_stride + HD + 1]; }
    const float mstar = wave_max(mz);
    const float w = mz == -INFINITY ? 0.0f : expf(mz - mstar);
    const float lsum = wave_sum(w * lz);
    wsh[lane] = w;
    __syncthreads();
    float o0 = 0.0f, o1 = 0.0f;
    for (int z = 0; z < nsplit; ++z) {
        const float* pz = base + z * split_stride;
        const float wz = wsh[z];
        if (lane < HD) o0 += wz * pz[lane];
        if (lane + 64 < HD) o1 += wz * pz[lane + 64];
    }
    const float inv = lsum > 0.0f ? 1.0f / lsum : 0.0f;
    const int qi = rho / p.G, qg = rho - qi * p.G;
    const int frame = kh / p.hpf, head0 = (kh % p.hpf) * p.G;
    T* orow = (T*)p.O + (size_t)(frame * p.T + qi) * p.o_stride + (size_t)(head0 + qg) * HD;
    if (lane < HD) orow[lane] = from_f32<T>(o0 * inv);
    if (lane + 64 < HD) orow[lane + 64] = from_f32<T>(o1 * inv);
}

template <typename T, int HD, int WAVES> void launch_attn_t(hipStream_t s, const AttnArgs& a) {
    using G = AttnGeom<T, HD>;
    const int rows = a.T * a.G;
    dim3 grid((rows + WAVES * 32 - 1) / (WAVES * 32), a.n_kv_total, a.nsplit), block(WAVES * 64);
    const size_t lds = G::K_TILE_BYTES + G::V_TILE_BYTES;
    hipLaunchKernelGGL((attn_kernel<T, HD, WAVES>), grid, block, lds, s, a);
}

}  // namespace

template <typename T> void launch_attention(hipStream_t s, const AttnArgs& a, int head_dim, int waves) {
    if (head_dim == 128) {
        if (waves == 1) launch_attn_t<T, 128, 1>(s, a); else launch_attn_t<T, 128, 4>(s, a);
    } else if (head_dim == 72) {
        if (waves == 1) launch_attn_t<T, 72, 1>(s, a); else launch_attn_t<T, 72, 4>(s, a);
    }
}
template <typename T> void launch_attention_combine(hipStream_t s, const AttnArgs& a, int head_dim) {
    dim3 grid(a.T * a.G, a.n_kv_total), block(64);
    if (head_dim == 128) hipLaunchKernelGGL((attn_combine_kernel<T, 128>), grid, block, 0, s, a);
    else if (head_dim == 72) hipLaunchKernelGGL((attn_combine_kernel<T, 72>), grid, block, 0, s, a);
}
template <typename T, int HD, int WAVES> static void attn_attr() {
    using G = AttnGeom<T, HD>;
    (void)hipFuncSetAttribute((const void*)attn_kernel<T, HD, WAVES>, hipFuncAttributeMaxDynamicSharedMemorySize,
                              G::K_TILE_BYTES + G::V_TILE_BYTES);
}
void attention_init_attrs() {
    attn_attr<bf16, 128, 1>(); attn_attr<bf16, 128, 4>(); attn_attr<bf16, 72, 1>(); attn_attr<bf16, 72, 4>();
    attn_attr<float, 128, 1>(); attn_attr<float, 128, 4>(); attn_attr<float, 72, 1>(); attn_attr<float, 72, 4>();
}
template void launch_attention<bf16>(hipStream_t, const AttnArgs&, int, int);
template void launch_attention<float>(hipStream_t, const AttnArgs&, int, int);
template void launch_attention_combine<bf16>(hipStream_t, const AttnArgs&, int);
template void launch_attention_combine<float>(hipStream_t, const AttnArgs&, int);

}  // namespace svln
