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
// Split-KV (decode, steady prefill with few row blocks): grid.z splits write (m, l, unnormalised O)
// partials (rows padded to 16 bytes, float4 stores); attn_combine_kernel merges them.  One-frame ViT: the key split stays inside the
// workgroup (KG key groups of 2 waves, merged through LDS).
//
// Roofline: decode = HBM (KV bytes 2*nkv*hd*len*sizeof(T) per layer); prefill/ViT = MFMA by flops, but at these sizes
// (<= 1.2 waves per SIMD) measured ~3 us of exposed latency per 64-key tile: the softmax VALU work, the MFMA chains and the
// LDS round trips of a wave serialise.
#include <type_traits>

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
    static constexpr bool KSWZ = (HDC % 16) == 0;
    // LDS row of the K tile: 256-byte rows (head dim 128) are XOR-swizzled; other widths (head dim 72: 160 / 288 B) get one chunk of
    // padding instead, which makes the bank stride of consecutive rows 44 (bf16) / 76 (fp32) dwords: 16 rows -> 16 distinct 16-byte slots
    static constexpr int KROW = (KSWZ ? HDC : HDC + 1) * 16;
    static constexpr int K_TILE_BYTES = 64 * KROW;
    static constexpr int V_TILE_BYTES = VROWS * VC * 16;
};

SVLN_DEV float fast_exp2(float x) { return __builtin_amdgcn_exp2f(x); }      // v_exp_f32

// K tile: chunk c of key row `row`
template <typename G> SVLN_DEV int k_off(int row, int c) {
    return row * G::KROW + ((G::KSWZ ? (c ^ (row & 15)) : c) << 4);
}
// Vt tile, float: 16 chunks (256 B) per row, ds_read_b128 -> XOR chunk with row & 15
SVLN_DEV int v_off_f32(int row, int c) { return row * 256 + ((c ^ (row & 15)) << 4); }
// Vt tile, bf16: 16 units of 8 B (128 B) per row, ds_read_b64 by 32-lane halves -> XOR unit with (row>>1)&15
SVLN_DEV int v_off_bf16(int row, int unit) { return row * 128 + ((unit ^ ((row >> 1) & 15)) << 3); }

// 8-byte LDS read as inline asm: left to itself hipcc pairs the Vt fragment reads of two output tiles into ds_read2st64_b64, which is
// banked modulo 32 in 16-lane groups (and moves half the bytes per LDS cycle): under that mapping the Vt swizzle below, built for
// ds_read_b64's 64 banks / 32-lane halves, is 2-way conflicted (measured: SQ_LDS_BANK_CONFLICT = 38-43 % of SQ_LDS_IDX_ACTIVE).
template <int OFF> SVLN_DEV void lds_read_b64(uint2& v, unsigned addr) { asm volatile("ds_read_b64 %0, %1 offset:%2" : "=v"(v) : "v"(addr), "n"(OFF)); }
template <int N> SVLN_DEV void lds_wait(uint2& a, uint2& b) { asm volatile("s_waitcnt lgkmcnt(%2)" : "+v"(a), "+v"(b) : "n"(N)); }
// compile-time loop with the index as a constant (immediate offsets / counted waits in inline asm)
template <int I, int N, typename F> SVLN_DEV void static_for(F&& f) {
    if constexpr (I < N) { f(std::integral_constant<int, I>{}); static_for<I + 1, N>(f); }
}

// KG > 1 (one-frame ViT): the workgroup is KG key groups of WAVES waves; group kg walks key tiles kg, kg + KG, ... of the SAME 32 * WAVES query
// rows through its own K / Vt tile pair in LDS, so KG tiles are in flight per workgroup at once, and the groups' (m, l, O) are merged through
// LDS at the end: the split-KV partials never leave the CU and there is no combine launch.
template <typename T, int HD, int WAVES, int KG = 1, int PF = 1>
__global__ __launch_bounds__(WAVES * 64 * KG, (KG == 1 && WAVES == 4 && sizeof(T) == 2) ? 2 : 1) void attn_kernel(AttnArgs p) {
    using G = AttnGeom<T, HD>;
    constexpr int NT = WAVES * 64;              // threads of one key group: they stage its tiles
    constexpr int EPC = G::EPC;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int kg = KG == 1 ? 0 : (int)threadIdx.x / NT;
    char* sK = smem + kg * (G::K_TILE_BYTES + G::V_TILE_BYTES);
    char* sV = sK + G::K_TILE_BYTES;

    const int tid = KG == 1 ? (int)threadIdx.x : (int)threadIdx.x - kg * NT, lane = tid & 63, wave = tid >> 6;
    const int r = lane & 31, h = lane >> 5;
    const int kh = blockIdx.y;
    const int frame = kh / p.hpf, head0 = (kh % p.hpf) * p.G;
    // first page id is independent of kv_len: fetch both scalars together (one round trip instead of two)
    const int kt0 = blockIdx.z * p.tiles_per_split + kg;
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
    const int kt_begin = blockIdx.z * p.tiles_per_split + kg;
    const int kt_end = min(tiles, (int)blockIdx.z * p.tiles_per_split + p.tiles_per_split);
    const int n_iter = (kt_end - (int)blockIdx.z * p.tiles_per_split + KG - 1) / KG;       // uniform over the workgroup (barriers)

    const float scale2 = p.scale * 1.4426950408889634f;      // scores are kept in the log2 domain (m too)
    f32x16 O[G::DT];
#pragma unroll
    for (int d = 0; d < G::DT; ++d)
#pragma unroll
        for (int e = 0; e < 16; ++e) O[d][e] = 0.0f;
    float m = -INFINITY, l = 0.0f;

    const size_t k_page_stride = (size_t)p.n_kv_total * 64 * G::HDP * sizeof(T);
    const size_t v_page_stride = (size_t)p.n_kv_total * G::VROWS * 64 * sizeof(T);

    // 4-wave workgroups (prefill / ViT): the next tile's K / Vt loads are issued into registers right after the barrier
    // that publishes the current tile and stay in flight under its MFMAs (one tile of HBM/L2 latency hidden per tile).
    // (PF: 0 = no register prefetch, 1 = K and Vt, 2 = K only -- the head_dim 128 key groups: both tiles would need 64 registers per lane)
    constexpr bool PFK = WAVES == 4 || (KG > 1 && PF >= 1), PFV = WAVES == 4 || (KG > 1 && PF == 1);
    constexpr int KTOT = 64 * G::HDC, VTOT = G::VROWS * G::VC, B = WAVES == 1 ? 4 : 8;
    constexpr int KL = (KTOT + NT - 1) / NT, VL = (VTOT + NT - 1) / NT;
    uint4 pk[PFK ? KL : 1], pv[PFV ? VL : 1];
    auto tile_ptrs = [&](int kt, const char*& gk, const char*& gv) {
        const int pg = kt == kt0 ? page0 : (p.page_table ? p.page_table[kt] : kt);
        gk = (const char*)p.Kpool + (size_t)pg * k_page_stride + (size_t)kh * 64 * G::HDP * sizeof(T);
        gv = (const char*)p.Vpool + (size_t)pg * v_page_stride + (size_t)kh * G::VROWS * 64 * sizeof(T);
    };
    auto put_k = [&](int q, const uint4& v) {
        const int row = q / G::HDC, c = q - row * G::HDC;
        *(uint4*)(sK + k_off<G>(row, c)) = v;
    };
    auto put_v = [&](int q, const uint4& v) {
        const int row = q / G::VC, c = q - row * G::VC;
        if (sizeof(T) == 4) {
            *(uint4*)(sV + v_off_f32(row, c)) = v;
        } else {
            *(uint2*)(sV + v_off_bf16(row, 2 * c)) = make_uint2(v.x, v.y);
            *(uint2*)(sV + v_off_bf16(row, 2 * c + 1)) = make_uint2(v.z, v.w);
        }
    };
    auto load_regs = [&](int kt) {
        const char *gk, *gv;
        tile_ptrs(kt, gk, gv);
        if (PFK) {
#pragma unroll
            for (int u = 0; u < KL; ++u) {
                const int q = tid + u * NT;
                pk[u] = q < KTOT ? *(const uint4*)(gk + (size_t)q * 16) : zero_chunk();
            }
        }
        if (PFV) {
#pragma unroll
            for (int u = 0; u < VL; ++u) {
                const int q = tid + u * NT;
                pv[u] = q < VTOT ? *(const uint4*)(gv + (size_t)q * 16) : zero_chunk();
            }
        }
    };
    if (PFK && kt_begin < kt_end) load_regs(kt_begin);

    for (int it = 0; it < n_iter; ++it) {
        const int kt = kt_begin + it * KG;
        const bool act = KG == 1 || kt < kt_end;        // (a key group past its last tile only keeps the barriers)
        // ---- stage K and Vt tiles: from the prefetch registers, or coalesced 16-byte loads issued in batches (all in flight before the
        //      first dependent LDS write); swizzled LDS writes
        if (act) {
            const char *gK, *gV;
            if (!PFK || !PFV) tile_ptrs(kt, gK, gV);
            if (PFK) {
#pragma unroll
                for (int u = 0; u < KL; ++u) {
                    const int q = tid + u * NT;
                    if (q < KTOT) put_k(q, pk[u]);
                }
            } else {
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
                        if (b0 + u < KL && q < KTOT) put_k(q, t[u]);
                    }
                }
            }
            if (PFV) {
#pragma unroll
                for (int u = 0; u < VL; ++u) {
                    const int q = tid + u * NT;
                    if (q < VTOT) put_v(q, pv[u]);
                }
            } else {
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
                        if (b0 + u < VL && q < VTOT) put_v(q, t[u]);
                    }
                }
            }
        }
        __syncthreads();
        if (PFK && kt + KG < kt_end) load_regs(kt + KG);
        if (act) {

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

        // ---- mask, online softmax (per lane = per query row; other half of the keys is in lane ^ 32).  Scores stay RAW in S: the
        // scale (log2 domain) is folded into one fma per element, exp2(S * scale2 - m).  Only a tile that reaches past kv_len or past
        // the causal diagonal of some row of this WORKGROUP is masked (uniform test); O is rescaled only when some lane's max moved.
        const int tile_last_key = kt * 64 + 63;
        const bool need_mask = tile_last_key >= kv_len || (p.causal && tile_last_key > P + (int)(blockIdx.x * WAVES * 32) / p.G);
        if (need_mask) {
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int e = 0; e < 16; ++e) {
                    const int key = kt * 64 + j * 32 + acc_row(e, lane);
                    if (!(key < kv_len && (!p.causal || key <= qpos))) S[j][e] = -INFINITY;
                }
        }
        float mloc = -INFINITY;
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) mloc = fmaxf(mloc, S[j][e]);
        mloc = fmaxf(mloc, __shfl_xor(mloc, 32, 64)) * scale2;             // scale2 > 0: max commutes with the scaling
        const float mnew = fmaxf(m, mloc);
        float alpha = 1.0f, psum = 0.0f;
        if (mnew == -INFINITY) {
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int e = 0; e < 16; ++e) S[j][e] = 0.0f;
        } else {
            alpha = fast_exp2(m - mnew);     // m = -inf -> 0
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int e = 0; e < 16; ++e) {
                    const float pv = fast_exp2(fmaf(S[j][e], scale2, -mnew));       // masked: -inf -> 0
                    S[j][e] = pv;
                    psum += pv;
                }
        }
        l = l * alpha + psum;
        m = mnew;
        if (__builtin_amdgcn_ballot_w64(alpha != 1.0f) != 0) {
#pragma unroll
            for (int d = 0; d < G::DT; ++d)
#pragma unroll
                for (int e = 0; e < 16; ++e) O[d][e] *= alpha;
        }

        // ---- O^T += Vt_tile . P^T   (A = Vt rows (d), B = P from the S accumulators of this lane)
        if (sizeof(T) == 2) {
            // row d*32 + r of the Vt tile: (row >> 1) & 15 does not depend on d, so the DT fragment pairs of a k-step share two address
            // registers (lo / hi unit) and differ by an immediate offset of 32 rows = 4 KiB
            const unsigned sv0 = (unsigned)(size_t)(__attribute__((address_space(3))) char*)sV + r * 128;
            const int xr = (r >> 1) & 15;
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) {
                const int j = ks >> 1, b0 = 8 * (ks & 1);
                const unsigned alo = sv0 + (((4 * ks + h) ^ xr) << 3), ahi = sv0 + (((4 * ks + 2 + h) ^ xr) << 3);
                uint2 vlo[G::DT], vhi[G::DT];
                static_for<0, G::DT>([&](auto dc) {
                    constexpr int d = decltype(dc)::value;
                    lds_read_b64<d * 4096>(vlo[d], alo);
                    lds_read_b64<d * 4096>(vhi[d], ahi);
                });
                // element jj of lane half h  <->  key 16*ks + 8*(jj>>2) + 4*h + (jj&3)   (regs b0 .. b0+7)
                const uint4 pf = make_uint4(pack_bf16x2(S[j][b0 + 0], S[j][b0 + 1]), pack_bf16x2(S[j][b0 + 2], S[j][b0 + 3]),
                                            pack_bf16x2(S[j][b0 + 4], S[j][b0 + 5]), pack_bf16x2(S[j][b0 + 6], S[j][b0 + 7]));
                static_for<0, G::DT>([&](auto dc) {
                    constexpr int d = decltype(dc)::value;
                    lds_wait<2 * (G::DT - 1 - d)>(vlo[d], vhi[d]);          // LDS reads return in order: only younger pairs remain
                    O[d] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, make_uint4(vlo[d].x, vlo[d].y, vhi[d].x, vhi[d].y)),
                                                                   __builtin_bit_cast(bf16x8, pf), O[d], 0, 0, 0);
                });
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
        }
        __syncthreads();
    }

    l += __shfl_xor(l, 32, 64);
    if constexpr (KG > 1) {
        // ---- merge of the key groups (the loop ended on a barrier: the tiles are dead).  (m, l) of every group and row -> each lane scales its
        // O by exp2(m - M) -> fp32 rows in LDS -> all threads sum the KG rows, divide by L and store 4 channels each (or, under a grid-level
        // key split on top, write the merged (O, M, L) partial row of this split for attn_combine_kernel).
        constexpr int ROWS = WAVES * 32, OP = HD + ATTN_PART_PAD;
        float* sML = (float*)smem;                                  // [KG][ROWS] x (m, l)
        float* sFin = (float*)(smem + KG * ROWS * 8);               // [ROWS] x (M, L)
        float* sO = (float*)(smem + (KG + 1) * ROWS * 8);           // [KG][ROWS][OP]
        const int rowl = wave * 32 + r;
        if (h == 0) *(float2*)(sML + (kg * ROWS + rowl) * 2) = make_float2(m, l);
        __syncthreads();
        float M = -INFINITY;
#pragma unroll
        for (int g = 0; g < KG; ++g) M = fmaxf(M, sML[(g * ROWS + rowl) * 2]);
        float L = 0.0f;
#pragma unroll
        for (int g = 0; g < KG; ++g) {
            const float2 ml = *(const float2*)(sML + (g * ROWS + rowl) * 2);
            L += ml.x == -INFINITY ? 0.0f : ml.y * fast_exp2(ml.x - M);
        }
        if (kg == 0 && h == 0) *(float2*)(sFin + rowl * 2) = make_float2(M, L);
        const float f = m == -INFINITY ? 0.0f : fast_exp2(m - M);
        float* od = sO + (size_t)(kg * ROWS + rowl) * OP;
#pragma unroll
        for (int d = 0; d < G::DT; ++d)
#pragma unroll
            for (int e4 = 0; e4 < 4; ++e4) {
                const int dd = d * 32 + acc_row(e4 * 4, lane);
                if (dd < HD) *(float4*)(od + dd) = make_float4(O[d][4 * e4] * f, O[d][4 * e4 + 1] * f, O[d][4 * e4 + 2] * f, O[d][4 * e4 + 3] * f);
            }
        __syncthreads();
        for (int q = threadIdx.x; q < ROWS * (HD / 4); q += NT * KG) {
            const int row = q / (HD / 4), c4 = q - row * (HD / 4);
            const int rho_o = blockIdx.x * ROWS + row;
            if (rho_o >= rows_total) continue;
            float4 acc = *(const float4*)(sO + (size_t)row * OP + c4 * 4);
#pragma unroll
            for (int g = 1; g < KG; ++g) {
                const float4 v = *(const float4*)(sO + (size_t)(g * ROWS + row) * OP + c4 * 4);
                acc.x += v.x; acc.y += v.y; acc.z += v.z; acc.w += v.w;
            }
            const float2 fin = *(const float2*)(sFin + row * 2);
            if (p.nsplit > 1) {
                float* dst = p.part + (((size_t)blockIdx.z * p.n_kv_total + kh) * p.rows_pad + rho_o) * OP;
                *(float4*)(dst + c4 * 4) = acc;
                if (c4 == 0) { dst[HD] = fin.x; dst[HD + 1] = fin.y; }
                continue;
            }
            const float inv = fin.y > 0.0f ? 1.0f / fin.y : 0.0f;
            const int qi_o = rho_o / p.G, qg_o = rho_o - qi_o * p.G;
            T* orow = (T*)p.O + (size_t)(frame * p.T + qi_o) * p.o_stride + (size_t)(head0 + qg_o) * HD + c4 * 4;
            if (sizeof(T) == 2) *(uint2*)orow = make_uint2(pack_bf16x2(acc.x * inv, acc.y * inv), pack_bf16x2(acc.z * inv, acc.w * inv));
            else *(float4*)orow = make_float4(acc.x * inv, acc.y * inv, acc.z * inv, acc.w * inv);
        }
        return;
    }
    if (!valid) return;

    if (p.nsplit > 1) {
        // partial: [split][kh][rho][HD + 4] = O (unnormalised), m, l, pad; 16-byte stores of 4 consecutive channels
        float* dst = p.part + (((size_t)blockIdx.z * p.n_kv_total + kh) * p.rows_pad + rho) * (HD + ATTN_PART_PAD);
#pragma unroll
        for (int d = 0; d < G::DT; ++d)
#pragma unroll
            for (int e4 = 0; e4 < 4; ++e4) {
                const int dd = d * 32 + acc_row(e4 * 4, lane);
                if (dd < HD) *(float4*)(dst + dd) = make_float4(O[d][4 * e4], O[d][4 * e4 + 1], O[d][4 * e4 + 2], O[d][4 * e4 + 3]);
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

// Decode step attention (one query position, G <= 32 q-head rows per kv head, head_dim 128): workgroup = (kv head, key
// split) with 4 waves.  All 256 threads put the split's K / Vt page loads in flight at once (8 x 16 B per thread),
// every wave builds the roped Q fragments and S^T = K.Q^T for the page (16 MFMAs, redundant per wave, trivial), then
// wave w owns output channels [32w, 32w+32) of O^T = Vt.P^T.  RoPE(q), RoPE(k) + the KV append of the token being
// decoded are fused (the workgroup that owns its page patches the LDS tile and writes the pools).
template <typename T>
__global__ __launch_bounds__(256) void attn_decode_kernel(AttnArgs p) {
    using G = AttnGeom<T, 128>;
    constexpr int EPC = G::EPC, NT = 256;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    if (p.skip && *p.skip) return;
    char* sK = smem;
    char* sV = smem + G::K_TILE_BYTES;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int r = lane & 31, h = lane >> 5;
    const int kh = blockIdx.x, z = blockIdx.y, env = blockIdx.z;
    const int kt0 = z * p.tiles_per_split;
    const int* page_table = p.slots ? p.slots[env].page_table : p.page_table;
    const int page0 = page_table[kt0];
    const int pos = p.slots ? p.slots[env].pos : *p.dyn_pos;
    const T* qkv_row = (const T*)p.Q + (size_t)env * p.q_stride;
    const int kv_len = pos + 1;
    const int tiles = (kv_len + 63) >> 6;
    if (kt0 >= tiles) return;                                   // combine only reads splits below ceil(tiles / tiles_per_split)
    const int kt_end = min(tiles, kt0 + p.tiles_per_split);
    const size_t k_page_stride = (size_t)p.n_kv_total * 64 * 128 * sizeof(T);
    const bool valid = r < p.G;

    // roped Q fragments of row rho = r (q head kh*G + r), identical in every wave
    uint4 qf[G::HDC / 2];
    {
        const T* qrow = qkv_row + (size_t)(kh * p.G + (valid ? r : 0)) * 128;
#pragma unroll
        for (int s = 0; s < G::HDC / 2; ++s) qf[s] = valid ? *(const uint4*)(qrow + (2 * s + h) * EPC) : zero_chunk();
        const float* tab = p.rope_tab + (size_t)pos * 128;
#pragma unroll
        for (int s = 0; s < G::HDC / 4; ++s) {
            const int d0 = (2 * s + h) * EPC;
            float x1[EPC], x2[EPC], o1[EPC], o2[EPC];
            chunk_to_f32<T>(qf[s], x1);
            chunk_to_f32<T>(qf[s + G::HDC / 4], x2);
#pragma unroll
            for (int e = 0; e < EPC; ++e) {
                const float c = tab[d0 + e], sn = tab[64 + d0 + e];
                o1[e] = x1[e] * c - x2[e] * sn;
                o2[e] = x2[e] * c + x1[e] * sn;
            }
            qf[s] = f32_to_chunk<T>(o1);
            qf[s + G::HDC / 4] = f32_to_chunk<T>(o2);
        }
    }

    const float scale2 = p.scale * 1.4426950408889634f;
    f32x16 O;
#pragma unroll
    for (int e = 0; e < 16; ++e) O[e] = 0.0f;
    float m = -INFINITY, l = 0.0f;

    for (int kt = kt0; kt < kt_end; ++kt) {
        const int page = kt == kt0 ? page0 : page_table[kt];
        const char* gK = (const char*)p.Kpool + (size_t)page * k_page_stride + (size_t)kh * 64 * 128 * sizeof(T);
        const char* gV = (const char*)p.Vpool + (size_t)page * k_page_stride + (size_t)kh * 128 * 64 * sizeof(T);
        constexpr int KL = 64 * G::HDC / NT, VL = 128 * G::VC / NT;
        uint4 tk[KL], tv[VL];
#pragma unroll
        for (int u = 0; u < KL; ++u) tk[u] = *(const uint4*)(gK + (size_t)(tid + u * NT) * 16);
#pragma unroll
        for (int u = 0; u < VL; ++u) tv[u] = *(const uint4*)(gV + (size_t)(tid + u * NT) * 16);
#pragma unroll
        for (int u = 0; u < KL; ++u) {
            const int q = tid + u * NT, row = q / G::HDC, c = q - row * G::HDC;
            *(uint4*)(sK + k_off<G>(row, c)) = tk[u];
        }
#pragma unroll
        for (int u = 0; u < VL; ++u) {
            const int q = tid + u * NT, row = q / G::VC, c = q - row * G::VC;
            if (sizeof(T) == 4) {
                *(uint4*)(sV + v_off_f32(row, c)) = tv[u];
            } else {
                *(uint2*)(sV + v_off_bf16(row, 2 * c)) = make_uint2(tv[u].x, tv[u].y);
                *(uint2*)(sV + v_off_bf16(row, 2 * c + 1)) = make_uint2(tv[u].z, tv[u].w);
            }
        }
        if (kt == (pos >> 6)) {
            __syncthreads();
            if (tid < 64) {
                const int off = pos & 63, d = tid;
                const T* krow = qkv_row + (size_t)(p.nq_heads + kh) * 128;
                const T* vrow = qkv_row + (size_t)(p.nq_heads + p.n_kv_total + kh) * 128;
                const float* tab = p.rope_tab + (size_t)pos * 128;
                const float k1 = to_f32(krow[d]), k2 = to_f32(krow[d + 64]);
                const float c = tab[d], sn = tab[64 + d];
                const T ko1 = from_f32<T>(k1 * c - k2 * sn), ko2 = from_f32<T>(k2 * c + k1 * sn);
                const T v1 = vrow[d], v2 = vrow[d + 64];
                T* gk = (T*)(const_cast<char*>(gK)) + (size_t)off * 128;
                T* gv = (T*)(const_cast<char*>(gV));
                gk[d] = ko1; gk[d + 64] = ko2;
                gv[(size_t)d * 64 + off] = v1; gv[(size_t)(d + 64) * 64 + off] = v2;
                *(T*)(sK + k_off<G>(off, d / EPC) + (d % EPC) * sizeof(T)) = ko1;
                *(T*)(sK + k_off<G>(off, (d + 64) / EPC) + ((d + 64) % EPC) * sizeof(T)) = ko2;
                if (sizeof(T) == 4) {
                    *(T*)(sV + v_off_f32(d, off >> 2) + (off & 3) * 4) = v1;
                    *(T*)(sV + v_off_f32(d + 64, off >> 2) + (off & 3) * 4) = v2;
                } else {
                    *(T*)(sV + v_off_bf16(d, off >> 2) + (off & 3) * 2) = v1;
                    *(T*)(sV + v_off_bf16(d + 64, off >> 2) + (off & 3) * 2) = v2;
                }
            }
        }
        __syncthreads();

        f32x16 S[2];
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) S[j][e] = 0.0f;
#pragma unroll
        for (int s = 0; s < G::HDC / 2; ++s)
#pragma unroll
            for (int j = 0; j < 2; ++j) mma_chunk<T>(*(const uint4*)(sK + k_off<G>(j * 32 + r, 2 * s + h)), qf[s], S[j]);
        float mloc = -INFINITY;
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const int key = kt * 64 + j * 32 + acc_row(e, lane);
                const float sv = key < kv_len ? S[j][e] * scale2 : -INFINITY;
                S[j][e] = sv;
                mloc = fmaxf(mloc, sv);
            }
        mloc = fmaxf(mloc, __shfl_xor(mloc, 32, 64));
        const float mnew = fmaxf(m, mloc);          // the first key of every processed tile is valid -> finite
        const float alpha = fast_exp2(m - mnew);
        float psum = 0.0f;
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const float pv = fast_exp2(S[j][e] - mnew);
                S[j][e] = pv;
                psum += pv;
            }
        l = l * alpha + psum;
        m = mnew;
#pragma unroll
        for (int e = 0; e < 16; ++e) O[e] *= alpha;
        const int row = wave * 32 + r;                // this wave's 32 output channels
        if (sizeof(T) == 2) {
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) {
                const int j = ks >> 1, b0 = 8 * (ks & 1);
                const uint4 pf = make_uint4(pack_bf16x2(S[j][b0 + 0], S[j][b0 + 1]), pack_bf16x2(S[j][b0 + 2], S[j][b0 + 3]),
                                            pack_bf16x2(S[j][b0 + 4], S[j][b0 + 5]), pack_bf16x2(S[j][b0 + 6], S[j][b0 + 7]));
                const uint2 lo = *(const uint2*)(sV + v_off_bf16(row, 4 * ks + h));
                const uint2 hi = *(const uint2*)(sV + v_off_bf16(row, 4 * ks + 2 + h));
                O = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, make_uint4(lo.x, lo.y, hi.x, hi.y)),
                                                            __builtin_bit_cast(bf16x8, pf), O, 0, 0, 0);
            }
        } else {
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int g4 = 0; g4 < 4; ++g4) {
                    const uint4 a = *(const uint4*)(sV + v_off_f32(row, j * 8 + 2 * g4 + h));
                    O = __builtin_amdgcn_mfma_f32_32x32x2f32(__uint_as_float(a.x), S[j][4 * g4 + 0], O, 0, 0, 0);
                    O = __builtin_amdgcn_mfma_f32_32x32x2f32(__uint_as_float(a.y), S[j][4 * g4 + 1], O, 0, 0, 0);
                    O = __builtin_amdgcn_mfma_f32_32x32x2f32(__uint_as_float(a.z), S[j][4 * g4 + 2], O, 0, 0, 0);
                    O = __builtin_amdgcn_mfma_f32_32x32x2f32(__uint_as_float(a.w), S[j][4 * g4 + 3], O, 0, 0, 0);
                }
        }
        __syncthreads();
    }
    l += __shfl_xor(l, 32, 64);
    if (!valid) return;
    float* dst = p.part + (size_t)env * p.part_bstride + (((size_t)z * p.n_kv_total + kh) * p.rows_pad + r) * (128 + ATTN_PART_PAD);
#pragma unroll
    for (int e4 = 0; e4 < 4; ++e4)
        *(float4*)(dst + wave * 32 + acc_row(e4 * 4, lane)) = make_float4(O[4 * e4], O[4 * e4 + 1], O[4 * e4 + 2], O[4 * e4 + 3]);
    if (wave == 0 && h == 0) { dst[128] = m; dst[129] = l; }
}

// merge split-KV partials: one wave per (kh, rho).  Pass 1: lane z owns split z (m_z, l_z) -> wave max / weights;
// pass 2: lane d owns output channels d and d + 64 and sums the weighted partial rows (independent loads).
// A wave per row; four rows per workgroup when there are many rows (prefill / ViT: thousands of single-wave workgroups are dispatch-bound),
// one row per workgroup for the few rows of a decode step (spread over as many CUs as possible: 5.8 us against 9.4 with four per workgroup).
template <typename T, int HD>
__global__ __launch_bounds__(256) void attn_combine_kernel(AttnArgs p) {
    __shared__ float wsh_all[4][64];
    if (p.skip && *p.skip) return;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    float* wsh = wsh_all[wave];         // written and read by this wave only
    const int rho = blockIdx.x * (blockDim.x >> 6) + wave, kh = blockIdx.y, env = blockIdx.z;
    if (rho >= p.T * p.G) return;
    const int kv_len = p.slots ? p.slots[env].pos + 1 : (p.dyn_kv_len ? *p.dyn_kv_len : p.kv_len);
    const int tiles = (kv_len + 63) >> 6;
    const int nsplit = min(min(p.nsplit, 64), (tiles + p.tiles_per_split - 1) / p.tiles_per_split);
    const size_t split_stride = (size_t)p.n_kv_total * p.rows_pad * (HD + ATTN_PART_PAD);
    const float* base = p.part + (size_t)env * p.part_bstride + ((size_t)kh * p.rows_pad + rho) * (HD + ATTN_PART_PAD);
    float mz = -INFINITY, lz = 0.0f;
    if (lane < nsplit) { mz = base[lane * split_stride + HD]; lz = base[lane * split_stride + HD + 1]; }
    const float mstar = wave_max(mz);
    const float w = mz == -INFINITY ? 0.0f : fast_exp2(mz - mstar);      // m values are in the log2 domain
    const float lsum = wave_sum(w * lz);
    wsh[lane] = w;                      // (LDS executes a wave's accesses in order: no workgroup barrier, and the rows that returned above never reach one)
    __builtin_amdgcn_wave_barrier();
    float o0 = 0.0f, o1 = 0.0f;
    for (int z = 0; z < nsplit; ++z) {
        const float* pz = base + z * split_stride;
        const float wz = wsh[z];        // (a v_readlane per split instead measured 9.4 us against 5.8 on the decode merge: it serialises the loads)
        if (lane < HD) o0 += wz * pz[lane];
        if (lane + 64 < HD) o1 += wz * pz[lane + 64];
    }
    const float inv = lsum > 0.0f ? 1.0f / lsum : 0.0f;
    const int qi = rho / p.G, qg = rho - qi * p.G;
    const int frame = kh / p.hpf, head0 = (kh % p.hpf) * p.G;
    T* orow = (T*)p.O + (size_t)(frame * p.T + qi + env) * p.o_stride + (size_t)(head0 + qg) * HD;     // env > 0 only when T == 1
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
// key groups inside the workgroup (a.key_groups > 1): 64 query rows x KG groups, one pass over all key tiles, no partials.
// One ViT frame, bf16, measured (tools/kbench.py attn under rocprofv3): split-KV 3 + combine 15.6 + 7.4 us; 4 groups with the register
// prefetch 14.3; 4 groups without it 19.3; 6 groups without 20.1; 6 groups with it 34.8 (12 waves leave 168 VGPRs: 72 spilled).
// fp32 (the verification engine): 3 groups, no prefetch (its tiles are twice the bytes: LDS and registers).
template <typename T> struct VitGroups { static constexpr int KG = sizeof(T) == 2 ? 4 : 3; static constexpr int PF = sizeof(T) == 2 ? 1 : 0; };
template <typename T, int HD, int KG_> struct GroupGeom {
    using G = AttnGeom<T, HD>;
    static constexpr int WAVES = 2, ROWS = WAVES * 32;
    static constexpr int KG = KG_;
    static constexpr size_t TILES = (size_t)KG * (G::K_TILE_BYTES + G::V_TILE_BYTES);
    static constexpr size_t MERGE = (size_t)(KG + 1) * ROWS * 8 + (size_t)KG * ROWS * (HD + ATTN_PART_PAD) * 4;
    static constexpr size_t LDS = TILES > MERGE ? TILES : MERGE;
};
template <typename T, int HD, int KG, int PF> void launch_attn_groups(hipStream_t s, const AttnArgs& a) {
    using GG = GroupGeom<T, HD, KG>;
    const int rows = a.T * a.G;
    dim3 grid((rows + GG::ROWS - 1) / GG::ROWS, a.n_kv_total, a.nsplit), block(GG::WAVES * 64 * GG::KG);
    hipLaunchKernelGGL((attn_kernel<T, HD, GG::WAVES, GG::KG, PF>), grid, block, GG::LDS, s, a);
}

}  // namespace

template <typename T> void launch_attention(hipStream_t s, const AttnArgs& a, int head_dim, int waves) {
    if (head_dim == 128 && waves == 1 && a.fuse_rope_append && a.T == 1) {
        using G = AttnGeom<T, 128>;
        hipLaunchKernelGGL((attn_decode_kernel<T>), dim3(a.n_kv_total, a.nsplit, a.batch > 0 ? a.batch : 1), dim3(256), G::K_TILE_BYTES + G::V_TILE_BYTES, s, a);
        return;
    }
    if (a.key_groups > 1) {
        if (head_dim != 72 || a.key_groups != attn_key_groups<T>(head_dim))
            throw std::runtime_error("attention: key groups are built for head_dim 72 with key_groups == attn_key_groups<T>(72)");
        launch_attn_groups<T, 72, VitGroups<T>::KG, VitGroups<T>::PF>(s, a);
        return;
    }
    if (head_dim == 128) {
        if (waves == 1) launch_attn_t<T, 128, 1>(s, a); else launch_attn_t<T, 128, 4>(s, a);
    } else if (head_dim == 72) {
        if (waves == 1) launch_attn_t<T, 72, 1>(s, a); else launch_attn_t<T, 72, 4>(s, a);
    }
}
template <typename T> void launch_attention_combine(hipStream_t s, const AttnArgs& a, int head_dim) {
    const int rows = a.T * a.G, rpw = rows >= 256 ? 4 : 1;
    dim3 grid((rows + rpw - 1) / rpw, a.n_kv_total, a.batch > 0 ? a.batch : 1), block(64 * rpw);
    if (head_dim == 128) hipLaunchKernelGGL((attn_combine_kernel<T, 128>), grid, block, 0, s, a);
    else if (head_dim == 72) hipLaunchKernelGGL((attn_combine_kernel<T, 72>), grid, block, 0, s, a);
}
template <typename T, int HD, int WAVES> static void attn_attr() {
    using G = AttnGeom<T, HD>;
    set_max_lds((const void*)attn_kernel<T, HD, WAVES>, G::K_TILE_BYTES + G::V_TILE_BYTES);
}
void attention_init_attrs() {
    set_max_lds((const void*)attn_decode_kernel<float>, AttnGeom<float, 128>::K_TILE_BYTES + AttnGeom<float, 128>::V_TILE_BYTES);
    attn_attr<bf16, 128, 1>(); attn_attr<bf16, 128, 4>(); attn_attr<bf16, 72, 1>(); attn_attr<bf16, 72, 4>();
    attn_attr<float, 128, 1>(); attn_attr<float, 128, 4>(); attn_attr<float, 72, 1>(); attn_attr<float, 72, 4>();
    set_max_lds((const void*)attn_kernel<bf16, 72, 2, VitGroups<bf16>::KG, VitGroups<bf16>::PF>, GroupGeom<bf16, 72, VitGroups<bf16>::KG>::LDS);
    set_max_lds((const void*)attn_kernel<float, 72, 2, VitGroups<float>::KG, VitGroups<float>::PF>, GroupGeom<float, 72, VitGroups<float>::KG>::LDS);
}
template <typename T> int attn_key_groups(int head_dim) { return head_dim == 72 ? VitGroups<T>::KG : 1; }
template int attn_key_groups<bf16>(int);
template int attn_key_groups<float>(int);
template void launch_attention<bf16>(hipStream_t, const AttnArgs&, int, int);
template void launch_attention<float>(hipStream_t, const AttnArgs&, int, int);
template void launch_attention_combine<bf16>(hipStream_t, const AttnArgs&, int);
template void launch_attention_combine<float>(hipStream_t, const AttnArgs&, int);

}  // namespace svln
