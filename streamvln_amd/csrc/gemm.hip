// MFMA GEMM  C[M,N] = epi(A[M,K] . W[N,K]^T + bias) + res   for the dense QKV / MLP / projector /
// patch-embed products (prefill and vision; decode uses gemv.hip).
//
// gemm_glds_kernel: MFMA 32x32 accumulators, MI x NJ of them per wave; both operands staged straight into LDS by LDS-DMA
// (global_load_lds_dwordx4) through a ring of stage buffers (one stage = 128 B of K per tile row), fragment reads by
// inline-asm ds_read_b128, one s_barrier per stage.  Tile configurations, picked in launch_epi from M, N, K:
//   * Cfg256  256x128, 8 waves (64x64 wave tiles), 3-deep ring, split-K over the grid  -- M <= 256 (steady prefill T ~ 212,
//     one-frame ViT 729 rows with few column tiles): the whole M extent sits in one row tile so every weight byte is
//     streamed from HBM once, and K is split so that about one workgroup per CU exists.  Split-K partials go to fp32 slabs
//     [split][M][N]; a second launch sums them and applies the epilogue (splitk_epilogue_kernel), or -- for o_proj /
//     down_proj / SigLIP out_proj / fc2 -- also the following RMSNorm / LayerNorm (splitk_rownorm_kernel).
//   * Cfg128 / Cfg128L  128x128, 4 waves, 2-deep ring, two workgroups per CU            -- mid-size M; Cfg128K2: the same tile with two
//     K-groups of 4 waves inside the workgroup (one-frame ViT q|k|v / fc1, projector)
//   * Cfg256N64  256x64, 8 waves stacked along M, 3-deep ring, split-K                  -- q|k|v and o_proj of a steady prefill
//   * CfgBig  256x256, 8 waves (128x64 wave tiles), 2-deep stage ring (fp8 / fp32 operands); Cfg8P: the same tile on the 8-phase
//     schedule with 16x16x32 MFMAs (bf16 operands)                                      -- >= 140 such tiles (T ~ 1952, batched-env
//     prefill, 9-frame ViT); with two K slices for long-K products of 96-128 tiles (window-restart down_proj)
//   * CfgSkinny  32x128, 2 waves                                                        -- M <= 32 (envs decoded in lockstep)
//   * Cfg64  64x64 (force_cfg only)
// LDS rows are 128 B: the 16-byte chunk index is XOR-swizzled with the row bits above the 256-byte bank row so the 16 lanes
// of every ds_read_b128 group hit 16 distinct 16-byte slots (the swizzle is applied on the DMA's source address).
// Workgroup ids are remapped so that each of the 8 XCDs owns a contiguous band of tiles (operand panels shared through the
// XCD's private L2).
//
// Roofline: MFMA for M >= ~512 (measured 1.0-1.1 PF/s on the 8-phase kernel, MFMA pipe busy 55-60 %); at M ~ 212 the per-CU stream rate
// (a CU keeps ~64 lines of 128 B in flight: stage ~ W_bytes / 17 GB/s + A_bytes / 69 GB/s per CU).
// Algorithmic flops = 2*M*N*K, algorithmic bytes = (M*K + N*K + M*N) * sizeof(T).
#include <cstdlib>
#include <type_traits>

#include "common.h"
#include "kernels.h"

namespace svln {

namespace {

template <int BM_, int BN_, int WM_, int WN_, int ROWB_, bool DEEP_, int NBUF_, bool ILV_ = false, int KG_ = 1, bool P8_ = false> struct TileCfg {
    // P8: the 8-phase schedule of p8_mainloop (256x256 tile, bf16 operands) instead of the stage ring
    static constexpr bool P8 = P8_;
    static constexpr int BM = BM_, BN = BN_, WM = WM_, WN = WN_, ROWB = ROWB_;
    // KG > 1: K is split INSIDE the workgroup: KG groups of WM x WN waves each run the stage pipeline (own LDS ring) over 1/KG of the K
    // stages of the same output tile and exchange their accumulators through LDS at the end -- the shorter K chain of a split-K launch
    // without fp32 slabs in HBM and without a second (reduce) launch
    static constexpr int KG = KG_;
    static constexpr bool DEEP = DEEP_;
    static constexpr int NBUF = NBUF_;                        // LDS ring depth of the direct-to-LDS (glds) pipeline
    // ILV: the fragment reads of macro step s+1 are issued one per gap between the MFMAs of step s instead of as one burst before
    // them (measured: 4-6 % faster when a CU runs many tiles back to back, slower for single-round launches with short K)
    static constexpr bool ILV = ILV_;
    static constexpr int THREADS = 64 * WM * WN * KG;
    static constexpr int CH = ROWB / 16;                      // 16-byte chunks per row per stage
    static constexpr int ROWS_PER_BANKROW = 256 / ROWB;       // 2 or 4
    static constexpr int SH = ROWS_PER_BANKROW == 2 ? 1 : 2;
    static constexpr int STAGE_BYTES = (BM + BN) * ROWB;
    static constexpr int A_LOADS = BM * CH / (THREADS / KG), W_LOADS = BN * CH / (THREADS / KG);
    static constexpr int LDS_BYTES = KG * NBUF * STAGE_BYTES;
    static constexpr int MI = BM / WM / 32, NJ = BN / WN / 32;   // 32x32 accumulator tiles per wave
    static_assert(BM / WM % 32 == 0 && BN / WN % 64 == 0, "wave tile: rows a multiple of 32, columns a multiple of 64 ([gate 32 | up 32] blocks for SwiGLU)");
    static_assert(BM * CH % (THREADS / KG) == 0 && BN * CH % (THREADS / KG) == 0, "staging must divide evenly");
};
using Cfg128 = TileCfg<128, 128, 2, 2, 128, false, 2>;
using Cfg256 = TileCfg<256, 128, 4, 2, 128, true, 3>;
using CfgBig = TileCfg<256, 256, 2, 4, 128, false, 2, true>;
using Cfg8P = TileCfg<256, 256, 2, 4, 128, false, 2, false, 1, true>;   // same tile, p8_mainloop schedule (bf16 operands, M > 512), 16x16x32 MFMAs
using Cfg8P32 = TileCfg<256, 256, 2, 4, 128, false, 2, true, 1, true>;  // ... with 32x32x16 MFMAs (bit-identical to the stage ring; A/B and tests)
using Cfg128L = TileCfg<128, 128, 2, 2, 128, false, 2, true>;  // Cfg128 for more than one round of tiles
using Cfg64 = TileCfg<64, 64, 2, 1, 128, false, 6>;          // 64x64, 2 waves (32x64 wave tiles), five 16 KB stages in flight: >= 200 workgroups WITHOUT a K split for
                                                             // products with few 128-wide column tiles and a short K (force_cfg 64 only, see launch_epi)
using Cfg128K2 = TileCfg<128, 128, 2, 2, 128, false, 2, false, 2>;   // 128x128 tile, 2 K-groups of 4 waves (128 KB of LDS: one workgroup per CU): one-round launches with a
                                                             // short K and an epilogue that wants the finished value (one-frame ViT qkv / fc1, projector)
using Cfg256N64 = TileCfg<256, 64, 8, 1, 128, true, 3>;       // 256x64, 8 waves stacked along M (32 x 64 wave tiles): the CDNA guide's tile for M = 256 projections
                                                             // (force_cfg 264 only: measured inside the turn against 256x128 x split-K)
using CfgSkinny = TileCfg<32, 128, 1, 2, 128, false, 3>;     // M <= 32 (lockstep decode of several envs): 2 waves, 20 KB stages, glds kernel only

template <typename C> SVLN_DEV int swz(int row, int c) { return (c ^ ((row >> C::SH) & (C::CH - 1))) << 4; }

constexpr int EPI_ARGMAX_ = 4;
static_assert(EPI_ARGMAX_ == EPI_ARGMAX, "enum");
template <typename T, int EPI> SVLN_DEV float epi_act(float v) {
    if (EPI == EPI_GELU_TANH) return gelu_tanh_f(v);
    if (EPI == EPI_GELU_ERF) return gelu_erf_f(v);
    return v;
}

typedef __attribute__((address_space(3))) void* lds_ptr_t;
typedef const __attribute__((address_space(1))) void* gbl_ptr_t;

// ---- 8-phase schedule for the 256x256 tile (bf16 operands; the window-restart products, M = 1952 and 6561) ----------------------------
// Wave (wr, wc) = (wave >> 2, wave & 3) owns 128 x 64 outputs: 8 x 4 accumulators of 16x16 (M16, the shipping form: v_mfma_f32_16x16x32_bf16
// holds a higher clock under the sustained load of these products) or 4 x 2 of 32x32 (the form that is bit-identical to the stage ring,
// kept for the tests).  A K tile (64 of K) is consumed in four phases, one 64 x 32 output quadrant each (16 MFMAs of 16x16x32 or 8 of
// 32x32x16 = 256 MFMA cycles; the fragment-read counts are the same for both forms):
//   phase 1: read B sub0 (4 ds_read_b128) + A sub0 (8)   -> quadrant (A0, B0)
//   phase 2: read B sub1 (4)                              -> (A0, B1)
//   phase 3: read A sub1 (8, over the A0 registers)       -> (A1, B1)
//   phase 4: no reads                                     -> (A1, B0)
// and is staged as four HALF-TILES of 16 KiB: A-h{a} = the sub-a rows of both wave rows, B-h{b} = the sub-b columns of the four
// wave columns -- so a half-tile is dead as soon as its phase has read it.  Two K tiles per iteration (even / odd buffer), every phase:
//   { fragment reads ; ONE half-tile of LDS-DMA (2 x 1 KiB per wave) ; [counted waits] ; barrier ; lgkmcnt(0) ; MFMAs ; barrier }
// DMA order (phase: half-tile):  1: odd.A-h1 (tile T+1) | 2: even.B-h0 (T+2) | 3: even.A-h0 | 4: even.B-h1 | 5: even.A-h1 |
//                                6: odd.B-h0 (T+3) | 7: odd.A-h0 | 8: odd.B-h1
//   WAR: a half-tile is re-staged two phases after the phase that read it, or one phase after when that phase retired the reads before
//        its first barrier (the B sub0 reads of phases 1 / 5: issued first, s_waitcnt lgkmcnt(8) before the barrier).
//   RAW: s_waitcnt vmcnt(6) at phases 4 and 8 only (three half-tiles stay in flight): phase 4's retires the four odd half-tiles (issued in
//        phases 6, 7, 8, 1), read from phase 5 on; phase 8's the even ones (phases 2-5), read from phase 1 on -- always behind both barriers
//        of the waiting phase, which also covers the other wave group.
// The wave rows run the same program one barrier apart (waves 4-7 take one extra barrier first, waves 0-3 one last): the two waves of
// a SIMD (w and w + 4) alternate between the read / DMA part and the MFMA part of a phase.
// Tiles past the K range of the launch are staged from the last real tile (never read); the last iteration stages nothing and drains.
template <typename C, bool M16, typename ACC>
SVLN_DEV void p8_mainloop(const GemmArgs& p, char* smem, int row0, int col0, int st_begin, int n, int kchunks, int wave, int lane, ACC& acc) {
    static_assert(C::BM == 256 && C::BN == 256 && C::WM == 2 && C::WN == 4 && C::KG == 1 && C::ROWB == 128 && C::LDS_BYTES == 131072, "8-phase geometry");
    constexpr int A_REG = 0, B_REG = 65536, BUFB = 32768;
    const int wr = wave >> 2, wc = wave & 3, l8 = lane >> 3;
    const int cj = (lane & 7) ^ (((wave * 8 + l8) >> 1) & 7);            // source chunk of this lane: the same for all eight pointers
    const char* sA[2][2];
    const char* sB[2][2];
    int dA[2][2], dB[2][2];
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int rowA = j * 128 + a * 64 + wave * 8;
            const int colB = ((wave >> 2) + 2 * j) * 64 + a * 32 + (wave & 3) * 8;
            dA[a][j] = A_REG + rowA * 128;
            dB[a][j] = B_REG + colB * 128;
            sA[a][j] = (const char*)((const bf16*)p.A + (size_t)min(row0 + rowA + l8, p.M - 1) * p.lda) + cj * 16;
            sB[a][j] = (const char*)((const bf16*)p.W + (size_t)min(col0 + colB + l8, p.N - 1) * p.ldw) + cj * 16;
        }
    auto issue = [&](auto KINDc, auto BUFc, int t) {                     // kind 0: A-h0, 1: A-h1, 2: B-h0, 3: B-h1
        constexpr int kind = decltype(KINDc)::value, bufi = decltype(BUFc)::value;
        const int st = st_begin + min(t, n - 1);
        const bool full = (st + 1) * 8 <= kchunks;
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const char* g = (kind < 2 ? sA[kind & 1][j] : sB[kind & 1][j]) + (size_t)st * 128;
            if (!full && st * 8 + cj >= kchunks) g = (const char*)p.zeros;
            char* dst = smem + bufi * BUFB + (kind < 2 ? dA[kind & 1][j] : dB[kind & 1][j]);
            __builtin_amdgcn_global_load_lds((gbl_ptr_t)g, (lds_ptr_t)dst, 16, 0, 0);
        }
    };
    // fragment read addresses.  32x32x16: lane (r32, h) reads row r32, chunk 2s + h of K step s (4 steps of 16); 16x16x32: lane (r16, g)
    // reads row r16, chunk 4s + g of K step s (2 steps of 32).  Row blocks and sub-tiles are immediate offsets of the read.
    constexpr int NS = M16 ? 2 : 4;
    const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) char*)smem;
    unsigned aA[NS], aB[NS];
#pragma unroll
    for (int s = 0; s < NS; ++s) {
        const int rr = M16 ? (lane & 15) : (lane & 31);
        const int ch = M16 ? 4 * s + (lane >> 4) : 2 * s + (lane >> 5);
        const unsigned sw = (unsigned)((ch ^ ((rr >> 1) & 7)) << 4);
        aA[s] = lds0 + A_REG + (wr * 128 + rr) * 128 + sw;
        aB[s] = lds0 + B_REG + (wc * 64 + rr) * 128 + sw;
    }
    u32x4 fa[8], fb[2][4];           // A sub: [row block][K step] (2 x 4 or 4 x 2); B sub0 / sub1: [column block][K step] (1 x 4 or 2 x 2)
#define SVLN_P8_READ(dst, addr, off) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(dst) : "v"(addr), "n"(off))
    using I0 = std::integral_constant<int, 0>; using I1 = std::integral_constant<int, 1>; using I2 = std::integral_constant<int, 2>; using I3 = std::integral_constant<int, 3>;
    auto phase = [&](auto PHc, int T, bool last) {
        constexpr int PH = decltype(PHc)::value, cur = PH >> 2, q = PH & 3, bo = cur * BUFB;
        constexpr int ao = bo + (q == 2 ? 8192 : 0), bbo = bo + (q == 1 ? 4096 : 0), bs = q == 1 ? 1 : 0;
        if constexpr (q == 0 || q == 1) {                                 // B sub0 / sub1
            if constexpr (M16) {
#pragma unroll
                for (int s = 0; s < 2; ++s) { SVLN_P8_READ(fb[bs][s], aB[s], bbo); SVLN_P8_READ(fb[bs][2 + s], aB[s], bbo + 2048); }
            } else {
#pragma unroll
                for (int s = 0; s < 4; ++s) SVLN_P8_READ(fb[bs][s], aB[s], bbo);
            }
            __builtin_amdgcn_sched_barrier(0);
        }
        if constexpr (q == 0 || q == 2) {                                 // A sub0 / sub1
            if constexpr (M16) {
#pragma unroll
                for (int s = 0; s < 2; ++s) {
                    SVLN_P8_READ(fa[s], aA[s], ao); SVLN_P8_READ(fa[2 + s], aA[s], ao + 2048);
                    SVLN_P8_READ(fa[4 + s], aA[s], ao + 4096); SVLN_P8_READ(fa[6 + s], aA[s], ao + 6144);
                }
            } else {
#pragma unroll
                for (int s = 0; s < 4; ++s) { SVLN_P8_READ(fa[s], aA[s], ao); SVLN_P8_READ(fa[4 + s], aA[s], ao + 4096); }
            }
        }
        if (!last) {
            if constexpr (PH == 0) issue(I1{}, I1{}, T + 1);
            if constexpr (PH == 1) issue(I2{}, I0{}, T + 2);
            if constexpr (PH == 2) issue(I0{}, I0{}, T + 2);
            if constexpr (PH == 3) issue(I3{}, I0{}, T + 2);
            if constexpr (PH == 4) issue(I1{}, I0{}, T + 2);
            if constexpr (PH == 5) issue(I2{}, I1{}, T + 3);
            if constexpr (PH == 6) issue(I0{}, I1{}, T + 3);
            if constexpr (PH == 7) issue(I3{}, I1{}, T + 3);
        } else if constexpr (PH == 0) {
            if (T + 1 < n) issue(I1{}, I1{}, T + 1);
        }
        if constexpr (q == 0) asm volatile("s_waitcnt lgkmcnt(8)" ::: "memory");
        if constexpr (q == 3) {
            if (last) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            else asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
        }
        __builtin_amdgcn_s_barrier();
        if constexpr (q == 0)
            asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(fa[0]), "+v"(fa[1]), "+v"(fa[2]), "+v"(fa[3]), "+v"(fa[4]), "+v"(fa[5]), "+v"(fa[6]), "+v"(fa[7]),
                         "+v"(fb[0][0]), "+v"(fb[0][1]), "+v"(fb[0][2]), "+v"(fb[0][3]));
        else if constexpr (q == 1) asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(fb[1][0]), "+v"(fb[1][1]), "+v"(fb[1][2]), "+v"(fb[1][3]));
        else if constexpr (q == 2)
            asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(fa[0]), "+v"(fa[1]), "+v"(fa[2]), "+v"(fa[3]), "+v"(fa[4]), "+v"(fa[5]), "+v"(fa[6]), "+v"(fa[7]));
        constexpr int qa = q >> 1, qb = (q == 1 || q == 2) ? 1 : 0;
        __builtin_amdgcn_s_setprio(1);
        if constexpr (M16) {
#pragma unroll
            for (int s = 0; s < 2; ++s)
#pragma unroll
                for (int ib = 0; ib < 4; ++ib)
#pragma unroll
                    for (int jb = 0; jb < 2; ++jb)
                        acc[4 * qa + ib][2 * qb + jb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, fa[2 * ib + s]),
                                                                                              __builtin_bit_cast(bf16x8, fb[qb][2 * jb + s]), acc[4 * qa + ib][2 * qb + jb], 0, 0, 0);
        } else {
#pragma unroll
            for (int s = 0; s < 4; ++s)
#pragma unroll
                for (int ib = 0; ib < 2; ++ib)
                    acc[2 * qa + ib][qb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, fa[4 * ib + s]), __builtin_bit_cast(bf16x8, fb[qb][s]),
                                                                                    acc[2 * qa + ib][qb], 0, 0, 0);
        }
        __builtin_amdgcn_s_setprio(0);
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_barrier();
    };
    if (n <= 0) return;
    // prologue: the even buffer (tile 0) and three half-tiles of the odd one, in the order the loop continues
    issue(I2{}, I0{}, 0); issue(I0{}, I0{}, 0); issue(I3{}, I0{}, 0); issue(I1{}, I0{}, 0);
    issue(I2{}, I1{}, 1); issue(I0{}, I1{}, 1); issue(I3{}, I1{}, 1);
    asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    if (wr == 1) __builtin_amdgcn_s_barrier();
    for (int T = 0; T < n; T += 2) {
        const bool last = T + 2 >= n;
        phase(std::integral_constant<int, 0>{}, T, last);
        phase(std::integral_constant<int, 1>{}, T, last);
        phase(std::integral_constant<int, 2>{}, T, last);
        phase(std::integral_constant<int, 3>{}, T, last);
        if (T + 1 >= n) break;
        phase(std::integral_constant<int, 4>{}, T, last);
        phase(std::integral_constant<int, 5>{}, T, last);
        phase(std::integral_constant<int, 6>{}, T, last);
        phase(std::integral_constant<int, 7>{}, T, last);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (wr == 0) __builtin_amdgcn_s_barrier();
#undef SVLN_P8_READ
}

// epilogue of the 16x16x32 form (accumulator (mi, nj): rows mi * 16 + 4 * (lane >> 4) + e, column nj * 16 + (lane & 15) of the wave's
// 128 x 64 tile): bias / activation / residual, or SwiGLU over the [gate 32 | up 32] column blocks (gate nj pairs with up nj + 2)
template <typename T, int EPI>
SVLN_DEV void p8_epilogue16(const GemmArgs& p, int row0, int col0, int wave, int lane, const f32x4 (&acc)[8][4]) {
    const int wr = wave >> 2, wc = wave & 3, r16 = lane & 15, g = lane >> 4;
    T* Cc = (T*)p.C;
    const T* bias = (const T*)p.bias;
    const T* res = (const T*)p.res;
    if constexpr (EPI == EPI_SWIGLU) {
#pragma unroll
        for (int nj = 0; nj < 2; ++nj) {
            const int n_out = ((col0 + wc * 64) >> 1) + nj * 16 + r16;
            const bool ok_n = (col0 + wc * 64 + 32 + nj * 16 + r16) < p.N;
#pragma unroll
            for (int mi = 0; mi < 8; ++mi)
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const int m = row0 + wr * 128 + mi * 16 + 4 * g + e;
                    if (m < p.M && ok_n) Cc[(size_t)m * p.ldc + n_out] = from_f32<T>(silu_f(acc[mi][nj][e]) * acc[mi][nj + 2][e]);
                }
        }
        return;
    }
#pragma unroll
    for (int nj = 0; nj < 4; ++nj) {
        const int nn = col0 + wc * 64 + nj * 16 + r16;
        if (nn >= p.N) continue;
        const float bv = bias ? to_f32(bias[nn]) : 0.0f;
#pragma unroll
        for (int mi = 0; mi < 8; ++mi)
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int m = row0 + wr * 128 + mi * 16 + 4 * g + e;
                if (m >= p.M) continue;
                float v = epi_act<T, EPI>(acc[mi][nj][e] + bv);
                if (res) {
                    const int rr = p.res_mod > 0 ? m % p.res_mod : m;
                    v += to_f32(res[(size_t)rr * p.ldr + nn]);
                }
                Cc[(size_t)m * p.ldc + nn] = from_f32<T>(v);
            }
    }
}


// The same epilogue with the stores staged through LDS (bf16 outputs, 16-byte aligned rows).  In the 16x16 accumulator layout a lane holds ONE
// column of 4 rows, so the direct form above stores 2 bytes per lane in 32-byte pieces: measured on isolated launches, 115 of the 542 us of the
// T = 1952 gate/up product and 35 of the 90 us of the nine-frame fc1 were those stores.  Here each wave rounds its values into a private LDS
// tile (64 rows at a time, rows padded by 16 B: the four row groups of a store instruction fall into different banks), reads it back as
// 16-byte row chunks and stores whole 64- / 128-byte row segments.  Same arithmetic and rounding as p8_epilogue16: bit-identical outputs.
// The caller has put a workgroup barrier between the last stage reads and this call (the tiles overlay the stage ring).
template <typename T, int EPI>
SVLN_DEV void p8_epilogue16_staged(const GemmArgs& p, char* smem, int row0, int col0, int wave, int lane, const f32x4 (&acc)[8][4]) {
    static_assert(sizeof(T) == 2, "staged epilogue: 2-byte outputs");
    constexpr bool GLU = EPI == EPI_SWIGLU;
    constexpr int COLS = GLU ? 32 : 64, STRIDE = COLS * 2 + 16, CPR = COLS / 8;       // chunks of 8 outputs per row
    const int wr = wave >> 2, wc = wave & 3, r16 = lane & 15, g = lane >> 4;
    char* sE = smem + wave * (64 * STRIDE);
    T* Cc = (T*)p.C;
    const T* bias = (const T*)p.bias;
    const T* res = (const T*)p.res;
    const int ncols_out = GLU ? p.N / 2 : p.N;                                          // columns of C
    const int cbase = GLU ? ((col0 + wc * 64) >> 1) : col0 + wc * 64;                   // first C column of this wave
#pragma unroll
    for (int half = 0; half < 2; ++half) {
#pragma unroll
        for (int mq = 0; mq < 4; ++mq) {
            const int mi = half * 4 + mq;
            if constexpr (GLU) {
#pragma unroll
                for (int nj = 0; nj < 2; ++nj)
#pragma unroll
                    for (int e = 0; e < 4; ++e)
                        *(T*)(sE + (mq * 16 + 4 * g + e) * STRIDE + (nj * 16 + r16) * 2) = from_f32<T>(silu_f(acc[mi][nj][e]) * acc[mi][nj + 2][e]);
            } else {
#pragma unroll
                for (int nj = 0; nj < 4; ++nj) {
                    const int nn = col0 + wc * 64 + nj * 16 + r16;
                    const float bv = (bias && nn < p.N) ? to_f32(bias[nn]) : 0.0f;
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        float v = epi_act<T, EPI>(acc[mi][nj][e] + bv);
                        if (res) {
                            const int m = row0 + wr * 128 + mi * 16 + 4 * g + e;
                            if (m < p.M && nn < p.N) {
                                const int rr = p.res_mod > 0 ? m % p.res_mod : m;
                                v += to_f32(res[(size_t)rr * p.ldr + nn]);
                            }
                        }
                        *(T*)(sE + (mq * 16 + 4 * g + e) * STRIDE + (nj * 16 + r16) * 2) = from_f32<T>(v);
                    }
                }
            }
        }
        // (the wave's own LDS writes and reads are ordered; no other wave touches this tile)
#pragma unroll
        for (int k = 0; k < 64 * CPR / 64; ++k) {
            const int q = k * 64 + lane, row = q / CPR, c = q - row * CPR;
            const uint4 v = *(const uint4*)(sE + row * STRIDE + c * 16);
            const int m = row0 + wr * 128 + half * 64 + row, cc = cbase + c * 8;
            if (m >= p.M || cc >= ncols_out) continue;
            T* dst = Cc + (size_t)m * p.ldc + cc;
            if (cc + 8 <= ncols_out) {
                *(uint4*)dst = v;
            } else {            // ragged last chunk of a row: element by element (unrolled: no address of v is taken)
                const unsigned w4[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
                for (int t = 0; t < 8; ++t)
                    if (cc + t < ncols_out) ((uint16_t*)dst)[t] = (uint16_t)((t & 1) ? w4[t >> 1] >> 16 : w4[t >> 1] & 0xFFFFu);
            }
        }
    }
}


// Same tiles and epilogues, operands staged with LDS-DMA (global_load_lds_dwordx4: HBM/L2 -> LDS with no VGPR or
// ds_write hop).  One wave instruction fills 1 KiB of LDS = 8 consecutive 128-byte tile rows, lane l -> row l/8,
// physical chunk l%8; the XOR swizzle is applied on the per-lane SOURCE address (the LDS image must stay lane-linear).
// Ring of NBUF stage buffers, NBUF-1 stages in flight behind a counted s_waitcnt vmcnt + one raw s_barrier per stage:
//   wait(stage i landed) ; barrier ; issue stage i+NBUF-1 into the buffer read in iteration i-1 ; MFMAs on stage i.
// Rows beyond M / N are clamped to the last valid row (their outputs are never stored); K-tail chunks read a zero line.
// NTW: the weight tile is staged with non-temporal LDS-DMA (products with ONE row tile: every weight byte is read once, by one workgroup).
// A template parameter, not a run-time flag: a branch around the DMA issue of the large-tile kernels cost the window-restart prefill 2.5 ms.
// VP: the fused K / V^T packing tail of the SigLIP QKV product (a template parameter: as a run-time flag its tests sat in the store loop of
// every plain epilogue and cost the window-restart turn 2.8 ms).
template <typename T, int EPI, typename C, bool SPLITK, typename TA = T, bool NTW = false, bool VP = false>
__global__ __launch_bounds__(C::THREADS) void gemm_glds_kernel(GemmArgs p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int EPC = Elt<TA>::PER_CHUNK;                   // TA = operand storage (T, or fp8_t with per-row scales applied in the epilogue)
    constexpr bool FP8 = sizeof(TA) == 1;
    static_assert(C::ROWB == 128 && C::CH == 8, "glds path: 128-byte tile rows (4 macro steps per stage)");
    constexpr int KG = C::KG;
    constexpr int WAVES = C::THREADS / 64 / KG;               // waves of one K-group
    constexpr int BLK_A = C::BM * C::ROWB / 1024, BLK_W = C::BN * C::ROWB / 1024;
    constexpr int PER_WAVE = (BLK_A + BLK_W) / WAVES;
    constexpr int D = C::NBUF - 1;
    static_assert((BLK_A + BLK_W) % WAVES == 0, "blocks must divide over waves");
    static_assert(!NTW || BLK_A % WAVES == 0, "NTW: a wave's blocks must be all-activation or all-weight per index");
    static_assert(KG == 1 || (!SPLITK && C::MI % KG == 0), "in-workgroup K groups: unsplit launches, each group finishes MI / KG accumulator rows");
    const int tid = threadIdx.x, lane = tid & 63, wave_all = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int kg = KG > 1 ? wave_all / WAVES : 0, wave = KG > 1 ? wave_all % WAVES : wave_all;
    const int wr = wave / C::WN, wc = wave % C::WN;
    const int r32 = lane & 31, h = lane >> 5;
    char* const ring = smem + (KG > 1 ? kg * C::NBUF * C::STAGE_BYTES : 0);      // this K-group's stage ring

    const int tiles_m = (p.M + C::BM - 1) / C::BM;
    const int nsplit = SPLITK ? p.nsplit : 1;
    const int nwg = p.launch_tiles * nsplit;            // p.launch_tiles = (row tiles) x (column tiles of THIS launch)
    int bid = blockIdx.x;
    {
        const int xcd = bid & 7, q = nwg >> 3, r = nwg & 7;
        bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (bid >> 3);
    }
    // Workgroup order (after the XCD remap consecutive ids share an XCD and its L2): row tiles fastest -- neighbours share the WEIGHT tile,
    // the activations stream through (the LLM products: weights >> activations); bn_fast: column tiles fastest -- neighbours share the
    // ACTIVATION rows (the nine-frame ViT products: 6561 rows against 1152-4304 columns; fc2 reads 56 MB of activations, 10 MB of weights)
    const int tiles_n_l = p.launch_tiles / tiles_m;
    const bool bnf = !SPLITK && p.bn_fast;
    const int bm = bnf ? bid / tiles_n_l : bid % tiles_m;
    const int ks = (bid / tiles_m) % nsplit;
    const int bn = p.tile_base + (bnf ? bid % tiles_n_l : bid / (tiles_m * nsplit));
    const int row0 = bm * C::BM, col0 = bn * C::BN;
    const int kchunks = p.K / EPC;
    const int stages_total = (kchunks + C::CH - 1) / C::CH;
    const int stages_per = (stages_total + nsplit - 1) / nsplit;
    int st_begin = ks * stages_per;
    int st_end = min(stages_total, st_begin + stages_per);
    int n_it = st_end - st_begin;                                                // barrier-uniform iteration count of the stage loop
    if (KG > 1) {                                                                // this K-group's share of the workgroup's stages
        const int per = (n_it + KG - 1) / KG;
        st_begin += kg * per;
        st_end = min(st_end, st_begin + per);
        n_it = per;
    }
    const int n = max(st_end - st_begin, 0);

    // per-lane source pointers (at K stage 0) and wave-uniform LDS offsets of this wave's 1-KiB blocks
    // (blocks wholly past M or N -- 5 of the 32 activation blocks of a 212-row steady prefill -- still staged: they re-read row M - 1, i.e. L1 hits;
    //  skipping them with per-wave wait counts measured 8.41 / 8.43 / 8.42 ms of prefill per turn against 8.45 / 8.43 / 8.42: nothing)
    const char* src[PER_WAVE];
    int cj[PER_WAVE], loff[PER_WAVE];
#pragma unroll
    for (int j = 0; j < PER_WAVE; ++j) {
        const int b = wave + WAVES * j;
        const bool isA = b < BLK_A;
        const int blk = isA ? b : b - BLK_A;
        const int row = blk * 8 + (lane >> 3);
        const int c = (lane & 7) ^ ((row >> C::SH) & (C::CH - 1));
        cj[j] = c;
        loff[j] = (isA ? 0 : C::BM * C::ROWB) + blk * 1024;
        if (isA) src[j] = (const char*)((const TA*)p.A + (size_t)min(row0 + row, p.M - 1) * p.lda) + c * 16;
        else src[j] = (const char*)((const TA*)p.W + (size_t)min(col0 + row, p.N - 1) * p.ldw) + c * 16;
    }
    auto issue = [&](int st, int buf) {
        char* base = ring + buf * C::STAGE_BYTES;
        const bool full = (st + 1) * C::CH <= kchunks;
        if (full) {                       // every stage but a ragged last one: no per-lane source select in front of the DMA instructions
#pragma unroll
            for (int j = 0; j < PER_WAVE; ++j) {
                const char* g = src[j] + (size_t)st * C::ROWB;
                if (NTW && j >= BLK_A / WAVES) __builtin_amdgcn_global_load_lds((gbl_ptr_t)g, (lds_ptr_t)(base + loff[j]), 16, 0, 2);
                else __builtin_amdgcn_global_load_lds((gbl_ptr_t)g, (lds_ptr_t)(base + loff[j]), 16, 0, 0);
            }
            return;
        }
#pragma unroll
        for (int j = 0; j < PER_WAVE; ++j) {
            const char* g = src[j] + (size_t)st * C::ROWB;
            if (st * C::CH + cj[j] >= kchunks) g = (const char*)p.zeros;
            // NTW: the weight blocks (j >= BLK_A / WAVES for every wave) non-temporal, aux = 2 (measured inside the turn: steady prefill
            // 6.80 -> 6.63 ms); the activation panel, re-read by every workgroup, stays cached
            if (NTW && j >= BLK_A / WAVES) __builtin_amdgcn_global_load_lds((gbl_ptr_t)g, (lds_ptr_t)(base + loff[j]), 16, 0, 2);
            else __builtin_amdgcn_global_load_lds((gbl_ptr_t)g, (lds_ptr_t)(base + loff[j]), 16, 0, 0);
        }
    };

    constexpr int MI = C::MI, NJ = C::NJ, STEPS = C::CH / 2, RD = MI + NJ;     // fragment reads per macro step
    constexpr bool INTERLEAVE = C::ILV;
    constexpr int WROWS = C::BM / C::WM, WCOLS = C::BN / C::WN;
    f32x16 acc[MI][NJ];
#pragma unroll
    for (int i = 0; i < MI; ++i)
#pragma unroll
        for (int j = 0; j < NJ; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.0f;

    // LDS fragment reads are issued as inline asm: hipcc orders every ds_read it can see behind ALL outstanding
    // LDS-DMA (it emits s_waitcnt vmcnt(0) before the first ds_read of the stage, draining the ring); the asm reads are
    // ordered by our own protocol instead (counted vmcnt + barrier above, lgkmcnt waits tied to the destinations below).
    const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) char*)ring;
    unsigned offA[STEPS][MI], offW[STEPS][NJ];
#pragma unroll
    for (int s = 0; s < STEPS; ++s) {
#pragma unroll
        for (int i = 0; i < MI; ++i) {
            const int ra_ = wr * WROWS + i * 32 + r32;
            offA[s][i] = lds0 + ra_ * C::ROWB + swz<C>(ra_, 2 * s + h);
        }
#pragma unroll
        for (int j = 0; j < NJ; ++j) {
            const int rw_ = wc * WCOLS + j * 32 + r32;
            offW[s][j] = lds0 + C::BM * C::ROWB + rw_ * C::ROWB + swz<C>(rw_, 2 * s + h);
        }
    }
    // fragments double-buffered by macro step: the reads of step s+1 are in flight under the MFMAs of step s
    u32x4 fa[2][MI], fb[2][NJ];
    auto read_step = [&](int s, unsigned bo) {
#pragma unroll
        for (int i = 0; i < MI; ++i) asm volatile("ds_read_b128 %0, %1" : "=v"(fa[s & 1][i]) : "v"(offA[s][i] + bo));
#pragma unroll
        for (int j = 0; j < NJ; ++j) asm volatile("ds_read_b128 %0, %1" : "=v"(fb[s & 1][j]) : "v"(offW[s][j] + bo));
    };
    auto wait_step = [&](int k, bool more) {          // step in buffer k landed (`more`: the next step's RD reads stay in flight)
        if constexpr (MI == 4 && NJ == 4) {
            if (more) asm volatile("s_waitcnt lgkmcnt(%8)" : "+v"(fa[k][0]), "+v"(fa[k][1]), "+v"(fa[k][2]), "+v"(fa[k][3]), "+v"(fb[k][0]), "+v"(fb[k][1]), "+v"(fb[k][2]), "+v"(fb[k][3]) : "n"(RD));
            else asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(fa[k][0]), "+v"(fa[k][1]), "+v"(fa[k][2]), "+v"(fa[k][3]), "+v"(fb[k][0]), "+v"(fb[k][1]), "+v"(fb[k][2]), "+v"(fb[k][3]));
        } else if constexpr (MI == 4) {
            if (more) asm volatile("s_waitcnt lgkmcnt(%6)" : "+v"(fa[k][0]), "+v"(fa[k][1]), "+v"(fa[k][2]), "+v"(fa[k][3]), "+v"(fb[k][0]), "+v"(fb[k][1]) : "n"(RD));
            else asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(fa[k][0]), "+v"(fa[k][1]), "+v"(fa[k][2]), "+v"(fa[k][3]), "+v"(fb[k][0]), "+v"(fb[k][1]));
        } else if constexpr (MI == 1) {
            if (more) asm volatile("s_waitcnt lgkmcnt(%3)" : "+v"(fa[k][0]), "+v"(fb[k][0]), "+v"(fb[k][1]) : "n"(RD));
            else asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(fa[k][0]), "+v"(fb[k][0]), "+v"(fb[k][1]));
        } else {
            if (more) asm volatile("s_waitcnt lgkmcnt(%4)" : "+v"(fa[k][0]), "+v"(fa[k][1]), "+v"(fb[k][0]), "+v"(fb[k][1]) : "n"(RD));
            else asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(fa[k][0]), "+v"(fa[k][1]), "+v"(fb[k][0]), "+v"(fb[k][1]));
        }
    };
    // one fragment read of step s (A fragments first, then W)
    auto read_one = [&](int s, int q, unsigned bo) {
#pragma unroll
        for (int i = 0; i < MI; ++i)
            if (q == i) asm volatile("ds_read_b128 %0, %1" : "=v"(fa[s & 1][i]) : "v"(offA[s][i] + bo));
#pragma unroll
        for (int j = 0; j < NJ; ++j)
            if (q == MI + j) asm volatile("ds_read_b128 %0, %1" : "=v"(fb[s & 1][j]) : "v"(offW[s][j] + bo));
    };
    // after_first_reads: the DMA issue of a later stage, placed behind the first fragment reads of this one so that its address arithmetic
    // (8 loads x ~8 VALU + the m0 set-up per wave, all waves at once right after the barrier) runs under their LDS latency
    auto compute = [&](int buf, auto&& after_first_reads) {
        const unsigned bo = buf * C::STAGE_BYTES;
        read_step(0, bo);
        after_first_reads();
#pragma unroll
        for (int s = 0; s < STEPS; ++s) {
            if (INTERLEAVE) {
                // the reads of step s+1 go into the gaps between the MFMAs of step s (one per gap) instead of one burst before them
                wait_step(s & 1, false);
                int q = 0;
#pragma unroll
                for (int i = 0; i < MI; ++i)
#pragma unroll
                    for (int j = 0; j < NJ; ++j) {
                        const u32x4 a4 = fa[s & 1][i], b4 = fb[s & 1][j];
                        mma_chunk<TA>(make_uint4(a4.x, a4.y, a4.z, a4.w), make_uint4(b4.x, b4.y, b4.z, b4.w), acc[i][j]);
                        if (s + 1 < STEPS && q < RD) read_one(s + 1, q, bo);
                        ++q;
                        __builtin_amdgcn_sched_barrier(0);
                    }
                if (s + 1 < STEPS)
                    for (; q < RD; ++q) read_one(s + 1, q, bo);
                continue;
            }
            if (s + 1 < STEPS) read_step(s + 1, bo);
            wait_step(s & 1, s + 1 < STEPS);
#pragma unroll
            for (int i = 0; i < MI; ++i)
#pragma unroll
                for (int j = 0; j < NJ; ++j) {
                    const u32x4 a4 = fa[s & 1][i], b4 = fb[s & 1][j];
                    mma_chunk<TA>(make_uint4(a4.x, a4.y, a4.z, a4.w), make_uint4(b4.x, b4.y, b4.z, b4.w), acc[i][j]);
                }
            __builtin_amdgcn_sched_barrier(0);      // keep step s's MFMAs ahead of step s+1's wait
        }
    };

    if constexpr (C::P8) {
        static_assert(std::is_same<TA, bf16>::value && MI == 4 && NJ == 2 && !NTW && !VP && (!SPLITK || !C::ILV), "8-phase schedule: bf16 operands, 128 x 64 wave tiles");
        if constexpr (C::ILV) {                        // (the ILV flag of the 8-phase configurations selects the 32x32x16 form)
            p8_mainloop<C, false>(p, smem, row0, col0, st_begin, n, kchunks, wave_all, lane, acc);
        } else {
            f32x4 acc16[8][4];
#pragma unroll
            for (int i = 0; i < 8; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j) acc16[i][j] = f32x4{0.0f, 0.0f, 0.0f, 0.0f};
            p8_mainloop<C, true>(p, smem, row0, col0, st_begin, n, kchunks, wave_all, lane, acc16);
            if constexpr (SPLITK) {                    // fp32 partials of this K slice; the reduce kernels apply the epilogue
                float* slab = p.ws + (size_t)ks * p.M * p.N;
                const int r16 = lane & 15, g4 = lane >> 4;
#pragma unroll
                for (int nj = 0; nj < 4; ++nj) {
                    const int nn = col0 + wc * 64 + nj * 16 + r16;
                    if (nn >= p.N) continue;
#pragma unroll
                    for (int mi = 0; mi < 8; ++mi)
#pragma unroll
                        for (int e = 0; e < 4; ++e) {
                            const int m = row0 + wr * 128 + mi * 16 + 4 * g4 + e;
                            if (m < p.M) slab[(size_t)m * p.N + nn] = acc16[mi][nj][e];
                        }
                }
            } else {
                // 2-byte outputs with 16-byte aligned rows (every product of the engine): stores staged through LDS; otherwise the direct form
                bool staged = false;
                if constexpr (sizeof(T) == 2 && EPI != EPI_ARGMAX)
                    staged = (p.ldc & 7) == 0 && ((size_t)p.C & 15) == 0 && ((EPI == EPI_SWIGLU ? col0 >> 1 : col0) & 7) == 0 && !(p.force_cfg & 0x20000);
                if (staged) {
                    if constexpr (sizeof(T) == 2 && EPI != EPI_ARGMAX) {
                        __syncthreads();            // every wave is done with the stage ring: the epilogue tiles overlay it
                        p8_epilogue16_staged<T, EPI>(p, smem, row0, col0, wave_all, lane, acc16);
                    }
                } else {
                    p8_epilogue16<T, EPI>(p, row0, col0, wave_all, lane, acc16);
                }
            }
            return;
        }
    } else {
#pragma unroll
    for (int d = 0; d < D; ++d)
        if (d < n) issue(st_begin + d, d);
    int buf = 0, nbuf = D % C::NBUF;              // buffer of stage i / of stage i + D
    for (int i = 0; i < n_it; ++i) {
        if (i + D <= n) asm volatile("s_waitcnt vmcnt(%0)" ::"n"((D - 1) * PER_WAVE) : "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        if (KG > 1 && i >= n) continue;               // a K-group with one stage less keeps the barrier count of the others
        compute(buf, [&]() { if (i + D < n) issue(st_begin + i + D, nbuf); });
        buf = buf + 1 == C::NBUF ? 0 : buf + 1;
        nbuf = nbuf + 1 == C::NBUF ? 0 : nbuf + 1;
    }
    }
    if constexpr (KG > 1) {
        // Exchange between the two K-groups: group g finishes accumulator rows i with i / (MI / KG) == g, so it hands the OTHER rows to its
        // partner wave (same wave tile, same lane -> same (m, n) mapping: no transposition) through LDS and adds the partner's copy of its
        // own rows.  Every DMA has landed (vmcnt(0) above) and every fragment read has been consumed by an MFMA.
        static_assert(KG == 2, "two K-groups");
        constexpr int MH = MI / KG;
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __syncthreads();
        float* xch = (float*)smem;                    // [K-group][wave][MH * NJ tiles][4 quads][64 lanes][4]
        float* mine = xch + (size_t)((kg * WAVES + wave) * MH * NJ) * 1024;
        const float* theirs = xch + (size_t)(((1 - kg) * WAVES + wave) * MH * NJ) * 1024;
#pragma unroll
        for (int i = 0; i < MI; ++i) {
            if (i / MH == kg) continue;               // (wave-uniform; i stays a compile-time index)
#pragma unroll
            for (int j = 0; j < NJ; ++j)
#pragma unroll
                for (int q4 = 0; q4 < 4; ++q4)
                    *(float4*)(mine + (size_t)(((i % MH) * NJ + j) * 4 + q4) * 256 + lane * 4) =
                        make_float4(acc[i][j][4 * q4], acc[i][j][4 * q4 + 1], acc[i][j][4 * q4 + 2], acc[i][j][4 * q4 + 3]);
        }
        __syncthreads();
#pragma unroll
        for (int i = 0; i < MI; ++i) {
            if (i / MH != kg) continue;
#pragma unroll
            for (int j = 0; j < NJ; ++j)
#pragma unroll
                for (int q4 = 0; q4 < 4; ++q4) {
                    const float4 t = *(const float4*)(theirs + (size_t)(((i % MH) * NJ + j) * 4 + q4) * 256 + lane * 4);
                    acc[i][j][4 * q4] += t.x; acc[i][j][4 * q4 + 1] += t.y; acc[i][j][4 * q4 + 2] += t.z; acc[i][j][4 * q4 + 3] += t.w;
                }
        }
    }
    // accumulator rows this wave stores (all of them without K-groups)
    auto mine_i = [&](int i) { return KG == 1 || i / (MI / KG) == kg; };

    if (FP8) {           // dequantise (linear, so split-K partials are scaled too): per-row activation scale x per-row weight scale
#pragma unroll
        for (int j = 0; j < NJ; ++j) {
            const float sw = p.w_scale[min(col0 + wc * WCOLS + j * 32 + r32, p.N - 1)];
#pragma unroll
            for (int i = 0; i < MI; ++i) {
                if (!mine_i(i)) continue;
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[i][j][r] *= sw * p.a_scale[min(row0 + wr * WROWS + i * 32 + acc_row(r, lane), p.M - 1)];
            }
        }
    }
    if (SPLITK) {
        float* slab = p.ws + (size_t)ks * p.M * p.N;
#pragma unroll
        for (int i = 0; i < MI; ++i)
#pragma unroll
            for (int j = 0; j < NJ; ++j) {
                const int nn = col0 + wc * WCOLS + j * 32 + r32;
                if (nn >= p.N) continue;
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int m = row0 + wr * WROWS + i * 32 + acc_row(r, lane);
                    if (m < p.M) slab[(size_t)m * p.N + nn] = acc[i][j][r];
                }
            }
        return;
    }
    if constexpr (EPI == EPI_ARGMAX) {
        // Arg-max epilogue (lm_head of several envs decoded together): row m of the tile keeps (greatest logit, lowest column) over the
        // tile's columns.  Lane r32 holds column j * 32 + r32 of 16 rows: reduce over j in the lane, over the 32 lanes of a half with
        // xor-shuffles (the halves hold different rows), then over the WN waves through LDS.  First-max-wins needs the index compare
        // because the shuffle tree does not visit columns in order.
        static_assert(MI == 1 && KG == 1 && !SPLITK, "arg-max epilogue: 32-row tiles, unsplit");
        float bv[16]; int bi[16];
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int m = min(row0 + acc_row(r, lane), p.M - 1);
            const uint8_t* fl = p.pen_flags ? p.pen_flags + (size_t)p.pen_rows[m] * p.N : nullptr;
            float v = -INFINITY; int vi = 0x7FFFFFFF;
#pragma unroll
            for (int j = 0; j < NJ; ++j) {
                const int nn = col0 + wc * WCOLS + j * 32 + r32;
                float c = acc[0][j][r];
                if (fl && nn < p.N && fl[nn]) c = c < 0.0f ? c * p.pen : c / p.pen;
                if (nn < p.N && (c > v || (c == v && nn < vi))) { v = c; vi = nn; }
            }
#pragma unroll
            for (int o = 1; o < 32; o <<= 1) {
                const float ov = __shfl_xor(v, o, 64);
                const int oi = __shfl_xor(vi, o, 64);
                if (ov > v || (ov == v && oi < vi)) { v = ov; vi = oi; }
            }
            bv[r] = v; bi[r] = vi;
        }
        __syncthreads();                                   // every wave is done with the stage ring: reuse its first bytes
        float* sv = (float*)smem; int* si = (int*)(smem + C::WN * 32 * sizeof(float));
        if (r32 == 0) {
#pragma unroll
            for (int r = 0; r < 16; ++r) { sv[wc * 32 + acc_row(r, lane)] = bv[r]; si[wc * 32 + acc_row(r, lane)] = bi[r]; }
        }
        __syncthreads();
        if (tid < 32 && row0 + tid < p.M) {
            float v = sv[tid]; int vi = si[tid];
#pragma unroll
            for (int w = 1; w < C::WN; ++w) {
                const float ov = sv[w * 32 + tid]; const int oi = si[w * 32 + tid];
                if (ov > v || (ov == v && oi < vi)) { v = ov; vi = oi; }
            }
            const int tiles_n = p.launch_tiles;            // (one row tile: launch_tiles = column tiles)
            p.part_val[(size_t)(row0 + tid) * tiles_n + bn] = v;
            p.part_idx[(size_t)(row0 + tid) * tiles_n + bn] = vi;
        }
        return;
    }
    T* Cc = (T*)p.C;
    const T* bias = (const T*)p.bias;
    const T* res = (const T*)p.res;
    if (EPI == EPI_SWIGLU) {
#pragma unroll
        for (int jp = 0; jp < NJ / 2; ++jp) {                 // one [gate 32 | up 32] block per accumulator pair
            const int n_out = ((col0 + wc * WCOLS + jp * 64) >> 1) + r32;
            const bool ok_n = (col0 + wc * WCOLS + jp * 64 + 32 + r32) < p.N;
#pragma unroll
            for (int i = 0; i < MI; ++i) {
                if (!mine_i(i)) continue;
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int m = row0 + wr * WROWS + i * 32 + acc_row(r, lane);
                    if (m < p.M && ok_n) Cc[(size_t)m * p.ldc + n_out] = from_f32<T>(silu_f(acc[i][2 * jp][r]) * acc[i][2 * jp + 1][r]);
                }
            }
        }
        return;
    }
    if constexpr (VP) {
        // fused tail of the SigLIP QKV product (same values and layout as splitk_qkv_vitpack_kernel / vit_kv_pack_kernel): the k columns also
        // go to the K page of their (64-key tile, frame, head), the v columns transposed to the V^T page.  Padding channels and the keys past
        // S of the last tile are never written: the pools are zero-filled at allocation and every writer leaves zeros there.
        const VitPackArgs& vp = p.vp;
        constexpr int EPC_T = Elt<T>::PER_CHUNK;
        const int vHD = vp.head_dim, vHv = vp.heads * vHD, vHDP = ((((vHD + EPC_T - 1) / EPC_T) + 1) & ~1) * EPC_T, vVR = ((vHD + 31) / 32) * 32;
        const int v_nkv = vp.F * vp.heads;
#pragma unroll
        for (int i = 0; i < MI; ++i) {
            if (!mine_i(i)) continue;
#pragma unroll
            for (int j = 0; j < NJ; ++j) {
                const int nn = col0 + wc * WCOLS + j * 32 + r32;
                if (nn >= p.N) continue;
                const float bv = bias ? to_f32(bias[nn]) : 0.0f;
                const int part = nn / vHv, w_ = nn - part * vHv, head = w_ / vHD, dd = w_ - head * vHD;
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int m = row0 + wr * WROWS + i * 32 + acc_row(r, lane);
                    if (m >= p.M) continue;
                    const T tv = from_f32<T>(acc[i][j][r] + bv);
                    Cc[(size_t)m * p.ldc + nn] = tv;
                    if (part > 0) {
                        const int f = m / vp.S, srow = m - f * vp.S, tile = srow >> 6, key = srow & 63;
                        const size_t pg = (size_t)tile * v_nkv + (size_t)f * vp.heads + head;
                        if (part == 1) ((T*)vp.Kpool)[(pg * 64 + key) * vHDP + dd] = tv;
                        else ((T*)vp.Vpool)[(pg * vVR + dd) * 64 + key] = tv;
                    }
                }
            }
        }
        return;
    }
#pragma unroll
    for (int i = 0; i < MI; ++i) {
        if (!mine_i(i)) continue;
#pragma unroll
        for (int j = 0; j < NJ; ++j) {
            const int nn = col0 + wc * WCOLS + j * 32 + r32;
            if (nn >= p.N) continue;
            const float bv = bias ? to_f32(bias[nn]) : 0.0f;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int m = row0 + wr * WROWS + i * 32 + acc_row(r, lane);
                if (m >= p.M) continue;
                float v = epi_act<T, EPI>(acc[i][j][r] + bv);
                if (res) {
                    const int rr = p.res_mod > 0 ? m % p.res_mod : m;
                    v += to_f32(res[(size_t)rr * p.ldr + nn]);
                }
                Cc[(size_t)m * p.ldc + nn] = from_f32<T>(v);
            }
        }
    }
}

// four consecutive elements as one 8- / 16-byte access when the address allows it (the rows of this kernel start at multiples of 4 columns)
template <typename T> SVLN_DEV void load4(const T* ptr, bool vec, float (&out)[4]) {
    if (vec) {
        if constexpr (sizeof(T) == 2) {
            const uint2 w = *(const uint2*)ptr;
            out[0] = __uint_as_float(w.x << 16); out[1] = __uint_as_float(w.x & 0xFFFF0000u);
            out[2] = __uint_as_float(w.y << 16); out[3] = __uint_as_float(w.y & 0xFFFF0000u);
        } else {
            const float4 w = *(const float4*)ptr;
            out[0] = w.x; out[1] = w.y; out[2] = w.z; out[3] = w.w;
        }
    } else {
#pragma unroll
        for (int e = 0; e < 4; ++e) out[e] = to_f32(ptr[e]);
    }
}
template <typename T> SVLN_DEV void store4(T* ptr, bool vec, const T (&v)[4]) {
    if (vec) {
        if constexpr (sizeof(T) == 2) {
            const uint16_t* h = (const uint16_t*)v;
            *(uint2*)ptr = make_uint2((unsigned)h[0] | ((unsigned)h[1] << 16), (unsigned)h[2] | ((unsigned)h[3] << 16));
        } else {
            *(float4*)ptr = make_float4(to_f32(v[0]), to_f32(v[1]), to_f32(v[2]), to_f32(v[3]));
        }
    } else {
#pragma unroll
        for (int e = 0; e < 4; ++e) ptr[e] = v[e];
    }
}
template <typename T> SVLN_DEV bool vec4_ok(const void* base, size_t ld) { return ((size_t)base % (4 * sizeof(T))) == 0 && ld % 4 == 0; }

// sum split-K slabs + epilogue.  One thread = 4 consecutive output columns of one row.
template <typename T, int EPI>
__global__ __launch_bounds__(256) void splitk_epilogue_kernel(GemmArgs p) {
    const int n_out_total = EPI == EPI_SWIGLU ? p.N / 2 : p.N;
    // this launch covers output columns [n_out_begin, n_out_total) (tail launches start past column 0)
    const int n_out_begin = EPI == EPI_SWIGLU ? p.tile_base * (Cfg256::BN / 2) : p.tile_base * Cfg256::BN;
    const int quads = (n_out_total - n_out_begin + 3) / 4;
    const size_t total = (size_t)p.M * quads;
    const size_t slab = (size_t)p.M * p.N;
    T* Cc = (T*)p.C;
    const T* bias = (const T*)p.bias;
    const T* res = (const T*)p.res;
    const bool cvec = vec4_ok<T>(Cc, p.ldc);
    // (M * quads < 2^31 for every product of the engine: 32-bit index arithmetic; the launcher asserts it)
    for (unsigned idx = blockIdx.x * 256u + threadIdx.x; idx < (unsigned)total; idx += gridDim.x * 256u) {
        const int m = (int)(idx / (unsigned)quads), n0 = n_out_begin + (int)(idx - (unsigned)m * (unsigned)quads) * 4;
        float out[4];
        if (EPI == EPI_SWIGLU) {
            // output j lives in packed columns (j/32)*64 + j%32 (gate) and +32 (up); 4 consecutive j share a block
            const int gcol = (n0 >> 5) * 64 + (n0 & 31);
            float g[4] = {0, 0, 0, 0}, u[4] = {0, 0, 0, 0};
            for (int s = 0; s < p.nsplit; ++s) {
                const float4 gv = *(const float4*)(p.ws + s * slab + (size_t)m * p.N + gcol);
                const float4 uv = *(const float4*)(p.ws + s * slab + (size_t)m * p.N + gcol + 32);
                g[0] += gv.x; g[1] += gv.y; g[2] += gv.z; g[3] += gv.w;
                u[0] += uv.x; u[1] += uv.y; u[2] += uv.z; u[3] += uv.w;
            }
#pragma unroll
            for (int e = 0; e < 4; ++e) out[e] = silu_f(g[e]) * u[e];
        } else {
            float a[4] = {0, 0, 0, 0};
            if (n0 + 3 < p.N) {
                for (int s = 0; s < p.nsplit; ++s) {
                    const float4 v = *(const float4*)(p.ws + s * slab + (size_t)m * p.N + n0);
                    a[0] += v.x; a[1] += v.y; a[2] += v.z; a[3] += v.w;
                }
            } else {
                for (int s = 0; s < p.nsplit; ++s)
                    for (int e = 0; e < 4; ++e)
                        if (n0 + e < p.N) a[e] += p.ws[s * slab + (size_t)m * p.N + n0 + e];
            }
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int n = n0 + e;
                if (n >= p.N) { out[e] = 0.0f; continue; }
                float v = epi_act<T, EPI>(a[e] + (bias ? to_f32(bias[n]) : 0.0f));
                if (res) {
                    const int rr = p.res_mod > 0 ? m % p.res_mod : m;
                    v += to_f32(res[(size_t)rr * p.ldr + n]);
                }
                out[e] = v;
            }
        }
        if (n0 + 3 < n_out_total) {
            const T o4[4] = {from_f32<T>(out[0]), from_f32<T>(out[1]), from_f32<T>(out[2]), from_f32<T>(out[3])};
            store4<T>(Cc + (size_t)m * p.ldc + n0, cvec, o4);
        } else {
#pragma unroll
            for (int e = 0; e < 4; ++e)
                if (n0 + e < n_out_total) Cc[(size_t)m * p.ldc + n0 + e] = from_f32<T>(out[e]);
        }
    }
}

// Split-K reduce of one output row per workgroup (+ bias + residual), fused with the norm of the finished row:
//   C[m] = round(sum_s slab[s][m] + bias + res[m]);
//   Y[m] = g * (C[m] * rsqrt(mean(C[m]^2) + eps))                       (RMSNorm: Qwen2 post_attention / input_layernorm)
//   Y[m] = (C[m] - mean) * rsqrt(var + eps) * g + b   when norm_b != 0   (LayerNorm: SigLIP ln2 / next layer's ln1)
// N <= 4096, N % 4 == 0; blockDim = the row's quads rounded up to whole waves.  Same arithmetic as
// splitk_epilogue_kernel<EPI_NONE> followed by rmsnorm_kernel / layernorm_kernel on the rounded row.
template <typename T>
__global__ __launch_bounds__(1024) void splitk_rownorm_kernel(GemmArgs p) {
    __shared__ float red[2][16];
    const int m = blockIdx.x, tid = threadIdx.x, n0 = tid * 4, nw = blockDim.x >> 6;      // one thread = 4 consecutive columns
    const size_t slab = (size_t)p.M * p.N;
    T* Cc = (T*)p.C;
    const T* bias = (const T*)p.bias;
    const T* res = (const T*)p.res;
    const T* g = (const T*)p.norm_w;
    const T* nb = (const T*)p.norm_b;
    T* Y = (T*)p.norm_out;
    const int rr = p.res_mod > 0 ? m % p.res_mod : m;
    const bool live = n0 < p.N;
    float v[4] = {0, 0, 0, 0};
    float s1 = 0.0f;                 // sum (LayerNorm) or sum of squares (RMSNorm)
    if (live) {
        float a[4] = {0, 0, 0, 0};
        const float* src = p.ws + (size_t)m * p.N + n0;
#pragma unroll 4
        for (int s = 0; s < p.nsplit; ++s) {
            const float4 t = *(const float4*)(src + s * slab);
            a[0] += t.x; a[1] += t.y; a[2] += t.z; a[3] += t.w;
        }
        float bf[4] = {0, 0, 0, 0}, rf[4] = {0, 0, 0, 0};
        if (bias) load4<T>(bias + n0, vec4_ok<T>(bias, 0), bf);
        if (res) load4<T>(res + (size_t)rr * p.ldr + n0, vec4_ok<T>(res, p.ldr), rf);
        T t4[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            float x = a[e] + bf[e];
            if (res) x += rf[e];
            t4[e] = from_f32<T>(x);
            v[e] = to_f32(t4[e]);
            s1 = nb ? s1 + v[e] : fmaf(v[e], v[e], s1);
        }
        store4<T>(Cc + (size_t)m * p.ldc + n0, vec4_ok<T>(Cc, p.ldc), t4);
    }
    auto block_sum = [&](float x, int k) {
        x = wave_sum(x);
        if ((tid & 63) == 0) red[k][tid >> 6] = x;
        __syncthreads();
        float tot = 0.0f;
        for (int w = 0; w < nw; ++w) tot += red[k][w];
        return tot;
    };
    const float t1 = block_sum(s1, 0);
    if (nb) {
        const float mu = t1 / (float)p.N;
        float s2 = 0.0f;
        if (live) {
#pragma unroll
            for (int e = 0; e < 4; ++e) s2 += (v[e] - mu) * (v[e] - mu);
        }
        const float sc = rsqrtf(block_sum(s2, 1) / (float)p.N + p.norm_eps);
        if (live) {
            float gf[4], nf[4];
            load4<T>(g + n0, vec4_ok<T>(g, 0), gf);
            load4<T>(nb + n0, vec4_ok<T>(nb, 0), nf);
            T y4[4];
#pragma unroll
            for (int e = 0; e < 4; ++e) y4[e] = from_f32<T>((v[e] - mu) * sc * gf[e] + nf[e]);
            store4<T>(Y + (size_t)m * p.N + n0, vec4_ok<T>(Y, p.N), y4);
        }
    } else {
        const float sc = rsqrtf(t1 / (float)p.N + p.norm_eps);
        float yq[4] = {0, 0, 0, 0};
        if (live) {
            float gf[4];
            load4<T>(g + n0, vec4_ok<T>(g, 0), gf);
            T y4[4];
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                y4[e] = from_f32<T>(gf[e] * (v[e] * sc));
                yq[e] = to_f32(y4[e]);
            }
            store4<T>(Y + (size_t)m * p.N + n0, vec4_ok<T>(Y, p.N), y4);
        }
        if constexpr (sizeof(T) == 2) {
            if (p.norm_q8) {              // e4m3 copy of the row (quant_fp8_rows_kernel's arithmetic on the rounded values)
                float amax = fmaxf(fmaxf(fabsf(yq[0]), fabsf(yq[1])), fmaxf(fabsf(yq[2]), fabsf(yq[3])));
                amax = wave_max(amax);
                __syncthreads();          // red[0] is free again (every thread has read the totals)
                if ((tid & 63) == 0) red[0][tid >> 6] = amax;
                __syncthreads();
                amax = 0.0f;
                for (int w = 0; w < nw; ++w) amax = fmaxf(amax, red[0][w]);
                const float qs = amax > 0.0f ? amax / 448.0f : 1.0f;
                if (tid == 0) p.norm_q8_scale[m] = qs;
                if (live) {
                    const float inv = 1.0f / qs;
                    float f[4];
#pragma unroll
                    for (int e = 0; e < 4; ++e) f[e] = fminf(fmaxf(yq[e] * inv, -448.0f), 448.0f);
                    int w8 = 0;
                    w8 = __builtin_amdgcn_cvt_pk_fp8_f32(f[0], f[1], w8, false);
                    w8 = __builtin_amdgcn_cvt_pk_fp8_f32(f[2], f[3], w8, true);
                    *(unsigned*)((uint8_t*)p.norm_q8 + (size_t)m * p.N + n0) = (unsigned)w8;
                }
            }
        }
    }
}

// Split-K reduce of the prefill QKV product fused with its tail (modeling_qwen2.py:195-235): C = round(sum_s slab + bias), RoPE on the
// q and k heads (cos / sin table, rotate_half), q back into C, roped k and v^T appended to the paged KV cache -- splitk_epilogue_kernel
// followed by rope_kv_kernel in one pass (same arithmetic, same intermediate rounding to T).  grid (rows, ceil((nq + 2 nkv) / 4)), 256 threads:
// a wave per head, lane d owns columns d and d + 64 of the 128-wide head.
template <typename T>
__global__ __launch_bounds__(256) void splitk_qkv_rope_kernel(GemmArgs p, RopeKvArgs r) {
    const int i = blockIdx.x, hd = blockIdx.y * 4 + (threadIdx.x >> 6), d = threadIdx.x & 63;      // a wave per head, four heads per workgroup
    if (hd >= r.nq + 2 * r.nkv) return;
    const size_t slab = (size_t)p.M * p.N;
    const int c0 = hd * 128 + d;
    const float* src = p.ws + (size_t)i * p.N + c0;
    float a1 = 0.0f, a2 = 0.0f;
    for (int s = 0; s < p.nsplit; ++s) { a1 += src[s * slab]; a2 += src[s * slab + 64]; }
    const T* bias = (const T*)p.bias;
    if (bias) { a1 += to_f32(bias[c0]); a2 += to_f32(bias[c0 + 64]); }
    const float x1 = to_f32(from_f32<T>(a1)), x2 = to_f32(from_f32<T>(a2));      // the unfused path stores C in T before the rotation
    const int pos = (r.dyn_pos ? *r.dyn_pos : r.P) + i;
    const int page = r.page_table[pos >> 6], off = pos & 63;
    T* row = (T*)p.C + (size_t)i * p.ldc + (size_t)hd * 128;
    if (hd < r.nq + r.nkv) {
        const float c = r.rope_tab[(size_t)pos * 128 + d], sn = r.rope_tab[(size_t)pos * 128 + 64 + d];
        const float o1 = x1 * c - x2 * sn, o2 = x2 * c + x1 * sn;
        if (hd < r.nq) {
            row[d] = from_f32<T>(o1);
            row[d + 64] = from_f32<T>(o2);
        } else {
            T* kr = (T*)r.Kpool + (((size_t)page * r.nkv + (hd - r.nq)) * 64 + off) * 128;
            kr[d] = from_f32<T>(o1);
            kr[d + 64] = from_f32<T>(o2);
        }
    } else {
        T* vt = (T*)r.Vpool + ((size_t)page * r.nkv + (hd - r.nq - r.nkv)) * 128 * 64;
        vt[(size_t)d * 64 + off] = from_f32<T>(x1);
        vt[(size_t)(d + 64) * 64 + off] = from_f32<T>(x2);
    }
}

// Split-K reduce of the SigLIP QKV product fused with the K / V^T packing of the ViT attention: C = round(sum_s slab + bias) for the
// q | k | v columns of one (64-key tile, frame * head), then the k rows go to the K page (zero padded to HDP) and the v rows through an
// LDS tile to the transposed V page -- splitk_epilogue_kernel followed by vit_kv_pack_kernel (misc.hip) in one pass, same values.
// grid (key tiles, F * heads, 3 parts), 256 threads.  C keeps all three parts (the attention reads q from it).
template <typename T>
__global__ __launch_bounds__(256) void splitk_qkv_vitpack_kernel(GemmArgs p, VitPackArgs v) {
    extern __shared__ __attribute__((aligned(16))) char vp_smem[];
    constexpr int EPC = Elt<T>::PER_CHUNK;
    T* vt = (T*)vp_smem;                                    // [64 keys][HD + EPC]
    const int tile = blockIdx.x, kh = blockIdx.y;
    const int f = kh / v.heads, head = kh % v.heads, HD = v.head_dim, Hv = v.heads * HD;
    const int hdc = (((HD + EPC - 1) / EPC) + 1) & ~1, HDP = hdc * EPC, VROWS = ((HD + 31) / 32) * 32;
    const int nkv = v.F * v.heads, pitch = HD + EPC, q4 = HD / 4;
    const size_t slab = (size_t)p.M * p.N;
    const T* bias = (const T*)p.bias;
    T* Cc = (T*)p.C;
    T* kp = (T*)v.Kpool + ((size_t)tile * nkv + kh) * 64 * HDP;
    T* vp = (T*)v.Vpool + ((size_t)tile * nkv + kh) * VROWS * 64;
    // blockIdx.z = part (0 = q, 1 = k, 2 = v): 576 workgroups per frame keep enough slab loads in flight; one thread = 4 consecutive
    // columns of one key row
    const int part = blockIdx.z;
    for (int e = threadIdx.x; e < 64 * q4; e += 256) {
        const int key = e / q4, c4 = (e - key * q4) * 4;
        const int srow = tile * 64 + key, col = part * Hv + head * HD + c4;
        T o[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) o[j] = from_f32<T>(0.0f);
        if (srow < v.S) {
            const size_t row = (size_t)f * v.S + srow;
            const float* src = p.ws + row * p.N + col;
            float a[4] = {0, 0, 0, 0};
            for (int s = 0; s < p.nsplit; ++s) {
                const float4 t = *(const float4*)(src + s * slab);
                a[0] += t.x; a[1] += t.y; a[2] += t.z; a[3] += t.w;
            }
#pragma unroll
            for (int j = 0; j < 4; ++j) o[j] = from_f32<T>(a[j] + (bias ? to_f32(bias[col + j]) : 0.0f));
#pragma unroll
            for (int j = 0; j < 4; ++j) Cc[row * p.ldc + col + j] = o[j];
        }
        if (part == 1) {
#pragma unroll
            for (int j = 0; j < 4; ++j) kp[(size_t)key * HDP + c4 + j] = o[j];
        } else if (part == 2) {
#pragma unroll
            for (int j = 0; j < 4; ++j) vt[key * pitch + c4 + j] = o[j];
        }
    }
    if (part == 0) return;
    if (part == 1) {
        for (int e = threadIdx.x; e < 64 * (HDP - HD); e += 256) {          // zero padding of the K rows
            const int key = e / (HDP - HD), c = HD + e % (HDP - HD);
            kp[(size_t)key * HDP + c] = from_f32<T>(0.0f);
        }
        return;
    }
    __syncthreads();
    constexpr int KC = 64 / EPC;                            // key chunks per Vt row
    for (int e = threadIdx.x; e < VROWS * KC; e += 256) {
        const int d = e / KC, kc = e % KC;
        T out[EPC];
#pragma unroll
        for (int j = 0; j < EPC; ++j) out[j] = d < HD ? vt[(kc * EPC + j) * pitch + d] : from_f32<T>(0.0f);
        *(uint4*)(vp + (size_t)d * 64 + kc * EPC) = *(const uint4*)out;
    }
}

// EVERY launch of gemm_glds_kernel goes through here: the block size and the dynamic-LDS size come from the same TileCfg as the kernel
// instantiation (its __launch_bounds__ and its LDS carve), so a geometry the code object cannot take -- which hipLaunchKernel does not
// report: the packet processor rejects the dispatch and the runtime aborts the process (the round-2 test_gemm abort) --
// cannot be written at a call site.  gemm_init_attrs() registers the same pairs with set_max_lds, which checks them against the code
// object's attributes at engine creation.
template <typename T, int EPI, typename C, bool SPLITK, typename TA = T, bool NTW = false, bool VP = false>
void launch_tile(hipStream_t s, const GemmArgs& a, int wgs) {
    hipLaunchKernelGGL((gemm_glds_kernel<T, EPI, C, SPLITK, TA, NTW, VP>), dim3(wgs), dim3(C::THREADS), C::LDS_BYTES, s, a);
}

template <typename T, int EPI, typename C, bool SPLITK> void launch_cfg(hipStream_t s, const GemmArgs& a, int nsplit) {
    const int wgs = a.launch_tiles * nsplit;
    if (wgs <= 0) return;
    constexpr bool HAS_NTW = std::is_same<C, Cfg256>::value || std::is_same<C, CfgSkinny>::value || std::is_same<C, Cfg256N64>::value;      // the single-row-tile configurations
    if constexpr (sizeof(T) == 2 && (EPI == EPI_NONE || EPI == EPI_SWIGLU)) {
        if (a.a_scale) {                                   // e4m3 operands (opt-in; the LLM linears: plain and SwiGLU epilogues)
            if constexpr (HAS_NTW) {
                if (a.nt_w) {
                    launch_tile<T, EPI, C, SPLITK, fp8_t, true>(s, a, wgs);
                    return;
                }
            }
            launch_tile<T, EPI, C, SPLITK, fp8_t>(s, a, wgs);
            return;
        }
    }
    if constexpr (HAS_NTW) {
        if (a.nt_w) {
            launch_tile<T, EPI, C, SPLITK, T, true>(s, a, wgs);
            return;
        }
    }
    launch_tile<T, EPI, C, SPLITK>(s, a, wgs);
}

// sum the split-K slabs and apply the epilogue (returns true when the reduce also emitted the following norm / the fused tail)
template <typename T, int EPI> bool launch_reduce(hipStream_t s, const GemmArgs& a);
template <typename T, int EPI, typename C = Cfg256> bool launch_split(hipStream_t s, GemmArgs a, int S) {
    static_assert(C::BN == Cfg256::BN, "tail launches (tile_base > 0) assume 128-column tiles");
    a.nsplit = S;
    launch_cfg<T, EPI, C, true>(s, a, S);
    return launch_reduce<T, EPI>(s, a);
}
template <typename T, int EPI> bool launch_reduce(hipStream_t s, const GemmArgs& a) {
    if (EPI == EPI_NONE && a.norm_out && a.norm_w && a.tile_base == 0 && a.N <= 4096 && a.N % 4 == 0) {
        hipLaunchKernelGGL((splitk_rownorm_kernel<T>), dim3(a.M), dim3(((a.N / 4 + 63) / 64) * 64), 0, s, a);
        return true;
    }
    if (EPI == EPI_NONE && a.vitpack && a.tile_base == 0 && !a.res && a.N == 3 * a.vitpack->heads * a.vitpack->head_dim &&
        a.M == a.vitpack->F * a.vitpack->S && a.vitpack->head_dim % 4 == 0) {
        const VitPackArgs& v = *a.vitpack;
        const size_t lds = (size_t)64 * (v.head_dim + Elt<T>::PER_CHUNK) * sizeof(T);
        hipLaunchKernelGGL((splitk_qkv_vitpack_kernel<T>), dim3((v.S + 63) / 64, v.F * v.heads, 3), dim3(256), lds, s, a, v);
        return true;
    }
    if (EPI == EPI_NONE && a.rope && a.tile_base == 0 && !a.res && a.N == (a.rope->nq + 2 * a.rope->nkv) * 128 && a.M == a.rope->T) {
        hipLaunchKernelGGL((splitk_qkv_rope_kernel<T>), dim3(a.M, (a.N / 128 + 3) / 4), dim3(256), 0, s, a, *a.rope);
        return true;
    }
    const int n_out = EPI == EPI_SWIGLU ? a.N / 2 : a.N;
    const int n_begin = EPI == EPI_SWIGLU ? a.tile_base * (Cfg256::BN / 2) : a.tile_base * Cfg256::BN;
    const size_t work = (size_t)a.M * ((n_out - n_begin) / 4 + 1);
    int grid = (int)((work + 255) / 256);
    if (grid > 4096) grid = 4096;
    if (work >= ((size_t)1 << 31)) throw std::runtime_error("split-K reduce: more than 2^31 output quads");      // (the kernel indexes in 32 bits)
    hipLaunchKernelGGL((splitk_epilogue_kernel<T, EPI>), dim3(grid), dim3(256), 0, s, a);
    return false;
}

template <typename T, int EPI> bool launch_epi(hipStream_t s, GemmArgs a) {
    if (a.M <= 0 || a.N <= 0) return false;
    a.vp_on = 0;
    // Column tiles fastest when the activations are the big operand AND larger than the L2s together (32 MB): the nine-frame ViT fc2
    // (6561 x 4304 activations = 56 MB against 10 MB of weights) 95 -> 85 us; the other nine-frame products (15 MB of activations) measured
    // 1-4 % slower that way and keep the row-tiles-fastest order.  force_cfg | 0x1000 = row tiles fastest, | 0x10000 = column tiles fastest (tests)
    a.bn_fast = (((size_t)a.M * a.K * sizeof(T) > ((size_t)32 << 20) && a.M > a.N && !(a.force_cfg & 0x1000)) || (a.force_cfg & 0x10000)) ? 1 : 0;
    a.nt_w = a.M <= 256 ? 1 : 0;          // the heuristics below give such products ONE row tile (256x128 or 32x128 tiles)
    const int EPC = a.a_scale ? 16 : Elt<T>::PER_CHUNK;      // 16-byte chunks of K: e4m3 operands hold 16 values per chunk
    a.tile_base = 0;
    if (a.force_split > 1) {      // a forced split obeys the same workspace / shape limits as the heuristic ones
        const bool ok = a.ws != nullptr && a.N % 4 == 0 && !(EPI == EPI_SWIGLU && a.N % 64 != 0);
        if (!ok) a.force_split = 1;
        while (a.force_split > 1 && (size_t)a.force_split * a.M * a.N > a.ws_elems) --a.force_split;
    }
    // Tile choice (measured on MI355X: tools/kbench.py for single launches, the kernel trace of bench.py for the choice inside a turn):
    //   M <= 256            -> 256x128 tiles + split-K: one row tile, every weight byte staged once
    //   M  > 256, >= 256 tiles of 128x128 (a full round at two workgroups per CU) -> 128x128 tiles, no split
    //   otherwise (less than a round of 128x128 tiles: every one-frame ViT product, o/down at T = 376) -> 256x128 tiles + split-K: inside the
    //   turn the K-split launch + reduce beats the half-empty unsplit launch (one-frame ViT fc1: 16.8 + 10.0 us against 31.4; qkv equal)
    //   large M AND N (M > 512 and >= 140 tiles of 256x256: window-restart gate/up, batched-env prefill, 9-frame ViT fc1/qkv) -> 256x256 tiles,
    //   wave tile 128x64: twice the MFMAs per stage and barrier of the 128x128 kernel.  With two row tiles (the 376-row first turn of an
    //   episode) the second one is mostly padding: 128x128 tiles take its gate/up from 192 to ~140 us (prefill of that turn 10.5 -> 9.0 ms)
    // M <= 32 (several envs decoded in lockstep): 32x128 tiles, 2 waves, three 20 KB stages in flight per workgroup and two
    // workgroups per CU -- a weight stream through LDS-DMA with 4 MFMAs per stage, K split when there are few column tiles
    if (a.M <= 32 && ((a.force_cfg & 0xFFF) == 0 || (a.force_cfg & 0xFFF) == 32)) {
        const int tiles_n = (a.N + 127) / 128;
        const int stages = (a.K / EPC + CfgSkinny::CH - 1) / CfgSkinny::CH;
        const bool can_split = a.ws != nullptr && a.N % 4 == 0 && !(EPI == EPI_SWIGLU && a.N % 64 != 0);
        int S = 1;
        if (a.force_split > 0) S = a.force_split;
        else if (tiles_n < 192 && can_split) {
            S = 512 / tiles_n;
            if (S > stages / 2) S = stages / 2 < 1 ? 1 : stages / 2;
            if (S > 16) S = 16;
            while (S > 1 && (size_t)S * a.M * a.N > a.ws_elems) --S;
        }
        a.launch_tiles = tiles_n;
        if (S <= 1) { a.nsplit = 1; launch_cfg<T, EPI, CfgSkinny, false>(s, a, 1); return false; }
        return launch_split<T, EPI, CfgSkinny>(s, a, S);
    }
    const int tilesbig = ((a.M + 255) / 256) * ((a.N + 255) / 256);
    // (>= 140 tiles: the T = 1952 qkv product, 8 x 18 tiles on 256 CUs, runs 85 us on 256x256 tiles against 101 on 576 tiles of 128x128;
    //  o / down at 112 tiles stay on 128x128: 375 against 308 us for down)
    // Long-K products with fewer 256x256 tiles than CUs (down_proj at T = 1952: 8 x 14 tiles, 296 K tiles): two K slices per tile on the
    // 8-phase kernel + the slab reduce (which also emits the following RMSNorm) instead of 448 tiles of 128x128
    if constexpr (std::is_same<T, bf16>::value) {
        // (M <= 256 long-K products -- down_proj of a steady prefill -- were tried on 256x256 tiles with 6 / 9 / 18 K slices in round 4, to halve
        //  the re-staging of their 8 MB activation panel: prefill +0.03 .. +0.4 ms per turn in the in-turn A/B, so they stay on 128x128 split-K)
        const int ktiles = (a.K / EPC + 7) / 8;
        const bool can_split = a.ws != nullptr && a.N % 4 == 0 && !(EPI == EPI_SWIGLU && a.N % 64 != 0) && (size_t)2 * a.M * a.N <= a.ws_elems;
        // (o_proj at T = 1952 -- K = 3584, 56 K tiles -- on the same two slices: restart turn 47.3 / 46.6 and 47.4 / 47.1 ms against 48.2 / 47.2 and
        //  47.6 / 47.1 on 128x128 tiles + a separate RMSNorm in two alternating rounds: inside the noise, not taken)
        if (((tilesbig >= 96 && tilesbig <= 128 && a.M > 512 && ktiles >= 128 && a.force_split == 0 && (a.force_cfg & 0xFFF) == 0) || (a.force_cfg & 0xFFF) == 258) &&
            a.zeros && !a.a_scale && can_split) {
            a.nsplit = 2;
            a.launch_tiles = tilesbig;
            launch_tile<T, EPI, Cfg8P, true>(s, a, tilesbig * 2);
            return launch_reduce<T, EPI>(s, a);
        }
    }
    if ((tilesbig >= 140 && a.M > 512 && a.zeros && a.force_split == 0 && (a.force_cfg & 0xFFF) == 0) || (a.force_cfg & 0xFFF) == 256) {
        a.nsplit = 1;
        a.launch_tiles = tilesbig;
        if constexpr (std::is_same<T, bf16>::value) {
            // 8-phase schedule with 16x16x32 MFMAs (bf16 operands).  Measured inside the turn (tools/ab_lib.sh, three alternating rounds):
            // window-restart turn 55.2 ms with the stage ring, 55.5 with the 8-phase schedule on 32x32x16 MFMAs, 52.3 with it on
            // 16x16x32 -- the schedule alone is worth nothing here (the tile is bound by the CU's staging rate), the MFMA shape is: the
            // chip holds a higher clock on it under sustained load.  force_cfg 256 | 0x4000 = stage ring, | 0x8000 = 32x32x16 form (tests)
            if (!a.a_scale && a.zeros && !(a.force_cfg & 0x4000)) {
                if (a.force_cfg & 0x8000) launch_tile<T, EPI, Cfg8P32, false>(s, a, tilesbig);
                else launch_tile<T, EPI, Cfg8P, false>(s, a, tilesbig);
                return false;
            }
        }
        launch_cfg<T, EPI, CfgBig, false>(s, a, 1);
        return false;
    }

    const int fc = a.force_cfg & 0xFFF;
    const int tiles128 = ((a.M + 127) / 128) * ((a.N + 127) / 128);
    // 64x64 tiles (no K split, no fp32 slabs) for several row tiles, few 128-wide column tiles and a short K -- the one-frame ViT out_proj,
    // 729 x 1152 x 1152 -- are only reachable through force_cfg: on isolated launches they win (12.7 us against 14.1 + 6.2 for split +
    // reduce), but inside the turn the caller then has to run the following LayerNorm itself and the pair measured 19.4 + 5.0 us against
    // 10.9 + 10.2 for the K-split product whose reduce emits the norm (one-frame ViT 3.69 -> 3.60 ms).
    // (Also measured and NOT kept: a 4-deep ring for one-round 128x128 launches, 19.9 vs 19.4 us on ViT qkv; issuing the LDS-DMA
    //  pieces of the next stage one by one between the MFMAs instead of as a burst ahead of them: 5-25 % slower on every config.)
    // 256x64 tiles (8 waves stacked along M, the CDNA guide's tile for M = 256 projections) for the short-K, few-column products of a
    // steady prefill -- q|k|v (N 4608) and o_proj (N 3584) at K = 3584: 72 / 56 tiles x 3-4 K slices instead of 36 / 28 tiles x 7-9, i.e.
    // half the fp32 slab traffic for twice the re-staged activation bytes.  Measured inside the turn, two alternating rounds
    // (steady prefill per turn): both products -0.11 / -0.14 ms, q|k|v alone -0.08 / 0.00, with down_proj (K = 18944) as well +0.03 /
    // +0.11, gate/up +0.44 / +0.49 (one round of 128-wide tiles is what that product wants); a forced 2-way split +1.05.
    const bool n64 = fc == 0 && a.force_split == 0 && EPI == EPI_NONE && a.M > 32 && a.M <= 256 && a.N <= 4608 && a.N >= 1024 && a.K <= 4096 && a.K >= 1024;
    // ... and the one-frame SigLIP out_proj (729 x 1152 x 1152, three row tiles: weights stay cached, not nt): 54 tiles x 3 K slices against
    // 27 tiles x 9, a third of the fp32 slabs the LayerNorm-emitting reduce reads: vision per turn 4.10 -> 4.00 ms in two alternating
    // rounds; fc2 (K = 4304) measured equal at 4 slices and +0.25 ms at 6, and keeps the 256x128 tile.
    const bool vit64 = fc == 0 && a.force_split == 0 && EPI == EPI_NONE && a.M > 512 && a.M <= 768 && a.N <= 1280 && a.K <= 1280 && a.norm_out != nullptr;
    if (vit64) a.force_split = 3;
    if ((fc == 264 || n64 || vit64) && a.zeros && !a.a_scale) {
        // K split so that about one round of workgroups exists (force_split overrides); several row tiles (forced only): weights cached, not nt
        const int tiles = ((a.M + 255) / 256) * ((a.N + 63) / 64), st = (a.K / EPC + Cfg256N64::CH - 1) / Cfg256N64::CH;
        const bool can_split = a.ws != nullptr && a.N % 4 == 0 && !(EPI == EPI_SWIGLU && a.N % 64 != 0);
        int S = a.force_split > 0 ? a.force_split : 256 / tiles;
        if (S < 1) S = 1;
        if (S > st / 2) S = st / 2 < 1 ? 1 : st / 2;
        if (!can_split) S = 1;
        while (S > 1 && (size_t)S * a.M * a.N > a.ws_elems) --S;
        a.launch_tiles = tiles;
        if (S <= 1) { a.nsplit = 1; launch_cfg<T, EPI, Cfg256N64, false>(s, a, 1); return false; }
        a.nsplit = S;
        launch_cfg<T, EPI, Cfg256N64, true>(s, a, S);
        return launch_reduce<T, EPI>(s, a);
    }
    if (fc == 64 && a.zeros) {
        a.nsplit = 1;
        a.launch_tiles = ((a.M + 63) / 64) * ((a.N + 63) / 64);
        launch_cfg<T, EPI, Cfg64, false>(s, a, 1);
        return false;
    }
    // One round (or less) of 128x128 tiles over more than one row tile, with an epilogue that needs the finished value and no norm
    // fused into a slab reduce -- the one-frame ViT qkv (K / V^T packing) and fc1 (GELU), the projector products: two K-groups inside the
    // workgroup (Cfg128K2) instead of a K split over workgroups + fp32 slabs + a reduce launch.
    const int stages128 = (a.K / EPC + Cfg128::CH - 1) / Cfg128::CH;
    const bool kgroups = a.force_split == 0 &&
                         ((fc == 0 && a.M > 256 && tiles128 >= 96 && tiles128 <= 256 && stages128 >= 8 && !a.norm_out && !a.rope) || fc == 129);
    if (kgroups) {
        a.nsplit = 1;
        a.launch_tiles = tiles128;
        if (EPI == EPI_NONE && a.vitpack && !a.res && a.N == 3 * a.vitpack->heads * a.vitpack->head_dim && a.M == a.vitpack->F * a.vitpack->S) {
            a.vp = *a.vitpack; a.vp_on = 1;
        }
        if constexpr (EPI == EPI_NONE) {
            if (a.vp_on && !a.a_scale) {
                launch_tile<T, EPI_NONE, Cfg128K2, false, T, false, true>(s, a, tiles128);
                return true;
            }
        }
        a.vp_on = 0;
        launch_cfg<T, EPI, Cfg128K2, false>(s, a, 1);
        return false;
    }
    const bool want128 = a.M > 256 && tiles128 >= 256;
    if ((want128 && a.force_split == 0) || fc == 128) {
        a.nsplit = 1;
        a.launch_tiles = tiles128;
        if (tiles128 > 256) launch_cfg<T, EPI, Cfg128L, false>(s, a, 1);
        else launch_cfg<T, EPI, Cfg128, false>(s, a, 1);
        return false;
    }
    // 256x128 tiles: ONE workgroup is resident per CU (144 KiB LDS ring), so workgroup counts are quantised in rounds
    // of 256.  Pick the K split that fills one round (tiles * S <= 256); when a single row tile has between 256 and 512
    // column tiles (gate/up at T <= 256: 296), run the first 256 tiles unsplit and the tail tiles K-split in a second
    // launch instead of a nearly empty second round.
    const int tiles_m = (a.M + 255) / 256, tiles_n = (a.N + 127) / 128;
    const int tiles256 = tiles_m * tiles_n;
    const int stages = (a.K / EPC + Cfg256::CH - 1) / Cfg256::CH;
    const bool can_split = a.ws != nullptr && a.N % 4 == 0 && !(EPI == EPI_SWIGLU && a.N % 64 != 0);
    // (re-measured inside the turn: capping the split count at 4 / 6 instead of filling one round of 256 workgroups costs 0.75 / 0.25 ms of
    //  steady prefill per turn -- the shorter K chains are worth more than the smaller fp32 slabs)
    auto pick = [&](int tiles) {
        int S = 256 / tiles;
        if (S < 1) S = 1;
        if (S > stages / 2) S = stages / 2 < 1 ? 1 : stages / 2;
        if (S > 16) S = 16;
        if (!can_split) S = 1;
        while (S > 1 && (size_t)S * a.M * a.N > a.ws_elems) --S;
        return S;
    };
    if (a.force_split > 0) {
        a.launch_tiles = tiles256;
        if (a.force_split == 1) { a.nsplit = 1; launch_cfg<T, EPI, Cfg256, false>(s, a, 1); }
        else return launch_split<T, EPI>(s, a, a.force_split);
        return false;
    }
    if (tiles_m == 1 && tiles_n > 256 && tiles_n < 512 && can_split && pick(tiles_n - 256) > 1) {
        a.nsplit = 1;
        a.launch_tiles = 256;
        launch_cfg<T, EPI, Cfg256, false>(s, a, 1);
        a.tile_base = 256;
        a.launch_tiles = tiles_n - 256;
        launch_split<T, EPI>(s, a, pick(tiles_n - 256));
        return false;
    }
    a.launch_tiles = tiles256;
    const int S = pick(tiles256);
    if (S <= 1) { a.nsplit = 1; launch_cfg<T, EPI, Cfg256, false>(s, a, 1); return false; }
    return launch_split<T, EPI>(s, a, S);
}

}  // namespace

// lm_head of <= 32 rows (envs decoded together) with the arg-max in the epilogue: one pass of 32x128 tiles over the vocabulary, the weight
// tile staged non-temporally, no C.  Returns the number of column tiles (= partials per row).
template <typename T> int launch_gemm_argmax(hipStream_t s, GemmArgs a) {
    a.nt_w = 1; a.tile_base = 0; a.nsplit = 1; a.vp_on = 0; a.bn_fast = 0;
    a.launch_tiles = (a.N + 127) / 128;
    launch_tile<T, EPI_ARGMAX, CfgSkinny, false, T, true>(s, a, a.launch_tiles);
    return a.launch_tiles;
}
template int launch_gemm_argmax<bf16>(hipStream_t, GemmArgs);
template int launch_gemm_argmax<float>(hipStream_t, GemmArgs);

template <typename T> bool launch_gemm(hipStream_t s, const GemmArgs& a) {
    switch (a.epi) {
        case EPI_NONE: return launch_epi<T, EPI_NONE>(s, a);
        case EPI_GELU_TANH: return launch_epi<T, EPI_GELU_TANH>(s, a);
        case EPI_GELU_ERF: return launch_epi<T, EPI_GELU_ERF>(s, a);
        case EPI_SWIGLU: return launch_epi<T, EPI_SWIGLU>(s, a);
        default: break;
    }
    return false;
}

template <typename T, int EPI> static void gemm_attr() {
    set_max_lds((const void*)gemm_glds_kernel<T, EPI, Cfg128, false>, Cfg128::NBUF * Cfg128::STAGE_BYTES, Cfg128::THREADS);
    set_max_lds((const void*)gemm_glds_kernel<T, EPI, Cfg256, false>, Cfg256::NBUF * Cfg256::STAGE_BYTES, Cfg256::THREADS);
    set_max_lds((const void*)gemm_glds_kernel<T, EPI, Cfg256, true>, Cfg256::NBUF * Cfg256::STAGE_BYTES, Cfg256::THREADS);
    set_max_lds((const void*)gemm_glds_kernel<T, EPI, Cfg256, false, T, true>, Cfg256::NBUF * Cfg256::STAGE_BYTES, Cfg256::THREADS);
    set_max_lds((const void*)gemm_glds_kernel<T, EPI, Cfg256, true, T, true>, Cfg256::NBUF * Cfg256::STAGE_BYTES, Cfg256::THREADS);
    set_max_lds((const void*)gemm_glds_kernel<T, EPI, CfgSkinny, false, T, true>, CfgSkinny::NBUF * CfgSkinny::STAGE_BYTES, CfgSkinny::THREADS);
    set_max_lds((const void*)gemm_glds_kernel<T, EPI, CfgSkinny, true, T, true>, CfgSkinny::NBUF * CfgSkinny::STAGE_BYTES, CfgSkinny::THREADS);
    set_max_lds((const void*)gemm_glds_kernel<T, EPI, CfgSkinny, false>, CfgSkinny::NBUF * CfgSkinny::STAGE_BYTES, CfgSkinny::THREADS);
    set_max_lds((const void*)gemm_glds_kernel<T, EPI, CfgSkinny, true>, CfgSkinny::NBUF * CfgSkinny::STAGE_BYTES, CfgSkinny::THREADS);
    set_max_lds((const void*)gemm_glds_kernel<T, EPI, Cfg64, false>, Cfg64::NBUF * Cfg64::STAGE_BYTES, Cfg64::THREADS);
    set_max_lds((const void*)gemm_glds_kernel<T, EPI, Cfg256N64, false>, Cfg256N64::LDS_BYTES, Cfg256N64::THREADS);
    set_max_lds((const void*)gemm_glds_kernel<T, EPI, Cfg256N64, true>, Cfg256N64::LDS_BYTES, Cfg256N64::THREADS);
    set_max_lds((const void*)gemm_glds_kernel<T, EPI, Cfg256N64, false, T, true>, Cfg256N64::LDS_BYTES, Cfg256N64::THREADS);
    set_max_lds((const void*)gemm_glds_kernel<T, EPI, Cfg256N64, true, T, true>, Cfg256N64::LDS_BYTES, Cfg256N64::THREADS);
    set_max_lds((const void*)gemm_glds_kernel<T, EPI, Cfg128L, false>, Cfg128L::NBUF * Cfg128L::STAGE_BYTES, Cfg128L::THREADS);
    set_max_lds((const void*)gemm_glds_kernel<T, EPI, Cfg128K2, false>, Cfg128K2::LDS_BYTES, Cfg128K2::THREADS);
    if constexpr (EPI == EPI_NONE) set_max_lds((const void*)gemm_glds_kernel<T, EPI_NONE, Cfg128K2, false, T, false, true>, Cfg128K2::LDS_BYTES, Cfg128K2::THREADS);
    set_max_lds((const void*)gemm_glds_kernel<T, EPI, CfgBig, false>, CfgBig::NBUF * CfgBig::STAGE_BYTES, CfgBig::THREADS);
    if constexpr (std::is_same<T, bf16>::value) set_max_lds((const void*)gemm_glds_kernel<T, EPI, Cfg8P, false>, Cfg8P::LDS_BYTES, Cfg8P::THREADS);
    if constexpr (std::is_same<T, bf16>::value) set_max_lds((const void*)gemm_glds_kernel<T, EPI, Cfg8P, true>, Cfg8P::LDS_BYTES, Cfg8P::THREADS);
    if constexpr (std::is_same<T, bf16>::value) set_max_lds((const void*)gemm_glds_kernel<T, EPI, Cfg8P32, false>, Cfg8P32::LDS_BYTES, Cfg8P32::THREADS);
    if constexpr (sizeof(T) == 2 && (EPI == EPI_NONE || EPI == EPI_SWIGLU)) {
        set_max_lds((const void*)gemm_glds_kernel<T, EPI, Cfg128, false, fp8_t>, Cfg128::NBUF * Cfg128::STAGE_BYTES, Cfg128::THREADS);
        set_max_lds((const void*)gemm_glds_kernel<T, EPI, Cfg256, false, fp8_t>, Cfg256::NBUF * Cfg256::STAGE_BYTES, Cfg256::THREADS);
        set_max_lds((const void*)gemm_glds_kernel<T, EPI, Cfg256, true, fp8_t>, Cfg256::NBUF * Cfg256::STAGE_BYTES, Cfg256::THREADS);
        set_max_lds((const void*)gemm_glds_kernel<T, EPI, Cfg256, false, fp8_t, true>, Cfg256::NBUF * Cfg256::STAGE_BYTES, Cfg256::THREADS);
        set_max_lds((const void*)gemm_glds_kernel<T, EPI, Cfg256, true, fp8_t, true>, Cfg256::NBUF * Cfg256::STAGE_BYTES, Cfg256::THREADS);
        set_max_lds((const void*)gemm_glds_kernel<T, EPI, CfgSkinny, false, fp8_t, true>, CfgSkinny::NBUF * CfgSkinny::STAGE_BYTES, CfgSkinny::THREADS);
        set_max_lds((const void*)gemm_glds_kernel<T, EPI, CfgSkinny, true, fp8_t, true>, CfgSkinny::NBUF * CfgSkinny::STAGE_BYTES, CfgSkinny::THREADS);
        set_max_lds((const void*)gemm_glds_kernel<T, EPI, CfgSkinny, false, fp8_t>, CfgSkinny::NBUF * CfgSkinny::STAGE_BYTES, CfgSkinny::THREADS);
        set_max_lds((const void*)gemm_glds_kernel<T, EPI, CfgSkinny, true, fp8_t>, CfgSkinny::NBUF * CfgSkinny::STAGE_BYTES, CfgSkinny::THREADS);
        set_max_lds((const void*)gemm_glds_kernel<T, EPI, Cfg64, false, fp8_t>, Cfg64::NBUF * Cfg64::STAGE_BYTES, Cfg64::THREADS);
        set_max_lds((const void*)gemm_glds_kernel<T, EPI, Cfg128L, false, fp8_t>, Cfg128L::NBUF * Cfg128L::STAGE_BYTES, Cfg128L::THREADS);
        set_max_lds((const void*)gemm_glds_kernel<T, EPI, Cfg128K2, false, fp8_t>, Cfg128K2::LDS_BYTES, Cfg128K2::THREADS);
        set_max_lds((const void*)gemm_glds_kernel<T, EPI, CfgBig, false, fp8_t>, CfgBig::NBUF * CfgBig::STAGE_BYTES, CfgBig::THREADS);
    }
}
template <typename T> static void gemm_argmax_attr() {
    set_max_lds((const void*)gemm_glds_kernel<T, EPI_ARGMAX, CfgSkinny, false, T, true>, CfgSkinny::LDS_BYTES, CfgSkinny::THREADS);
}
void gemm_init_attrs() {
    gemm_argmax_attr<bf16>(); gemm_argmax_attr<float>();
    gemm_attr<bf16, EPI_NONE>(); gemm_attr<bf16, EPI_GELU_TANH>(); gemm_attr<bf16, EPI_GELU_ERF>(); gemm_attr<bf16, EPI_SWIGLU>();
    gemm_attr<float, EPI_NONE>(); gemm_attr<float, EPI_GELU_TANH>(); gemm_attr<float, EPI_GELU_ERF>(); gemm_attr<float, EPI_SWIGLU>();
}
template bool launch_gemm<bf16>(hipStream_t, const GemmArgs&);
template bool launch_gemm<float>(hipStream_t, const GemmArgs&);

}  // namespace svln
