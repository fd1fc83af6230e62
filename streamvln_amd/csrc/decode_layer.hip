// Persistent batch-1 decode layer (round 4): everything of a Qwen2 decoder layer that follows the decode attention -- merge of the
// split-KV partials, o_proj + residual, post_attention_layernorm, gate/up + SwiGLU, down_proj + residual, and the NEXT layer's
// input_layernorm + q|k|v projection (modeling_qwen2.py:269-299) -- as ONE launch of one workgroup per CU, instead of five launches
// (attn_combine, o GEMV, gate/up GEMV, down GEMV, qkv GEMV).  A decode step is HBM-bound on its weight stream (466 MB per layer at true
// size); what the launched form loses is the ramp / tail of every short kernel and the latency chain between them, during which HBM idles.
//
// Structure (third form; the two LDS-ring forms before it are described in DESIGN.md 4.1).  What sets the streaming rate of a CU is
// the number of BYTES IN FLIGHT: chip-wide streaming sees ~5 us of loaded HBM latency, so 6.3 TB/s needs ~128 KB in flight per CU
// (tools/micro/stream_layout.hip: 9 / 18 / 26 GB/s per CU at 40 / 90 / 110 KB in flight, LDS-DMA and register loads alike).  An LDS ring
// next to the 58 KB of activation vectors tops out at 48-64 KB (measured: 10.7 GB/s per CU, 202 us per layer).  Registers do not:
//   * all 8 waves of the workgroup stream: each keeps PF = 16 one-KiB blocks (16 B per lane, non-temporal) of ITS share of the weights in
//     flight in registers (8 x 16 KB = 128 KB per CU), computes block i while block i + 16 loads, and -- because the weights do not depend
//     on the activations -- its prefetch runs across the product boundaries: while the workgroup waits at an edge, the first 16 blocks of
//     the next product are already on their way;
//   * a CU owns outputs [c * N / G, (c + 1) * N / G) of every product.  Rows of K <= 15 KiB go to waves whole (round-robin); longer rows
//     (down_proj) are dealt to the waves block by block, the per-(row, wave) partial sums meet in LDS and are added in a fixed order;
//   * every product's input is an ALL-GATHER of the previous product's output over all CUs: four edges (attention out, x after o_proj,
//     SwiGLU product, x after down_proj) as data-tagged granules (CDNA guide, Guideline 16 R2): 8 bytes {tag, payload} written by ONE sc1
//     store each, swept by all waves with several loads in flight per lane until every tag equals this launch's epoch; no flag, no fence,
//     no grid barrier.  Measured in this harness (tools/micro/seam_bench.hip): 2.9-3.1 us for the 14 KB edges, 5.5-5.7 us for the 74 KB
//     one, against 6.4 us for an XCD-hierarchical grid barrier.
// The attention itself (RoPE, KV append, per-page partials) stays the launch it was (attn_decode_kernel): its partials reach this kernel
// across a kernel boundary, which needs no hand-off.
// Arithmetic per output = the launched kernels' (gemv.hip): fp32 FMA over K, RMSNorm folded as rstd * (W . (g * x)), one rounding to T
// per stored value; the summation ORDER over K differs, so results agree to fp32 rounding, not bit for bit.
// Every spin is bounded: on a timeout a give-up code is stored, every later wait falls through and the grid drains; the host reads
// the word after the turn's synchronisation (svln_generate fails with it).
#include <hip/hip_ext.h>

#include "common.h"
#include "kernels.h"

namespace svln {

namespace {

typedef __attribute__((address_space(1))) unsigned gu32;
typedef __attribute__((address_space(1))) unsigned long long gu64;
typedef __attribute__((ext_vector_type(4))) unsigned int u32x4_t;
#define RLX_AGENT __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT

constexpr int BLK = 1024;                       // bytes one wave instruction moves (64 lanes x 16 B)
constexpr int NW = 8, NTHREADS = 64 * NW;       // waves per workgroup: all of them stream and compute
constexpr int PF = 16;                          // blocks in flight per wave (16 KB: 128 KB per CU; 24 measured slower -- 230 VGPRs -- and 32 no longer unrolls into registers)
constexpr int ROWMODE_MAX_BPR = 15;             // rows of at most this many blocks go to one wave whole
constexpr unsigned SPIN_LIMIT = 50000u;
constexpr size_t LDS_FLOOR = 100 * 1024;        // dynamic LDS requested at least: two workgroups never share a CU

struct Op {            // one product as this CU sees it
    const char* W; size_t ld_bytes;      // weight matrix, row pitch in bytes
    int bpr;                             // 1 KiB blocks per row
    int nrows, pair, first;              // stream rows; pair: rows come as (gate j, up j) of the [gate 32 | up 32] packing; first output index
    int nblk;                            // blocks of this CU's share
    int ch;                              // blocks dealt to a wave at a time: bpr (whole rows) or 1
    int bm;                              // block mode: rows longer than ROWMODE_MAX_BPR blocks, dealt block by block (ch = 1)
};

SVLN_DEV size_t op_row(const Op& o, int j) {                   // matrix row of stream row j
    if (!o.pair) return (size_t)(o.first + j);
    const int out = o.first + (j >> 1);
    return (size_t)(out >> 5) * 64 + (out & 31) + (j & 1) * 32;
}
// slots of wave w in op o (chunks of o.ch blocks dealt round-robin), rounded up to whole groups of PF
SVLN_DEV int op_slots(const Op& o, int w) {
    const int nchunk = (o.nblk + o.ch - 1) / o.ch;
    const int mine = nchunk > w ? (nchunk - w + NW - 1) / NW : 0;
    const int ns = (mine * o.ch + PF - 1) / PF * PF;
    return ns < PF ? PF : ns;            // at least one group, even without work: the last group of an op is where the next op's first PF slots are loaded
}
// Walking the slots of one wave through one op.  Row mode (o.ch == o.bpr): a chunk is one stream row, rows w, w + NW, ...; block mode
// (o.ch == 1, rows longer than ROWMODE_MAX_BPR blocks, stored back to back): blocks w, w + NW, ... of the CU's contiguous share.
// LoadCur yields the source address of every slot (a padding slot re-reads the op's first block); CompCur says which (row, block) a
// slot is and when a row's partial sum is complete.
struct LoadCur {
    const char* ptr; int kc, row, gb;
    SVLN_DEV const char* row_ptr(const Op& o, int r, int lane) const { return o.W + op_row(o, r < o.nrows ? r : 0) * o.ld_bytes + lane * 16; }
    SVLN_DEV void start(const Op& o, int w, int lane) {
        kc = 0; row = w; gb = w;
        ptr = o.bm ? (w < o.nblk ? o.W + (size_t)o.first * o.ld_bytes + (size_t)w * BLK + lane * 16 : o.W + lane * 16) : row_ptr(o, w, lane);
    }
    SVLN_DEV void next(const Op& o, int lane) {
        if (o.bm) {
            gb += NW;
            ptr = gb < o.nblk ? ptr + (size_t)NW * BLK : o.W + (size_t)o.first * o.ld_bytes + lane * 16;
        } else if (++kc == o.ch) {
            kc = 0; row += NW;
            ptr = row_ptr(o, row, lane);
        } else {
            ptr += BLK;
        }
    }
};
struct CompCur {
    int kc, row, pb, gb;
    SVLN_DEV void start(const Op& o, int w) {
        kc = 0; gb = w;
        if (o.bm) { row = w / o.bpr; pb = w - row * o.bpr; } else { row = w; pb = 0; }
    }
    SVLN_DEV bool valid(const Op& o) const { return o.bm ? gb < o.nblk : row < o.nrows; }
    // returns true when the slot just consumed was the last one of its row FOR THIS WAVE (its partial sum is complete)
    SVLN_DEV bool next(const Op& o) {
        if (o.bm) {
            gb += NW; pb += NW;
            if (pb >= o.bpr) { pb -= o.bpr; ++row; return true; }
            return false;
        }
        if (++kc == o.ch) { kc = 0; row += NW; pb = 0; return true; }
        ++pb;
        return false;
    }
};

template <typename T> struct Gran;       // payload of one 8-byte granule: two bf16 values, or one fp32 value
template <> struct Gran<bf16> {
    static constexpr int VPG = 2;
    static SVLN_DEV unsigned pack(const float* v) { return pack_bf16x2(v[0], v[1]); }
    static SVLN_DEV void unpack(unsigned u, float* v) { v[0] = __uint_as_float(u << 16); v[1] = __uint_as_float(u & 0xFFFF0000u); }
};
template <> struct Gran<float> {
    static constexpr int VPG = 1;
    static SVLN_DEV unsigned pack(const float* v) { return __float_as_uint(v[0]); }
    static SVLN_DEV void unpack(unsigned u, float* v) { v[0] = __uint_as_float(u); }
};

// x planes in LDS (gemv.hip's layout): floats of chunk ci, part q (4 floats each) at xs[q * nch * 4 + ci * 4 ...]
template <typename T> SVLN_DEV int xs_index(int k, int nch) {
    constexpr int EPC = Elt<T>::PER_CHUNK;
    const int ci = k / EPC, e = k % EPC;
    return (e >> 2) * nch * 4 + ci * 4 + (e & 3);
}

// The block stream is plain non-temporal loads in STRAIGHT-LINE code: every slot issues exactly one load (an invalid padding slot re-reads
// a valid block and is not computed) and the source pointer is chosen by selects, not branches, so that hipcc's waitcnt insertion can
// count -- loads of one wave return in order, and when slot k is consumed PF - 1 younger loads may stay outstanding.  (Inline-asm loads
// into a register array are NOT safe here: the compiler does not know the destination is still being written and moved / reused those
// registers -- a first version faulted with a memory aperture violation.)
SVLN_DEV u32x4_t load_block(const char* p) { return __builtin_nontemporal_load((const u32x4_t*)p); }

template <typename T>
__global__ __launch_bounds__(NTHREADS) void decode_layer_kernel(DecodeLayerArgs p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    if (p.skip && *p.skip) return;
    constexpr int EPC = Elt<T>::PER_CHUNK, PARTS = EPC / 4, VPG = Gran<T>::VPG;
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int G = gridDim.x, c = blockIdx.x;
    const int H = p.H, I = p.I, qd = p.qd;
    const int kx = H > qd ? H : qd;
    const bool has_next = p.next_qkv_w != nullptr;
    const int perA = qd / G, perH = H / G, perI = I / G, perQ = has_next ? p.qkv_dim / G : 0;
    const int nr_max = max(max(2 * perI, perH), perQ);
    float* xs = (float*)smem;                        // fp32 planes of the current product's input (g * x folded in)
    T* xres = (T*)(xs + kx);                         // residual row
    T* hvec = xres + H;                              // SwiGLU product (down_proj's input)
    float* part = (float*)(hvec + I);                // [row][wave] partial dot products of the current product
    float* outbuf = part + (size_t)nr_max * NW;      // this CU's finished outputs of the current product
    float* ssw = outbuf + 128;                       // per-wave partial sums of squares (RMSNorm)
    const unsigned seq = *p.seq;                     // this launch's epoch: the tags of its four edges are seq * 4 + 1 .. + 4 (never 0; unique
                                                     // over all launches, so granules of earlier launches -- or zeroed memory -- never match)
    gu32* giveup = (gu32*)p.giveup;
    bool dead = false;
    auto stamp = [&](int k) { if (p.dbg && tid == 0) p.dbg[(size_t)c * 16 + k] = wall_clock64(); };

    // (separate objects, never indexed at run time: an array of them would live in scratch memory)
    auto mk = [&](const void* W, int K, int nrows, int pair, int first) {
        Op o;
        o.W = (const char*)W; o.ld_bytes = (size_t)K * sizeof(T); o.bpr = K * (int)sizeof(T) / BLK;
        o.nrows = nrows; o.pair = pair; o.first = first; o.nblk = nrows * o.bpr; o.bm = o.bpr > ROWMODE_MAX_BPR ? 1 : 0; o.ch = o.bm ? 1 : o.bpr;
        return o;
    };
    const Op op_o = mk(p.o_w, qd, perH, 0, c * perH);
    const Op op_gu = mk(p.gu_w, H, 2 * perI, 1, c * perI);
    const Op op_dn = mk(p.down_w, I, perH, 0, c * perH);
    const Op op_qkv = mk(p.next_qkv_w, H, perQ, 0, c * perQ);

    u32x4_t buf[PF];                                 // this wave's blocks in flight
    // the first PF slots of an op into buf (before the op's input exists: the weights do not depend on it)
    auto prefetch = [&](const Op& o) {
        LoadCur lc; lc.start(o, wave, lane);
#pragma unroll
        for (int j = 0; j < PF; ++j) {
            buf[j] = load_block(lc.ptr);
            lc.next(o, lane);
        }
    };
    // One product: this wave walks its slots in groups of PF; slot k is computed from buf[k % PF], which is then refilled with slot k + PF --
    // of this op, or, in the last group, of the NEXT op (its first PF slots: what `prefetch` would load).  xt: the input vector is T values
    // in `hvec` (down_proj) instead of fp32 planes in `xs`.  A row's partial sum (this wave's blocks of it) goes to part[row][wave].
    auto product = [&](const Op& o, const Op& rf, bool xt) {      // rf: the op whose first PF slots refill the last group (the next op; o itself after the last)
        const int ns = op_slots(o, wave), nch = o.bpr * 64;
        CompCur cc; cc.start(o, wave);
        LoadCur lc; lc.start(o, wave, lane);
        for (int j = 0; j < PF; ++j) lc.next(o, lane);             // the load cursor runs PF slots ahead
        const float* xq0 = xs + lane * 4;                          // this lane's 16 bytes of plane 0 / plane 1 in block 0
        const float* xq1 = xs + (size_t)nch * 4 + lane * 4;
        const T* xh = hvec + (size_t)lane * EPC;
        float acc = 0.0f;
        auto slot = [&](int j, const char* refill) {
            if (cc.valid(o)) {
                float wf[EPC], xf[EPC];
                chunk_to_f32<T>(make_uint4(buf[j].x, buf[j].y, buf[j].z, buf[j].w), wf);
                if (xt) {
                    chunk_to_f32<T>(*(const uint4*)(xh + (size_t)cc.pb * 64 * EPC), xf);
                } else {
                    const float4 v0 = *(const float4*)(xq0 + (size_t)cc.pb * 256);
                    xf[0] = v0.x; xf[1] = v0.y; xf[2] = v0.z; xf[3] = v0.w;
                    if (PARTS > 1) {
                        const float4 v1 = *(const float4*)(xq1 + (size_t)cc.pb * 256);
                        xf[4 % EPC] = v1.x; xf[5 % EPC] = v1.y; xf[6 % EPC] = v1.z; xf[7 % EPC] = v1.w;
                    }
                }
#pragma unroll
                for (int e = 0; e < EPC; ++e) acc = fmaf(wf[e], xf[e], acc);
                const int r = cc.row;
                if (cc.next(o)) {                                   // this wave's last block of row r
                    const float t = wave_sum(acc);
                    if (lane == 0) part[r * NW + wave] = t;
                    acc = 0.0f;
                }
            } else {
                cc.next(o);
            }
            buf[j] = load_block(refill);
        };
        int k0 = 0;
        for (; k0 + PF < ns; k0 += PF) {                            // groups refilled from this op
#pragma unroll
            for (int j = 0; j < PF; ++j) {
                const char* src = lc.ptr;
                lc.next(o, lane);
                slot(j, src);
            }
        }
        LoadCur nc; nc.start(rf, wave, lane);                       // the last group: refilled with the next op's first PF slots
#pragma unroll
        for (int j = 0; j < PF; ++j) {
            const char* src = nc.ptr;
            nc.next(rf, lane);
            slot(j, src);
        }
        // (every row's partial has been written: in block mode a wave's last block of a row is always followed by a slot in a later row --
        //  valid or not -- because the share is a whole number of rows)
    };
    // sum of the NW partials of stream row r, in wave order
    auto row_sum = [&](int r) {
        float t = 0.0f;
#pragma unroll
        for (int q = 0; q < NW; ++q) t += part[r * NW + q];
        return t;
    };
    auto zero_part = [&](int rows) { for (int k = tid; k < rows * NW; k += NTHREADS) part[k] = 0.0f; };
    // publish this CU's `per` values of outbuf as granules of edge e (wave 0)
    auto publish = [&](int e, int per) {
        if (wave != 0) return;
        const int gpc = per / VPG;
        const unsigned tag = seq * 4u + (unsigned)e + 1u;
        for (int gi = lane; gi < gpc; gi += 64) {
            float v[2] = {outbuf[gi * VPG], VPG > 1 ? outbuf[gi * VPG + 1] : 0.0f};
            __hip_atomic_store((gu64*)p.gran[e] + (size_t)c * gpc + gi, ((unsigned long long)tag << 32) | Gran<T>::pack(v), RLX_AGENT);
        }
    };
    // gather edge e (n values from all CUs): every wave sweeps an eighth of the granules, 8 loads in flight per lane, re-reading a batch
    // until every tag matches; sink(k, value) stores value k of the vector
    auto gather = [&](int e, int n, auto&& sink) {
        const unsigned tag = seq * 4u + (unsigned)e + 1u;
        const int ng = n / VPG, share = (ng + NW - 1) / NW, first = wave * share, count = min(share, ng - first);
        const unsigned long long* g = p.gran[e];
        for (int b0 = 0; b0 < count; b0 += 64 * 8) {
            unsigned spins = 0;
            for (;;) {
                unsigned long long v[8];
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    int k = b0 + j * 64 + lane;
                    k = first + (k < count ? k : count - 1);
                    const unsigned long long* q = g + k;
                    asm volatile("global_load_dwordx2 %0, %1, off sc1" : "=v"(v[j]) : "v"(q) : "memory");
                }
                asm volatile("s_waitcnt vmcnt(0)" : "+v"(v[0]), "+v"(v[1]), "+v"(v[2]), "+v"(v[3]), "+v"(v[4]), "+v"(v[5]), "+v"(v[6]), "+v"(v[7]));
                bool ok = true;
#pragma unroll
                for (int j = 0; j < 8; ++j) ok &= (unsigned)(v[j] >> 32) == tag;
                if (__all(ok) || dead) {
#pragma unroll
                    for (int j = 0; j < 8; ++j) {
                        const int k = b0 + j * 64 + lane;
                        if (k < count) {
                            float f[2];
                            Gran<T>::unpack((unsigned)v[j], f);
#pragma unroll
                            for (int q = 0; q < VPG; ++q) sink((first + k) * VPG + q, f[q]);
                        }
                    }
                    break;
                }
                __builtin_amdgcn_s_sleep(1);
                if (++spins > SPIN_LIMIT || __hip_atomic_load(giveup, RLX_AGENT)) { dead = true; __hip_atomic_store(giveup, 0x400u + (unsigned)e, RLX_AGENT); }
            }
        }
    };
    // gather an H-vector and fold the following RMSNorm: xs = g * x (fp32 planes), optional xres = x; returns rsqrt(mean x^2 + eps)
    auto gather_norm = [&](int e, const T* gw, bool keep_res, int next_rows) {
        float ss = 0.0f;
        const int nch = H / EPC;
        gather(e, H, [&](int k, float v) {
            ss = fmaf(v, v, ss);
            if (keep_res) xres[k] = from_f32<T>(v);
            xs[xs_index<T>(k, nch)] = v * to_f32(gw[k]);
        });
        ss = wave_sum(ss);
        if (lane == 0) ssw[wave] = ss;
        zero_part(next_rows);
        __syncthreads();
        float tot = 0.0f;
#pragma unroll
        for (int q = 0; q < NW; ++q) tot += ssw[q];
        return rsqrtf(tot / (float)H + p.eps);
    };

    stamp(9);
    // (0) residual row of this layer's input; this CU's slice of the merged attention output; o_proj's first blocks on their way.  Loads of a
    // wave return in order: the small loads go FIRST (behind 16-32 KB of weight blocks the merge's two dependent load rounds took ~12 us),
    // so the merging wave starts its share of the weight stream only after the merge.
    for (int ci = tid; ci < H / EPC; ci += NTHREADS) *(uint4*)(xres + (size_t)ci * EPC) = *(const uint4*)((const T*)p.x + (size_t)ci * EPC);
    if (wave != 1 % NW) prefetch(op_o);
    if (wave == 1 % NW) {
        // merge of the split-KV partials for this CU's perA output elements (attn_combine_kernel's arithmetic; the sum over the splits is a
        // wave reduction here): lane z owns split z.  The elements span at most two q heads (perA <= 64 < 128).  (Wave 1: wave 0 publishes.)
        const int kv_len = *p.dyn_kv_len, tiles = (kv_len + 63) >> 6;
        const int nsplit = min(min(p.nsplit, 64), (tiles + p.tiles_per_split - 1) / p.tiles_per_split);
        const size_t split_stride = (size_t)p.n_kv * 32 * (128 + ATTN_PART_PAD);
        const int n0 = c * perA, hqA = n0 >> 7, hqB = (n0 + perA - 1) >> 7;
        auto head_base = [&](int hq) { const int kh = hq / p.Gq, rho = hq - kh * p.Gq; return p.part + ((size_t)kh * 32 + rho) * (128 + ATTN_PART_PAD) + (size_t)lane * split_stride; };
        const bool live = lane < nsplit;
        const float* bA = head_base(hqA);
        const float* bB = head_base(hqB);
        const float mA = live ? bA[128] : -INFINITY, lA = live ? bA[129] : 0.0f;
        const float mB = live ? bB[128] : -INFINITY, lB = live ? bB[129] : 0.0f;
        const float msA = wave_max(mA), msB = wave_max(mB);
        const float wA = mA == -INFINITY ? 0.0f : __builtin_amdgcn_exp2f(mA - msA), wB = mB == -INFINITY ? 0.0f : __builtin_amdgcn_exp2f(mB - msB);
        const float sA = wave_sum(wA * lA), sB = wave_sum(wB * lB);
        const float invA = sA > 0.0f ? 1.0f / sA : 0.0f, invB = sB > 0.0f ? 1.0f / sB : 0.0f;
        for (int j0 = 0; j0 < perA; j0 += 16) {
            float t[16];
#pragma unroll
            for (int j = 0; j < 16; ++j) {
                const int n = n0 + min(j0 + j, perA - 1), hq = n >> 7, d = n & 127;
                t[j] = live ? (hq == hqA ? wA * bA[d] : wB * bB[d]) : 0.0f;
            }
#pragma unroll
            for (int j = 0; j < 16; ++j) {
                const float sum = wave_sum(t[j]);
                const int hq = (n0 + min(j0 + j, perA - 1)) >> 7;
                if (lane == 0 && j0 + j < perA) outbuf[j0 + j] = to_f32(from_f32<T>(sum * (hq == hqA ? invA : invB)));
            }
        }
        prefetch(op_o);
    }
    __syncthreads();
    stamp(0);
    // (1) edge 0: merged attention output (qd values) -> xs, no norm
    publish(0, perA);
    {
        const int nch = qd / EPC;
        gather(0, qd, [&](int k, float v) { xs[xs_index<T>(k, nch)] = v; });
    }
    zero_part(op_o.nrows);
    __syncthreads();
    stamp(1);
    // (2) o_proj + residual -> x1
    product(op_o, op_gu, false);
    __syncthreads();
    if (tid < perH) outbuf[tid] = to_f32(from_f32<T>(row_sum(tid) + to_f32(xres[c * perH + tid])));
    __syncthreads();
    stamp(2);
    // (3) edge 1: x1 -> xres, xs = post_attention_layernorm weight * x1, rstd
    publish(1, perH);
    const float rstd1 = gather_norm(1, (const T*)p.post_norm, true, op_gu.nrows);
    stamp(3);
    // (4) gate / up + SwiGLU -> this CU's slice of the SwiGLU product
    product(op_gu, op_dn, false);
    __syncthreads();
    if (tid < perI) outbuf[tid] = to_f32(from_f32<T>(silu_f(row_sum(2 * tid) * rstd1) * (row_sum(2 * tid + 1) * rstd1)));
    __syncthreads();
    stamp(4);
    // (5) edge 2: SwiGLU product (I values) -> hvec
    publish(2, perI);
    gather(2, I, [&](int k, float v) { hvec[k] = from_f32<T>(v); });
    zero_part(op_dn.nrows);
    __syncthreads();
    stamp(5);
    // (6) down_proj + residual -> x2 (also to global x: the next layer's launch / the head read it there)
    if (has_next) product(op_dn, op_qkv, true); else product(op_dn, op_dn, true);
    __syncthreads();
    if (tid < perH) {
        const int n = c * perH + tid;
        const T vt = from_f32<T>(row_sum(tid) + to_f32(xres[n]));
        outbuf[tid] = to_f32(vt);
        ((T*)p.x)[n] = vt;
    }
    __syncthreads();
    stamp(6);
    if (has_next) {
        // (7) edge 3: x2 -> xs = next input_layernorm weight * x2, rstd
        publish(3, perH);
        const float rstd2 = gather_norm(3, (const T*)p.next_norm, false, op_qkv.nrows);
        stamp(7);
        // (8) the next layer's q | k | v rows (+ bias), un-roped: its attn_decode launch ropes q / k and appends k / v
        product(op_qkv, op_qkv, false);
        __syncthreads();
        if (tid < perQ) {
            const int n = c * perQ + tid;
            float v = row_sum(tid) * rstd2;
            if (p.next_qkv_b) v += to_f32(((const T*)p.next_qkv_b)[n]);
            ((T*)p.qkv_out)[n] = from_f32<T>(v);
        }
        stamp(8);
    }
    if (c == 0 && tid == 0) *p.seq = seq + 1u;       // (every workgroup read seq before it could publish; the last edge is behind everyone)
}

template <typename T> size_t layer_lds_bytes(const DecodeLayerArgs& a, int G) {
    const int kx = a.H > a.qd ? a.H : a.qd;
    int nr = 2 * (a.I / G);
    if (a.H / G > nr) nr = a.H / G;
    if (a.qkv_dim / G > nr) nr = a.qkv_dim / G;
    const size_t need = (size_t)kx * 4 + (size_t)a.H * sizeof(T) + (size_t)a.I * sizeof(T) + (size_t)nr * NW * 4 + 128 * 4 + NW * 4 + 64;
    return need > LDS_FLOOR ? need : LDS_FLOOR;
}

}  // namespace

template <typename T> bool decode_layer_supported(const DecodeLayerArgs& a, int cus) {
    constexpr int VPG = Gran<T>::VPG;
    const int G = cus;
    auto rows_ok = [&](int K) { return (K * (int)sizeof(T)) % BLK == 0; };
    if (G < 1 || a.H % G || a.I % G || a.qd % G || a.qkv_dim % G) return false;
    if ((a.H / G) % VPG || (a.I / G) % VPG || (a.qd / G) % VPG) return false;
    if (a.H / G > 64 || a.I / G > 128 || a.qd / G > 64 || a.qkv_dim / G > 128) return false;     // a CU's outputs fit outbuf; the merge spans <= 2 heads
    if (!rows_ok(a.H) || !rows_ok(a.I) || !rows_ok(a.qd)) return false;
    if (a.H * (int)sizeof(T) / BLK > ROWMODE_MAX_BPR) return false;      // [gate | up] row pairs are dealt to the waves as whole rows
    if (a.H % Elt<T>::PER_CHUNK || a.I % Elt<T>::PER_CHUNK) return false;
    return layer_lds_bytes<T>(a, G) <= (size_t)160 * 1024;
}
template <typename T> void launch_decode_layer(hipStream_t s, const DecodeLayerArgs& a, int cus, hipEvent_t start, hipEvent_t stop) {
    if (start || stop) hipExtLaunchKernelGGL((decode_layer_kernel<T>), dim3(cus), dim3(NTHREADS), layer_lds_bytes<T>(a, cus), s, start, stop, 0, a);
    else hipLaunchKernelGGL((decode_layer_kernel<T>), dim3(cus), dim3(NTHREADS), layer_lds_bytes<T>(a, cus), s, a);
}
void decode_layer_init_attrs() {
    set_max_lds((const void*)decode_layer_kernel<bf16>, 160 * 1024, NTHREADS);
    set_max_lds((const void*)decode_layer_kernel<float>, 160 * 1024, NTHREADS);
}
template bool decode_layer_supported<bf16>(const DecodeLayerArgs&, int);
template bool decode_layer_supported<float>(const DecodeLayerArgs&, int);
template void launch_decode_layer<bf16>(hipStream_t, const DecodeLayerArgs&, int, hipEvent_t, hipEvent_t);
template void launch_decode_layer<float>(hipStream_t, const DecodeLayerArgs&, int, hipEvent_t, hipEvent_t);

}  // namespace svln
