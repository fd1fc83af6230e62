// Persistent batch-1 decode layer (round 4): everything of a Qwen2 decoder layer that follows the decode attention -- merge of the
// split-KV partials, o_proj + residual, post_attention_layernorm, gate/up + SwiGLU, down_proj + residual, and the NEXT layer's
// input_layernorm + q|k|v projection (modeling_qwen2.py:269-299) -- as ONE launch of one workgroup per CU, instead of five launches
// (attn_combine, o GEMV, gate/up GEMV, down GEMV, qkv GEMV).  A decode step is HBM-bound on its weight stream (466 MB per layer at true
// size); what the launched form loses is the ramp / tail of every short kernel and the latency chain between them, during which HBM idles.
// Here the stream never stops:
//   * waves 0-3 of every workgroup are LOADERS: together they move this CU's share of the four weight matrices, in consumption order,
//     HBM -> LDS by LDS-DMA (global_load_lds, non-temporal) into a ring of 16 KiB slots (wave l issues every fourth 1 KiB block), up to D
//     slots in flight behind each wave's counted s_waitcnt vmcnt, and run ahead of every dependency as far as the ring has room.  Four,
//     because one wave streams only ~9 GB/s whatever its depth and four reach the chip's 6.6 TB/s (tools/micro/stream_layout.hip);
//   * waves 4-7 are CONSUMERS: fp32 FMA dot products of the weight bytes in the ring against the op's input vector in LDS; a weight row
//     (or a [gate | up] row pair) belongs to one wave, which carries its accumulator across slots;
//   * a CU owns outputs [c * N / G, (c + 1) * N / G) of every product, so every product's input is an ALL-GATHER of the previous product's
//     output over all CUs.  The four edges (attention out, x after o_proj, SwiGLU product, x after down_proj) are data-tagged granules
//     (CDNA guide, Guideline 16 R2): 8 bytes {tag, payload} written by ONE sc1 store each, swept by the consumer waves with 16
//     loads in flight per lane until every tag equals this launch's epoch; no flag, no fence.  Measured in this harness
//     (tools/micro/seam_bench.hip): 2.9-3.1 us for the 14 KB edges, 5.5-5.7 us for the 74 KB one, against 6.4 us for an XCD-hierarchical
//     grid barrier; the loader's run-ahead hides most of it.
// The attention itself (RoPE, KV append, per-page partials) stays the launch it was (attn_decode_kernel): its partials reach this kernel
// across a kernel boundary, which needs no hand-off.
// Arithmetic per output = the launched kernels' (gemv.hip): fp32 FMA over K, RMSNorm folded as rstd * (W . (g * x)), one rounding to T
// per stored value; the summation ORDER over K differs (lane-strided blocks here), so results agree to fp32 rounding, not bit for bit.
// Every spin is bounded: on a timeout a give-up code is stored, every later wait falls through and the grid drains; the host reads
// the word after the turn's synchronisation (svln_generate fails with it).
#include <hip/hip_ext.h>

#include "common.h"
#include "kernels.h"

namespace svln {

namespace {

typedef __attribute__((address_space(3))) void* lds_ptr_t;
typedef const __attribute__((address_space(1))) void* gbl_ptr_t;
typedef __attribute__((address_space(1))) unsigned gu32;
typedef __attribute__((address_space(1))) unsigned long long gu64;
#define RLX_AGENT __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT

constexpr int SLOT = 16384, BLK = 1024, BPS = SLOT / BLK;      // ring slot, bytes per LDS-DMA wave instruction, blocks per slot
// Waves of a workgroup: NLOAD loaders + NCONS consumers.  ONE wave sustains only ~9 GB/s of LDS-DMA (or register) streaming whatever its
// depth, two 18, four 26 GB/s per CU = 6.6 TB/s chip-wide (tools/micro/stream_layout.hip, profiles/r04_stream_layout_bench.txt): the first
// version of this kernel, with one loader wave, streamed 10.7 GB/s per CU and took 202 us per layer.
constexpr int NLOAD = 4, NCONS = 4, NTHREADS = 64 * (NLOAD + NCONS);
constexpr unsigned SPIN_LIMIT = 400000u;

// LDS sync words (dynamic LDS tail), all monotonic within a launch
enum { W_READY = 0 /* +loader */, W_FREED = 4 /* +w */, W_UNITS = 8, W_CBAR = 9 /* +w */, W_SS = 13 /* +w (float bits) */, W_N = 20 };

struct Op {            // one product as this CU sees it
    const char* W; size_t ld_bytes;      // weight matrix, row pitch in bytes
    int rb, bpr;                         // row bytes streamed (K * sizeof(T)), 1 KiB blocks per row
    int nrows, pair, first;              // stream rows; pair: rows come as (gate j, up j) of the [gate 32 | up 32] packing; first output index
    int nblk, slot0, nslots;
};

SVLN_DEV size_t op_row(const Op& o, int j) {                   // matrix row of stream row j
    if (!o.pair) return (size_t)(o.first + j);
    const int out = o.first + (j >> 1);
    return (size_t)(out >> 5) * 64 + (out & 31) + (j & 1) * 32;
}

// LDS words through inline asm: hipcc orders every LDS access it can see behind ALL outstanding LDS-DMA of the wave (s_waitcnt vmcnt(0)),
// which would drain the loader's ring at every poll (gemm.hip has the same note)
SVLN_DEV unsigned lds_read_u32(unsigned addr) {
    unsigned v;
    asm volatile("ds_read_b32 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(v) : "v"(addr) : "memory");
    return v;
}
SVLN_DEV void lds_write_u32(unsigned addr, unsigned v) { asm volatile("ds_write_b32 %0, %1\n\ts_waitcnt lgkmcnt(0)" ::"v"(addr), "v"(v) : "memory"); }
SVLN_DEV void lds_add_u32(unsigned addr, unsigned v) { asm volatile("ds_add_u32 %0, %1\n\ts_waitcnt lgkmcnt(0)" ::"v"(addr), "v"(v) : "memory"); }

template <int N> SVLN_DEV void wait_vm() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }
SVLN_DEV void wait_vm_le(int n) {        // at most n (<= 48) of this wave's vector-memory operations still outstanding
    switch (n) {
#define SVLN_CASE(k) case k: wait_vm<k>(); break;
        SVLN_CASE(0) SVLN_CASE(1) SVLN_CASE(2) SVLN_CASE(3) SVLN_CASE(4) SVLN_CASE(5) SVLN_CASE(6) SVLN_CASE(7) SVLN_CASE(8) SVLN_CASE(9)
        SVLN_CASE(10) SVLN_CASE(11) SVLN_CASE(12) SVLN_CASE(13) SVLN_CASE(14) SVLN_CASE(15) SVLN_CASE(16) SVLN_CASE(17) SVLN_CASE(18) SVLN_CASE(19)
        SVLN_CASE(20) SVLN_CASE(21) SVLN_CASE(22) SVLN_CASE(23) SVLN_CASE(24) SVLN_CASE(25) SVLN_CASE(26) SVLN_CASE(27) SVLN_CASE(28) SVLN_CASE(29)
        SVLN_CASE(30) SVLN_CASE(31) SVLN_CASE(32) SVLN_CASE(33) SVLN_CASE(34) SVLN_CASE(35) SVLN_CASE(36) SVLN_CASE(37) SVLN_CASE(38) SVLN_CASE(39)
        SVLN_CASE(40) SVLN_CASE(41) SVLN_CASE(42) SVLN_CASE(43) SVLN_CASE(44) SVLN_CASE(45) SVLN_CASE(46) SVLN_CASE(47) SVLN_CASE(48)
#undef SVLN_CASE
        default: wait_vm<0>(); break;
    }
}

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

template <typename T, int NS>
__global__ __launch_bounds__(NTHREADS) void decode_layer_kernel(DecodeLayerArgs p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    if (p.skip && *p.skip) return;
    constexpr int EPC = Elt<T>::PER_CHUNK, PARTS = EPC / 4, VPG = Gran<T>::VPG;
    constexpr int D = NS >= 5 ? 3 : (NS >= 4 ? 2 : 1);           // slots in flight behind the loader (the rest of the ring: published / being read)
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int G = gridDim.x, c = blockIdx.x;
    const int H = p.H, I = p.I, qd = p.qd;
    const int kx = H > qd ? H : qd;
    char* ring = smem;
    float* xs = (float*)(ring + (size_t)NS * SLOT);
    T* xres = (T*)(xs + kx);
    T* hvec = xres + H;
    float* outbuf = (float*)(hvec + I);
    unsigned* syncw = (unsigned*)(outbuf + 128);
    const unsigned sync0 = (unsigned)(size_t)(__attribute__((address_space(3))) char*)(char*)syncw;
    if (tid < W_N) syncw[tid] = 0u;
    // this launch's epoch: tags of its four edges are seq * 4 + 1 .. + 4 (never 0; unique over all launches, so granules left by earlier
    // launches -- or zeroed memory -- never match)
    const unsigned seq = *p.seq;
    __syncthreads();

    // optional phase stamps (100 MHz wall clock) of this workgroup: [0..9] consumer wave 0, [10..14] the loader (tools / DESIGN.md only)
    auto stamp = [&](int k) { if (p.dbg && lane == 0) p.dbg[(size_t)c * 16 + k] = wall_clock64(); };
    const bool has_next = p.next_qkv_w != nullptr;
    const int perA = qd / G, perH = H / G, perI = I / G, perQ = has_next ? p.qkv_dim / G : 0;
    // (four separate objects, never indexed at run time: an array of them lands in scratch memory, and a scratch load in the loader's
    //  loop makes hipcc wait for vmcnt(0) -- the whole DMA ring -- before every slot)
    auto mk = [&](const void* W, int K, int nrows, int pair, int first, int slot0) {
        Op o;
        o.W = (const char*)W; o.ld_bytes = (size_t)K * sizeof(T); o.rb = K * (int)sizeof(T); o.bpr = o.rb / BLK;
        o.nrows = nrows; o.pair = pair; o.first = first; o.nblk = nrows * o.bpr; o.slot0 = slot0; o.nslots = (o.nblk + BPS - 1) / BPS;
        return o;
    };
    const Op op_o = mk(p.o_w, qd, perH, 0, c * perH, 0);
    const Op op_gu = mk(p.gu_w, H, 2 * perI, 1, c * perI, op_o.nslots);
    const Op op_dn = mk(p.down_w, I, perH, 0, c * perH, op_gu.slot0 + op_gu.nslots);
    const Op op_qkv = mk(p.next_qkv_w, H, perQ, 0, c * perQ, op_dn.slot0 + op_dn.nslots);
    const int total_slots = has_next ? op_qkv.slot0 + op_qkv.nslots : op_dn.slot0 + op_dn.nslots;
    gu32* giveup = (gu32*)p.giveup;

    if (wave < NLOAD) {
        // ------------------------------------------------------------------------------------------------ loaders (wave l issues blocks l, l + NLOAD, ... of every slot)
        int cnt[D + 1];                       // loads issued per slot, slots sg - D .. sg (ring of the last D + 1)
#pragma unroll
        for (int k = 0; k <= D; ++k) cnt[k] = 0;
        int sg = 0, outstanding = 0;
        bool dead = false;
        // (one call per product with a constant index: the Op fields stay in scalar registers instead of a runtime-indexed private array)
        // issue loop: rows -> 1 KiB blocks, the per-lane source pointer advances by 1 KiB per block (one 64-bit add), the slot position is
        // scalar; per SLOT: room check before its first block, counted wait + publish after its last
        int b = 0, mine = 0;                  // blocks of the current slot walked so far / issued by THIS wave
        int published = 0;                    // slots announced to the consumers (W_READY)
        char* dst = ring;
        auto slot_begin = [&](int oi) {
            if (sg >= NS && !dead) {          // slot sg reuses the buffer of slot sg - NS, which every consumer must have released
                unsigned spins = 0;
                for (;;) {
                    unsigned f0, f1, f2, f3;  // (four reads in flight, one wait)
                    asm volatile("ds_read_b32 %0, %4\n\tds_read_b32 %1, %4 offset:4\n\tds_read_b32 %2, %4 offset:8\n\tds_read_b32 %3, %4 offset:12\n\ts_waitcnt lgkmcnt(0)"
                                 : "=&v"(f0), "=&v"(f1), "=&v"(f2), "=&v"(f3) : "v"(sync0 + 4 * W_FREED) : "memory");
                    const unsigned fm = min(min(f0, f1), min(f2, f3));
                    if ((int)fm + NS > sg) break;
                    if (spins == 0) {
                        // the ring is full: the consumers are behind (an edge).  Nothing can be issued, so waiting for everything in
                        // flight costs nothing -- and publishes the D slots that would otherwise stay unannounced until the next issue
                        wait_vm<0>();
                        published = sg;
                        lds_write_u32(sync0 + 4 * (W_READY + wave), (unsigned)sg);
#pragma unroll
                        for (int k = 0; k <= D; ++k) cnt[k] = 0;
                    }
                    __builtin_amdgcn_s_sleep(1);
                    if (++spins > SPIN_LIMIT) { dead = true; __hip_atomic_store(giveup, 0x100u + (unsigned)oi, RLX_AGENT); break; }
                }
            }
            dst = ring + (size_t)(sg % NS) * SLOT;
        };
        auto slot_end = [&](int nb) {
            // slot sg - D has landed once at most the loads of slots sg - D + 1 .. sg are outstanding
#pragma unroll
            for (int k = 0; k < D; ++k) cnt[k] = cnt[k + 1];
            cnt[D] = nb;
            outstanding = 0;
#pragma unroll
            for (int k = 1; k <= D; ++k) outstanding += cnt[k];
            if (sg >= D) {
                wait_vm_le(outstanding);
                if (sg - D + 1 > published) {         // (never move the published count backwards: a blocked loader may have published further)
                    published = sg - D + 1;
                    lds_write_u32(sync0 + 4 * (W_READY + wave), (unsigned)published);
                }
            }
            ++sg;
            b = 0;
        };
        auto stream_op = [&](const Op& o, int oi) {
            for (int j = 0; j < o.nrows; ++j) {
                const char* src = o.W + op_row(o, j) * o.ld_bytes + lane * 16;
                for (int pb = 0; pb < o.bpr; ++pb) {
                    if (b == 0) { slot_begin(oi); mine = 0; }
                    if ((b & (NLOAD - 1)) == wave) {
                        __builtin_amdgcn_global_load_lds((gbl_ptr_t)src, (lds_ptr_t)(dst + b * BLK), 16, 0, 2);      // aux 2 = non-temporal: read once, by this CU
                        ++mine;
                    }
                    src += BLK;
                    if (++b == BPS) slot_end(mine);
                }
            }
            if (b) slot_end(mine);            // the op's last, partial slot (the next op starts a new slot)
        };
        if (wave == 0) stamp(10);
        stream_op(op_o, 0);
        if (wave == 0) stamp(11);
        stream_op(op_gu, 1);
        if (wave == 0) stamp(12);
        stream_op(op_dn, 2);
        if (wave == 0) stamp(13);
        if (has_next) stream_op(op_qkv, 3);
        wait_vm<0>();
        if (wave == 0) stamp(14);
        lds_write_u32(sync0 + 4 * (W_READY + wave), (unsigned)total_slots);
        return;
    }

    // ---------------------------------------------------------------------------------------------------- consumers
    const int w = wave - NLOAD;
    const int ctid = tid - 64 * NLOAD;                         // 0 .. 64 * NCONS - 1
    bool dead = false;
    auto fail = [&](unsigned code) { dead = true; __hip_atomic_store(giveup, code, RLX_AGENT); };
    volatile unsigned* vs = syncw;
    // consumer-only barrier (the loader never joins an s_barrier after the start): wave w stores its epoch, all wait for the three
    unsigned cb_epoch = 0;
    auto cbarrier = [&]() {
        ++cb_epoch;
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
        if (lane == 0) vs[W_CBAR + w] = cb_epoch;
        unsigned spins = 0;
        while (!dead && (vs[W_CBAR] < cb_epoch || vs[W_CBAR + 1] < cb_epoch || vs[W_CBAR + 2] < cb_epoch || vs[W_CBAR + 3] < cb_epoch)) {
            __builtin_amdgcn_s_sleep(1);
            if (++spins > SPIN_LIMIT) fail(0x200u + cb_epoch);
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
    };

    // (0) residual stream of this layer's input, and this CU's slice of the merged attention output
    for (int ci = ctid; ci < H / EPC; ci += 64 * NCONS) *(uint4*)(xres + (size_t)ci * EPC) = *(const uint4*)((const T*)p.x + (size_t)ci * EPC);
    if (w == 0) {
        // merge of the split-KV partials for this CU's perA output elements (attn_combine_kernel's arithmetic; the sum over the splits is a
        // wave reduction here): lane z owns split z.  The elements span at most two q heads (perA <= 64 < 128): split weights exp2(m_z - max m)
        // and the normaliser once per head, then the elements in batches of 16 independent loads per lane.
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
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");

    unsigned units_target = 0;                 // units finalised by this CU's consumers so far (W_UNITS counts them)
    // publish this CU's `per` values of outbuf as granules of edge e (consumer wave 0, after all `per` units are in outbuf)
    auto publish = [&](int e, int per, bool wait_units) {
        if (w != 0) return;
        if (wait_units) {
            unsigned spins = 0;
            while (!dead && vs[W_UNITS] < units_target) {
                __builtin_amdgcn_s_sleep(1);
                if (++spins > SPIN_LIMIT) fail(0x300u + (unsigned)e);
            }
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
        }
        const int gpc = per / VPG;
        const unsigned tag = seq * 4u + (unsigned)e + 1u;
        for (int gi = lane; gi < gpc; gi += 64) {
            float v[2] = {outbuf[gi * VPG], VPG > 1 ? outbuf[gi * VPG + 1] : 0.0f};
            __hip_atomic_store((gu64*)p.gran[e] + (size_t)c * gpc + gi, ((unsigned long long)tag << 32) | Gran<T>::pack(v), RLX_AGENT);
        }
    };
    // gather edge e (n values from all CUs): the three consumer waves sweep a third of the granules each, 16 loads in flight per lane,
    // re-reading a batch until every tag matches; sink(k, value) stores value k of the vector
    auto gather = [&](int e, int n, auto&& sink) {
        const unsigned tag = seq * 4u + (unsigned)e + 1u;
        const int ng = n / VPG, third = (ng + NCONS - 1) / NCONS, first = w * third, count = min(third, ng - first);
        const unsigned long long* g = p.gran[e];
        for (int b0 = 0; b0 < count; b0 += 64 * 16) {
            unsigned spins = 0;
            for (;;) {
                unsigned long long v[16];
#pragma unroll
                for (int j = 0; j < 16; ++j) {
                    int k = b0 + j * 64 + lane;
                    k = first + (k < count ? k : count - 1);
                    const unsigned long long* q = g + k;
                    asm volatile("global_load_dwordx2 %0, %1, off sc1" : "=v"(v[j]) : "v"(q) : "memory");
                }
                asm volatile("s_waitcnt vmcnt(0)" : "+v"(v[0]), "+v"(v[1]), "+v"(v[2]), "+v"(v[3]), "+v"(v[4]), "+v"(v[5]), "+v"(v[6]), "+v"(v[7]),
                             "+v"(v[8]), "+v"(v[9]), "+v"(v[10]), "+v"(v[11]), "+v"(v[12]), "+v"(v[13]), "+v"(v[14]), "+v"(v[15]));
                bool ok = true;
#pragma unroll
                for (int j = 0; j < 16; ++j) ok &= (unsigned)(v[j] >> 32) == tag;
                if (__all(ok) || dead) {
#pragma unroll
                    for (int j = 0; j < 16; ++j) {
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
                if (++spins > SPIN_LIMIT / 8) fail(0x400u + (unsigned)e);
            }
        }
    };
    // gather an H-vector and fold the following RMSNorm: xs = g * x (fp32 planes), optional xres = x; returns rsqrt(mean x^2 + eps)
    auto gather_norm = [&](int e, const T* gw, bool keep_res) {
        float ss = 0.0f;
        const int nch = H / EPC;
        gather(e, H, [&](int k, float v) {
            ss = fmaf(v, v, ss);
            if (keep_res) xres[k] = from_f32<T>(v);
            xs[xs_index<T>(k, nch)] = v * to_f32(gw[k]);
        });
        ss = wave_sum(ss);
        if (lane == 0) vs[W_SS + w] = __float_as_uint(ss);
        cbarrier();
        const float tot = __uint_as_float(vs[W_SS]) + __uint_as_float(vs[W_SS + 1]) + __uint_as_float(vs[W_SS + 2]) + __uint_as_float(vs[W_SS + 3]);
        cbarrier();                           // (W_SS is rewritten by the next norm: everyone has read it)
        return rsqrtf(tot / (float)H + p.eps);
    };

    // one product: walk the op's slots in stream order; this wave takes the blocks of its units (unit u -> wave u % 3) and carries the
    // unit's accumulators across slots.  XT: the input vector is T values in `hvec` (down_proj) instead of fp32 planes in `xs`.
    // one product: this wave takes the rows of its units (unit u -> wave u % 3), walking each row's 1 KiB blocks slot by slot.  Entering
    // slot sg releases every earlier slot (freed = sg) and waits until the loader has published it.
    int cur_slot = -1;
    auto enter_slot = [&](int sg, int oi) {
        if (sg == cur_slot) return;
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");           // this wave's reads of the earlier slots are done
        if (lane == 0) vs[W_FREED + w] = (unsigned)sg;
        unsigned spins = 0;
        while (!dead && (int)min(min(vs[W_READY], vs[W_READY + 1]), min(vs[W_READY + 2], vs[W_READY + 3])) <= sg) {
            __builtin_amdgcn_s_sleep(1);
            if (++spins > SPIN_LIMIT) fail(0x500u + (unsigned)oi);
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
        cur_slot = sg;
    };
    // XT: the input vector is T values in `hvec` (down_proj) instead of fp32 planes in `xs`
    auto product = [&](const Op& o, int oi, bool xt, auto&& finalize) {
        const int nch = (o.rb / (int)sizeof(T)) / EPC;
        const int rpu = o.pair ? 2 : 1, nunits = o.nrows / rpu;
        for (int unit = w; unit < nunits; unit += NCONS) {
            float acc[2] = {0.0f, 0.0f};
            for (int r = 0; r < rpu; ++r) {
                int gb = (unit * rpu + r) * o.bpr, pb = 0;
                float a = 0.0f;
                while (pb < o.bpr) {
                    const int sg = o.slot0 + (gb >> 4), b0 = gb & (BPS - 1);
                    const int nseg = min(o.bpr - pb, BPS - b0);
                    enter_slot(sg, oi);
                    const char* wp = ring + (size_t)(sg % NS) * SLOT + b0 * BLK + lane * 16;
                    const int ci0 = pb * 64 + lane;
                    if (xt) {
                        const T* xp = hvec + (size_t)ci0 * EPC;
                        for (int t = 0; t < nseg; ++t) {
                            float wf[EPC], xf[EPC];
                            chunk_to_f32<T>(*(const uint4*)(wp + t * BLK), wf);
                            chunk_to_f32<T>(*(const uint4*)(xp + (size_t)t * 64 * EPC), xf);
#pragma unroll
                            for (int e = 0; e < EPC; ++e) a = fmaf(wf[e], xf[e], a);
                        }
                    } else {
                        const float* xp = xs + (size_t)ci0 * 4;
                        for (int t = 0; t < nseg; ++t) {
                            float wf[EPC], xf[EPC];
                            chunk_to_f32<T>(*(const uint4*)(wp + t * BLK), wf);
#pragma unroll
                            for (int q = 0; q < PARTS; ++q) {
                                const float4 v4 = *(const float4*)(xp + (size_t)q * nch * 4 + (size_t)t * 256);
                                xf[4 * q] = v4.x; xf[4 * q + 1] = v4.y; xf[4 * q + 2] = v4.z; xf[4 * q + 3] = v4.w;
                            }
#pragma unroll
                            for (int e = 0; e < EPC; ++e) a = fmaf(wf[e], xf[e], a);
                        }
                    }
                    pb += nseg; gb += nseg;
                }
                acc[r] = a;
            }
            const float r0 = wave_sum(acc[0]), r1 = o.pair ? wave_sum(acc[1]) : 0.0f;
            if (lane == 0) finalize(unit, r0, r1);
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");           // the op's slots are all released (no wait: the next op's product enters its own)
        if (lane == 0) vs[W_FREED + w] = (unsigned)(o.slot0 + o.nslots);
        cur_slot = -1;
    };
    auto unit_done = [&]() {                  // (lane 0 of the finalising wave) the unit's value is in outbuf
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
        lds_add_u32(sync0 + 4 * W_UNITS, 1u);
    };

    // (1) edge 0: merged attention output (qd values) -> xs, no norm
    if (w == 0) stamp(0);
    publish(0, perA, false);
    {
        const int nch = qd / EPC;
        gather(0, qd, [&](int k, float v) { xs[xs_index<T>(k, nch)] = v; });
    }
    cbarrier();
    if (w == 0) stamp(1);
    // (2) o_proj + residual -> x1
    product(op_o, 0, false, [&](int unit, float r0, float) {
        const int n = c * perH + unit;
        outbuf[unit] = to_f32(from_f32<T>(r0 + to_f32(xres[n])));
        unit_done();
    });
    units_target += perH;
    if (w == 0) stamp(2);
    // (3) edge 1: x1 -> xres, xs = post_attention_layernorm weight * x1, rstd
    publish(1, perH, true);
    cbarrier();                               // every wave is past its last read of xs / xres (o_proj) before the gather rewrites them
    const float rstd1 = gather_norm(1, (const T*)p.post_norm, true);
    if (w == 0) stamp(3);
    // (4) gate / up + SwiGLU -> hvec slice
    product(op_gu, 1, false, [&](int unit, float r0, float r1) {
        outbuf[unit] = to_f32(from_f32<T>(silu_f(r0 * rstd1) * (r1 * rstd1)));
        unit_done();
    });
    units_target += perI;
    if (w == 0) stamp(4);
    // (5) edge 2: SwiGLU product (I values) -> hvec
    publish(2, perI, true);
    gather(2, I, [&](int k, float v) { hvec[k] = from_f32<T>(v); });
    cbarrier();
    if (w == 0) stamp(5);
    // (6) down_proj + residual -> x2 (also to global x: the next layer's launch / the head read it there)
    product(op_dn, 2, true, [&](int unit, float r0, float) {
        const int n = c * perH + unit;
        const T vt = from_f32<T>(r0 + to_f32(xres[n]));
        outbuf[unit] = to_f32(vt);
        ((T*)p.x)[n] = vt;
        unit_done();
    });
    units_target += perH;
    if (w == 0) stamp(6);
    if (!has_next) {                          // last layer: x goes to the head; the launch still consumes its epoch
        if (c == 0 && w == 0 && lane == 0) *p.seq = seq + 1u;
        return;
    }
    // (7) edge 3: x2 -> xs = next input_layernorm weight * x2, rstd
    publish(3, perH, true);
    cbarrier();
    const float rstd2 = gather_norm(3, (const T*)p.next_norm, false);
    if (w == 0) stamp(7);
    // (8) the next layer's q | k | v rows (+ bias), un-roped: its attn_decode launch ropes q / k and appends k / v
    product(op_qkv, 3, false, [&](int unit, float r0, float) {
        const int n = c * perQ + unit;
        float v = r0 * rstd2;
        if (p.next_qkv_b) v += to_f32(((const T*)p.next_qkv_b)[n]);
        ((T*)p.qkv_out)[n] = from_f32<T>(v);
    });
    if (w == 0) stamp(8);
    if (c == 0 && w == 0 && lane == 0) *p.seq = seq + 1u;       // (every workgroup read seq before it could publish; the last edge is behind us)
}

template <typename T> constexpr int ring_slots() { return sizeof(T) == 2 ? 6 : 3; }
template <typename T> size_t layer_lds_bytes(const DecodeLayerArgs& a) {
    const int kx = a.H > a.qd ? a.H : a.qd;
    return (size_t)ring_slots<T>() * SLOT + (size_t)kx * 4 + (size_t)a.H * sizeof(T) + (size_t)a.I * sizeof(T) + 128 * 4 + W_N * 4;
}

}  // namespace

template <typename T> bool decode_layer_supported(const DecodeLayerArgs& a, int cus) {
    constexpr int VPG = Gran<T>::VPG;
    const int G = cus;
    auto rows_ok = [&](int K) { return (K * (int)sizeof(T)) % BLK == 0; };
    if (G < 1 || a.H % G || a.I % G || a.qd % G || a.qkv_dim % G) return false;
    if ((a.H / G) % VPG || (a.I / G) % VPG || (a.qd / G) % VPG) return false;
    if (a.H / G > 64 || a.I / G > 128 || a.qd / G > 64 || a.qkv_dim / G > 128) return false;     // a CU's outputs fit outbuf / one wave publishes them
    if (!rows_ok(a.H) || !rows_ok(a.I) || !rows_ok(a.qd)) return false;
    if (a.H % Elt<T>::PER_CHUNK || a.I % Elt<T>::PER_CHUNK) return false;
    return layer_lds_bytes<T>(a) <= (size_t)160 * 1024;
}
template <typename T> void launch_decode_layer(hipStream_t s, const DecodeLayerArgs& a, int cus, hipEvent_t start, hipEvent_t stop) {
    if (start || stop) hipExtLaunchKernelGGL((decode_layer_kernel<T, ring_slots<T>()>), dim3(cus), dim3(NTHREADS), layer_lds_bytes<T>(a), s, start, stop, 0, a);
    else hipLaunchKernelGGL((decode_layer_kernel<T, ring_slots<T>()>), dim3(cus), dim3(NTHREADS), layer_lds_bytes<T>(a), s, a);
}
void decode_layer_init_attrs() {
    set_max_lds((const void*)decode_layer_kernel<bf16, ring_slots<bf16>()>, 160 * 1024, NTHREADS);
    set_max_lds((const void*)decode_layer_kernel<float, ring_slots<float>()>, 160 * 1024, NTHREADS);
}
template bool decode_layer_supported<bf16>(const DecodeLayerArgs&, int);
template bool decode_layer_supported<float>(const DecodeLayerArgs&, int);
template void launch_decode_layer<bf16>(hipStream_t, const DecodeLayerArgs&, int, hipEvent_t, hipEvent_t);
template void launch_decode_layer<float>(hipStream_t, const DecodeLayerArgs&, int, hipEvent_t, hipEvent_t);

}  // namespace svln
