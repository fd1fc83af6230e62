// Host-side launch interface of the hand-written gfx950 kernels.  One launcher per kernel,
// each instantiated for T = bf16 (shipping) and T = float (parity mode).  Launchers only
// enqueue on `stream` (no allocation, no sync) so a caller may capture them in a hipGraph.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <vector>

namespace svln {

enum Epi : int { EPI_NONE = 0, EPI_GELU_TANH = 1, EPI_GELU_ERF = 2, EPI_SWIGLU = 3, EPI_ARGMAX = 4 };

struct RopeKvArgs;
// K / V^T pools of the ViT attention (layout of launch_vit_kv_pack): the fused tail of the SigLIP QKV product
struct VitPackArgs { void* Kpool; void* Vpool; int F, S, heads, head_dim; };
// C[M,N] = epi(A[M,K] . W[N,K]^T + bias[N]) + res[row % res_mod or row][N]      (all T, fp32 accumulate)
// K, lda, ldw multiples of one 16-byte chunk; EPI_SWIGLU: W rows are 32-row blocks
// [gate 32 | up 32] and C is [M, N/2] = silu(gate) * up.
struct GemmArgs {
    const void* A; int lda;
    const void* W; int ldw;
    void* C; int ldc;
    const void* bias;
    const void* res; int ldr; int res_mod;
    int M, N, K;
    int epi;
    float* ws; size_t ws_elems;   // split-K slab workspace (fp32) or null
    int nsplit, tile_base, launch_tiles;   // filled by the launcher (K splits; first column tile and tile count of this launch)
    int nt_w;                              // filled by the launcher: stage the weight tile with non-temporal loads (weights read once)
    int bn_fast;                           // filled by the launcher: consecutive workgroups walk the COLUMN tiles of one row tile (activations larger than weights)
    const void* zeros;            // >= 16 B of zeros in device memory: the K-tail source of the LDS-DMA staging (required)
    // optional fused RMSNorm of the finished output rows (o_proj -> post_attention_layernorm, down_proj -> next input_layernorm):
    // when the product takes the split-K path with N <= 4096 the slab reduce also writes norm_out = rmsnorm(C) * norm_w and
    // launch_gemm returns true; otherwise norm_out is untouched (false) and the caller runs launch_rmsnorm itself
    const void* norm_w; void* norm_out; float norm_eps;
    // optional fused tail of the QKV product of a prefill (Qwen2Attention: bias, RoPE on q / k, KV append): when the product takes the
    // split-K path the slab reduce also applies RoPE and appends k / v^T to the paged cache (rope->qkv must be C) and launch_gemm
    // returns true; otherwise C holds the plain product + bias and the caller runs launch_rope_kv itself
    const RopeKvArgs* rope;
    // optional fused tail of the SigLIP QKV product (siglip_encoder.py:197-232): when the product takes the split-K path the slab reduce
    // (sum + bias, rounded to T) also writes the K pages and the transposed V pages of the ViT attention -- splitk_epilogue followed by
    // vit_kv_pack in one pass -- and launch_gemm returns true; otherwise the caller runs launch_vit_kv_pack itself
    const VitPackArgs* vitpack;
    // optional with norm_out (bf16 engine, opt-in fp8 products): the fused reduce also writes the e4m3 copy of the normalised row and its
    // scale (max |y| / 448), exactly what launch_quant_fp8_rows produces from norm_out -- the A operand of the next product
    void* norm_q8 = nullptr; float* norm_q8_scale = nullptr;
    const void* norm_b;           // non-null: LayerNorm (mean/variance, weight norm_w, bias norm_b) instead of RMSNorm -- the ViT's ln1 / ln2
    // opt-in fp8 (OCP e4m3) operands (bf16 engine): when a_scale is set, A [M][K] and W [N][K] are e4m3 bytes (lda / ldw in
    // elements), C = a_scale[m] * w_scale[n] * (A . W^T) then the usual epilogue in bf16; K % 16 == 0
    const float* a_scale; const float* w_scale;
    // EPI_ARGMAX (M <= 32 rows, 32x128 tiles, no K split: the lm_head of several envs decoded together): no C; row m's (max, lowest index)
    // over the 128 columns of tile t goes to part_val / part_idx [m][tiles], reduced by launch_argmax_final_batched.  Optional repetition
    // penalty as in GemvBatchArgs.
    float* part_val = nullptr; int* part_idx = nullptr;
    const uint8_t* pen_flags = nullptr; const int* pen_rows = nullptr; float pen = 1.0f;
    VitPackArgs vp; int vp_on;    // filled by the launcher: device-side copy of *vitpack for the unsplit kernels that pack in their epilogue
    int force_cfg, force_split;   // tests: 0 = heuristic; force_cfg 129 -> 128x128 tiles with two in-workgroup K groups; force_cfg low bits 128 -> 128x128 tiles, 264 -> 256x64 tiles; force_split S -> 256x128 tiles, S splits
};
template <typename T> bool launch_gemm(hipStream_t s, const GemmArgs& a);   // true: a.norm_out was produced
template <typename T> int launch_gemm_argmax(hipStream_t s, GemmArgs a);    // EPI_ARGMAX form (M <= 32); returns the partials per row

// y[N] = epi(W[N,K] . x'[K] + bias) + res,  x' = x or rmsnorm(x) * norm_w (fused prologue).
// EPI_SWIGLU as above (y has N/2 entries).  EPI_ARGMAX: no y; per-workgroup (max, lowest index)
// partials go to part_val/part_idx, reduced by launch_argmax_final.
struct GemvArgs {
    const void* W; int ldw;
    const void* x;
    const void* norm_w; float eps;
    const void* bias;
    const void* res;
    void* y;
    int N, K;
    int epi;
    float* part_val; int* part_idx;
    // optional fp8 (OCP e4m3) weight-only path (bf16 engine, opt-in): w8 [N][K] bytes, one fp32 scale per row,
    // W[n][k] ~= scale[n] * e4m3(w8[n][k]); when set, W is ignored
    const void* w8; const float* scale;
    const int* skip;              // optional device flag: the launch is a no-op when *skip != 0 (generation finished: run-ahead decode steps)
    // EPI_ARGMAX only, optional: HF RepetitionPenaltyLogitsProcessor on the fp32 logits before the arg-max (a checkpoint's
    // generation_config.json `repetition_penalty`): pen_flags[n] != 0 marks token n as already generated in this turn;
    // its logit becomes v < 0 ? v * pen : v / pen.  null = no penalty.
    const uint8_t* pen_flags = nullptr; float pen = 1.0f;
};
template <typename T> void launch_gemv(hipStream_t s, const GemvArgs& a);
// per-row e4m3 quantisation of a bf16 matrix [rows][cols] (cols % 16 == 0): scale[r] = max|W[r]| / 448
void launch_quant_fp8_rows(hipStream_t s, const void* w_bf16, int ld, void* w8, float* scale, int64_t rows, int cols);
template <typename T> void launch_gemv_timed(hipStream_t s, const GemvArgs& a, hipEvent_t start, hipEvent_t stop);   // events get the kernel's own begin/end
int gemv_grid(int N);                       // workgroups launch_gemv uses for N rows

// B (1, 2, 4 or 8) activation vectors against one weight stream: x [B][ldx], y [B][ldy], res [B][ldr].
// EPI_ARGMAX: part_val / part_idx are [B][gemv_batched_grid(N, epi, B)]; launch_argmax_final_batched reduces them to B tokens.
struct GemvBatchArgs {
    const void* W; int ldw;
    const void* x; int ldx;
    const void* norm_w; float eps;
    const void* bias;
    const void* res; int ldr;
    void* y; int ldy;
    int N, K, epi, B;
    float* part_val; int* part_idx;
    // EPI_ARGMAX only, optional repetition penalty (see GemvArgs): row b of the batch uses the flag row pen_rows[b] of pen_flags [.][N]
    const uint8_t* pen_flags = nullptr; const int* pen_rows = nullptr; float pen = 1.0f;
};
template <typename T> void launch_gemv_batched(hipStream_t s, const GemvBatchArgs& a);
int gemv_batched_grid(int N, int epi, int B);
void launch_argmax_final_batched(hipStream_t s, const float* part_val, const int* part_idx, int n, int B, int* out_tokens);
void launch_argmax_final(hipStream_t s, const float* part_val, const int* part_idx, int n, int* out_token,
                         float* out_top /*[2]: best, runner-up of partial maxima (diagnostic)*/);
// Device-side state of one greedy generation (GenerationMixin._sample: append the arg-max, stop on EOS or max_new_tokens), so that
// decode steps can be enqueued ahead of the host: every kernel of a step is a no-op once `done` is set.
struct GenCtl {
    int pos;        // position of the token the next decode step feeds (read by the attention kernel: RoPE, KV append)
    int kv_len;     // keys visible to that step (= pos + 1)
    int done;       // set by the arg-max step that emitted EOS / the max_new-th token / a non-finite arg-max (-1)
    int count;      // tokens emitted so far (out_ids[0 .. count))
    int max_new, n_eos, pad0, pad1;
};
// launch_argmax_final + the bookkeeping above: out_ids[count++] = token; done |= token in eos[0 .. n_eos) || count == max_new || token < 0;
// otherwise pos / kv_len advance by one.  No-op when ctl->done is already set.
// pen_flags (optional): the emitted token's byte is set (repetition penalty of the following steps)
void launch_argmax_step(hipStream_t s, const float* part_val, const int* part_idx, int n, int* out_token, float* out_top, GenCtl* ctl,
                        const int* eos, int* out_ids, uint8_t* pen_flags = nullptr);
// flags[ids[k]] = value for k < *count (count: device scalar) or k < n_host when count is null; ids < 0 are skipped
void launch_set_flags(hipStream_t s, uint8_t* flags, const int* ids, const int* count, int n_host, int value);

// Flash-style attention over paged K / V^T tiles (64 keys per page).
//   pools:  K  [page][n_kv_total][64][HDP]        HDP = head dim padded to an even chunk count
//           Vt [page][n_kv_total][DT*32][64]      DT = ceil(HD/32); transposed so keys are contiguous
//   rows of one "kv head" kh: rho = i*G + g  (query position i, q-head g of the group)
//   Q/O element (frame, i, head, d) at  ((frame*T + i) * stride) + head*HD + d, head = (kh % hpf)*G + g
// one environment of a batched decode step
struct DecodeSlot { const int* page_table; int pos; int pad; };

constexpr int ATTN_PART_PAD = 4;    // floats after the HD partial outputs of a row: m, l, 2 unused
struct AttnArgs {
    const void* Q; int q_stride;
    void* O; int o_stride;
    const void* Kpool; const void* Vpool;
    const int* page_table;        // device [tiles]; null = identity
    int n_kv_total, hpf, G, T;
    int P;                        // absolute position of query 0 (causal mask: key <= P + i)
    int kv_len;
    const int* dyn_kv_len;        // device scalar overriding kv_len (P = kv_len - T); for hipGraph replay
    float scale;
    int causal;
    int nsplit, tiles_per_split;  // split-KV: gridDim.z = nsplit, partials in `part`
    float* part;                  // [nsplit][n_kv_total][rows_pad][HD + ATTN_PART_PAD] fp32 (O unnormalised, m, l, 2 pad: rows stay 16-byte aligned)
    int rows_pad;
    // decode fusion (T == 1, head_dim 128, one wave per workgroup): Q points at the UN-roped qkv row
    // [(nq + 2 nkv) * 128]; the kernel applies RoPE to q in registers, and the workgroup that owns the page of
    // position *dyn_pos ropes k, appends k / v to the pools (global) and patches its LDS tiles.
    int fuse_rope_append;
    const float* rope_tab;        // [max_positions][128]: cos[64] | sin[64]
    const int* dyn_pos;
    int nq_heads;
    // batched decode (gridDim.z = environments): per-env page table / position; Q, O and the partials advance by
    // q_stride, o_stride and part_bstride per environment
    const DecodeSlot* slots;
    size_t part_bstride;
    int batch;
    const int* skip;              // optional device flag: decode attention / combine are no-ops when *skip != 0
    int key_groups;               // > 1: split-KV inside the workgroup (64 query rows x key groups, merged through LDS); nsplit > 1 on top: the merge writes the split's partial row
};
template <typename T> void launch_attention(hipStream_t s, const AttnArgs& a, int head_dim, int waves);
template <typename T> void launch_attention_combine(hipStream_t s, const AttnArgs& a, int head_dim);
template <typename T> int attn_key_groups(int head_dim);       // the key-group count launch_attention<T> is built for at this head dim (AttnArgs::key_groups)

// Persistent batch-1 decode layer (decode_layer.hip): merge of this layer's split-KV attention partials, o_proj + residual,
// post_attention_layernorm, gate/up + SwiGLU, down_proj + residual, and the NEXT layer's input_layernorm + q|k|v projection, as ONE launch
// of one workgroup per CU (LDS-DMA loader ring + consumer waves, data-tagged granule all-gathers between the products).
struct DecodeLayerArgs {
    const float* part; int nsplit, tiles_per_split; const int* dyn_kv_len; int n_kv, Gq;      // attn_decode_kernel's partials [split][n_kv][32][132]
    const void *o_w, *post_norm, *gu_w, *down_w;                                            // this layer (engine dtype, packed as the engine holds them)
    const void *next_norm, *next_qkv_w, *next_qkv_b;                                        // next layer's input_layernorm / q|k|v; null for the last layer
    void* x;                      // residual stream [H]: read at the start, rewritten with the layer's output
    void* qkv_out;                // [qkv_dim]: the next layer's q | k | v rows (bias added, un-roped)
    int H, I, qd, qkv_dim; float eps;
    unsigned long long* gran[4];  // granule buffers of the four all-gather edges (qd, H, I, H values)
    unsigned* seq;                // launch counter (device): the epoch of the launch's granule tags
    unsigned* giveup;             // != 0: a bounded spin timed out (code); results are invalid
    const int* skip;              // optional device flag: no-op when *skip != 0 (run-ahead steps past the end of the generation)
    unsigned long long* dbg;      // optional [workgroups][16] phase stamps (100 MHz clock): svln_probe_decode_layer
};
template <typename T> bool decode_layer_supported(const DecodeLayerArgs& a, int cus);
template <typename T> void launch_decode_layer(hipStream_t s, const DecodeLayerArgs& a, int cus, hipEvent_t start = nullptr, hipEvent_t stop = nullptr);
void decode_layer_init_attrs();

// RMSNorm / LayerNorm over rows of length n (T in, T out).
// y2 / y2_row / y2_cap: optional second copy of the rows at row *y2_row (device scalar) of y2, clamped to y2_cap rows
template <typename T> void launch_rmsnorm(hipStream_t s, const void* x, const void* g, void* y, int rows, int n, float eps, const int* skip = nullptr,
                                          void* y2 = nullptr, const int* y2_row = nullptr, int y2_cap = 0);
template <typename T> void launch_layernorm(hipStream_t s, const void* x, const void* g, const void* b, void* y, int rows, int n, float eps);

// RoPE on q,k (in place on q) + append roped k and v to the paged cache.
// qkv: [T][(nq + 2 nkv) * 128]; positions P + i (P = *dyn_pos if given).
struct RopeKvArgs {
    void* qkv; int ld;
    void* Kpool; void* Vpool;
    const int* page_table;
    const float* rope_tab;        // [max_positions][128]: cos[64] | sin[64]  (launch_rope_table)
    int T, nq, nkv, P;
    const int* dyn_pos;
};
template <typename T> void launch_rope_kv(hipStream_t s, const RopeKvArgs& a);
void launch_rope_table(hipStream_t s, float* tab, const float* inv_freq, int positions);
// out[2f], out[2f+1] += 128-bit content key of frame f (caller zeroes `out` first)
void launch_frame_hash(hipStream_t s, const float* pix, int F, size_t words_per_frame, unsigned long long* out);

// ViT: qkv [F*S][3*Hv] -> K pages [page][F*heads][64][HDP], Vt pages [page][F*heads][96][64]
template <typename T> void launch_vit_kv_pack(hipStream_t s, const void* qkv, int ld, void* Kpool, void* Vpool, int F, int S,
                                             int heads, int head_dim);

// pixels [F,3,S,S] (fp32) -> patches [F*side*side][kp] (T), k = c*p*p + ky*p + kx, zero padded to kp
template <typename T> void launch_patchify(hipStream_t s, const float* pix, void* out, int F, int image, int patch, int kp);

// bilinear 2-D pooling of token grids: in [F][side*side][C] -> out [F][out*out][C]
// taps: device int2/float2 tables [out] (i0,i1) and (w0,w1), shared by both axes.
template <typename T> void launch_pool(hipStream_t s, const void* in, void* out, const int* tap_idx, const float* tap_w, int F,
                                       int side, int out_side, int C);

// out[r][:] = src[r] >= 0 ? embed[src[r]][:] : feats[-(src[r]+1)][:]
// opt-in slow-memory pruning (misc.hip): sel[0..keep) = ascending indices of the `keep` rows of m [n_rows][H] least similar (cosine) to the mean row
template <typename T> void launch_memory_prune(hipStream_t s, const void* m, int n_rows, int H, int keep, float* partial, float* mean, float* score,
                                               int* sel);
template <typename T> void launch_gather_rows(hipStream_t s, const int* src, const void* embed, const void* feats, void* out,
                                              int rows, int n, const int* skip = nullptr);

// weights: synthesize (see weights.py) or convert canonical rows into packed rows
// dst row = (r / blk) * blk * nint + phase * blk + r % blk ; cols copied to [0, cols), dst_ld >= cols
struct RowMap { int blk, nint, phase; };
template <typename T> void launch_synth(hipStream_t s, void* dst, int dst_ld, int64_t rows, int cols, RowMap m, uint64_t seed_t,
                                        float half_width, float base);
template <typename T> void launch_convert(hipStream_t s, void* dst, int dst_ld, int64_t rows, int cols, RowMap m, const void* src,
                                          int src_is_f32);
template <typename T> void launch_to_f32(hipStream_t s, const void* src, float* dst, int64_t n);
template <typename T> void launch_from_f32(hipStream_t s, const float* src, void* dst, int64_t n);

// a-1 on the GPU (preprocess.hip): Pillow's two-pass fixed-point bicubic resize + rescale / normalise, bit-exact.
// ResampleAxis = the host-built coefficient table of one axis (precompute_coeffs + normalize_coeffs_8bpc of Pillow's Resample.c):
// for output index i, source taps xmin[i] .. xmin[i]+cnt[i]-1 with 22-bit fixed-point weights k[i*ksize ..].
struct ResampleAxis { int ksize = 0; std::vector<int> xmin, cnt, k; };
struct ResampleDev { const int *hmin, *hcnt, *hk, *vmin, *vcnt, *vk; int ks_h, ks_v; };   // device copies (h = along the width, v = along the height)
void build_resample_table(int in_size, int out_size, ResampleAxis& ax);
void build_normalize_lut(float* lut256, float mean, float std);
size_t preprocess_lds_bytes(int W, int S, int ks_v, int ks_h);
// rgb uint8 [n][H][W][3] (device, readable up to 16 bytes past its end) -> out fp32 [n][3][S][S]
void launch_preprocess(hipStream_t s, const uint8_t* rgb, float* out, int n_frames, int H, int W, int S, const ResampleDev& t, const float* lut);
void launch_upload(hipStream_t s, const void* src_dev_visible, void* dst, size_t bytes);    // 16-byte granules; src may be pinned host memory
void preprocess_init_attrs();

// raise the dynamic-LDS limit of the kernels that may ask for more than 64 KiB (call once, outside capture).  A refused attribute would
// only show up later as a failed launch of that one kernel: the first refusal is kept and returned by init_kernel_attributes().
inline hipError_t& attr_status() { static hipError_t e = hipSuccess; return e; }
// threads > 0: also check, on the host, that the code object accepts a block of that size (its max flat workgroup size comes from
// __launch_bounds__): a launch above it -- or above the dynamic-LDS attribute -- is not reported by hipLaunchKernel; the packet
// processor rejects the packet and the runtime aborts the process from its queue-error callback, with no message at the default log
// level (the round-2 test_gemm abort).
inline void set_max_lds(const void* fn, int bytes, int threads = 0) {
    hipError_t e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
    if (e == hipSuccess && threads > 0) {
        hipFuncAttributes at;
        e = hipFuncGetAttributes(&at, fn);
        if (e == hipSuccess && (at.maxThreadsPerBlock < threads || bytes > 160 * 1024)) e = hipErrorInvalidConfiguration;
    }
    if (e != hipSuccess && attr_status() == hipSuccess) attr_status() = e;
}
void gemm_init_attrs();
void gemv_init_attrs();
void attention_init_attrs();
inline hipError_t init_kernel_attributes() {
    attr_status() = hipSuccess;
    gemm_init_attrs(); gemv_init_attrs(); attention_init_attrs(); preprocess_init_attrs(); decode_layer_init_attrs();
    return attr_status();
}

}  // namespace svln
