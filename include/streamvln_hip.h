/* streamvln_hip.h -- C ABI of the MI355X-native StreamVLN streaming-inference engine.
 *
 * The reference has no FFI: its hot path sits behind the Python class
 * `StreamVLNForCausalLM` (streamvln/model/stream_video_vln.py).  This library is what sits
 * UNDER a Python class of the same shape (streamvln_amd/model.py); each entry point names the
 * reference interface it replaces.  Plain pointers and sizes only; device pointers are raw HIP
 * device addresses; no torch types.  All functions return 0 on success, < 0 on error (message
 * via svln_last_error()).  One engine = one GPU = one HIP stream; not re-entrant per engine.
 */
#ifndef STREAMVLN_HIP_H
#define STREAMVLN_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct svln_engine svln_engine;

enum { SVLN_BF16 = 0, SVLN_F32 = 1 };

/* Model dimensions (streamvln_amd/config.py; reference: siglip_encoder.py:73-86, Qwen2-7B). */
typedef struct svln_config {
    int32_t v_hidden, v_inter, v_heads, v_layers, v_patch, v_image;
    float v_eps;
    int32_t hidden, layers, q_heads, kv_heads, head_dim, inter, vocab;
    float rope_theta, rms_eps;
    int32_t max_positions;   /* KV / embeds capacity per env (model_max_length, streamvln_eval.py:501) */
    int32_t max_envs;        /* model.reset(env_num), stream_video_vln.py:473 */
    int32_t max_frames;      /* views per generate call: 1 + num_history */
    int32_t dtype;           /* SVLN_BF16 (shipping) or SVLN_F32 (parity mode) */
} svln_config;

/* -- lifetime: StreamVLNForCausalLM.from_pretrained(...).to(device) (streamvln_eval.py:523-533) -- */
int svln_create(const svln_config* cfg, int device, svln_engine** out);
void svln_destroy(svln_engine* h);
const char* svln_last_error(void);
int svln_sync(svln_engine* h);

/* -- weights: HF state-dict names (model.layers.N.self_attn.q_proj.weight, ...).
 * svln_synth_tensor fills a tensor on the device from the counter-based generator of
 * streamvln_amd/weights.py (seed_t = fnv1a64(name) ^ splitmix64(seed)); svln_set_tensor uploads
 * a canonical row-major tensor (host or device memory, fp32 or bf16). */
int svln_synth_tensor(svln_engine* h, const char* name, uint64_t seed_t, float half_width, float base);
int svln_set_tensor(svln_engine* h, const char* name, const void* data, int dtype, int64_t numel, int on_device);
int svln_weights_ready(svln_engine* h);            /* 0 when every tensor of the path has been provided */
int svln_get_tensor_f32(svln_engine* h, const char* name, float* host_out, int64_t numel);   /* canonical order */

/* -- session state: model.reset(env_num) / model.reset_for_env(i) (stream_video_vln.py:473-479);
 * svln_kv_reset = caller passing past_key_values=None (streamvln_eval.py:349). */
int svln_reset_env(svln_engine* h, int env);
int svln_kv_reset(svln_engine* h, int env);
int svln_env_state(svln_engine* h, int env, int32_t* n_embeds, int32_t* kv_len);

/* -- vision: encode_rgbd (stream_video_vln.py:102-142) minus the memory/image split:
 * pixels fp32 [F,3,S,S] (device or host) -> F*196 pooled rows kept in the engine's frame buffer. */
int svln_encode_frames(svln_engine* h, const float* pixels, int n_frames, int on_device);

/* -- image preprocess: SigLipImageProcessor.preprocess (llava/model/multimodal_encoder/siglip_encoder.py:47-67) =
 * PIL bicubic resize of the camera frame to v_image x v_image (aspect not preserved), x/255, (x - 0.5)/0.5, channels first.
 * rgb uint8 [n_frames][height][width][3] (host memory, or device memory when on_device) -> out_dev fp32 [n_frames][3][S][S]
 * (device).  Bit-exact with Pillow's two-pass fixed-point resampler (pillow==11.2.1, requirements.txt:97); complete on return.
 * svln_preprocess_time: accumulated GPU time (upload + kernel, HIP events) and frame count since the last reset. */
int svln_preprocess_frames(svln_engine* h, const uint8_t* rgb, int n_frames, int height, int width, int on_device, float* out_dev);
/* The same work without the final wait: returns once the frame bytes have been consumed (copied to pinned staging) and the upload +
 * kernel are enqueued on the engine's stream.  svln_encode_frames / svln_generate on the same engine are ordered behind it; any other
 * stream that touches out_dev must first wait on the engine's stream (svln_engine_stream: the hipStream_t as a void*).
 * Lifetime of rgb: with on_device = 0 the bytes have been copied when the call returns and the caller may reuse the buffer.  With
 * on_device = 1 the engine's stream reads the caller's DEVICE buffer asynchronously: it must stay allocated and unmodified until the
 * work enqueued here has run (wait on svln_engine_stream, or call svln_sync), and whatever stream produced it must have finished --
 * or the engine's stream must have been ordered behind it -- before this call. */
int svln_preprocess_frames_enqueue(svln_engine* h, const uint8_t* rgb, int n_frames, int height, int width, int on_device, float* out_dev);
int svln_engine_stream(svln_engine* h, void** stream);
/* Engine-owned frame ring: `slots` frames of height x width x 3 bytes in pinned, device-mapped host memory (*host_base, slots
 * *slot_stride bytes apart, 256-byte aligned).  The camera / simulator side writes its RGB frames into the slots; a frame passed to
 * svln_preprocess_frames[_enqueue] (on_device = 0) whose bytes lie inside the ring is read by the GPU where it is -- the staging copy on
 * the host (921 KB per 640x480 frame) disappears.  Any other host pointer takes the staging path as before.  A slot may be rewritten
 * once the upload that last read it has run: svln_frame_ring_wait(slot) returns when that is the case (immediately if none is pending).
 * A second call replaces the ring (the old memory is freed). */
int svln_frame_ring(svln_engine* h, int slots, int height, int width, uint8_t** host_base, int64_t* slot_stride);
int svln_frame_ring_wait(svln_engine* h, int slot);
int svln_preprocess_time(svln_engine* h, double* gpu_ms, int64_t* frames, int reset);

/* -- splice: prepare_inputs_labels_for_multimodal (stream_video_vln.py:182-238) for one env.
 * ids hold text tokens and the sentinels -200 (<image>) / -300 (<memory>); the first n_memory frames
 * of the last svln_encode_frames call form the memory block, the rest are consumed by <image> in order.
 * Rows are appended to the env's inputs_embeds (stream_video_vln.py:396-401). */
int svln_append_turn(svln_engine* h, int env, const int64_t* ids, int n_ids, int n_memory);
/* config.tokenizer_model_max_length of the reference (stream_video_vln.py:241-244): the spliced rows of ONE turn are truncated to
 * `rows` before they are appended (new_input_embeds[:tokenizer_model_max_length]).  0 = no truncation (the attribute is None, the
 * reference's default).  Independent of max_positions, the capacity of the accumulated sequence, which the reference does not have:
 * exceeding it is an error ("inputs_embeds exceeds max_positions"). */
int svln_set_turn_row_limit(svln_engine* h, int rows);

/* -- greedy generation: StreamVLNForCausalLM.generate -> GenerationMixin (do_sample=False, num_beams=1):
 * prefill embeds[kv_len:], arg-max, feed generated ids until one is in eos_ids (appended, not fed) or
 * max_new_tokens; afterwards kv_len = n_embeds + n_out - 1. */
int svln_generate(svln_engine* h, int env, int max_new_tokens, const int64_t* eos_ids, int n_eos, int64_t* out_ids,
                  int out_cap, int32_t* n_out);
/* -- ONE call per model turn: what StreamVLNForCausalLM.generate does between the harness's call and its return
 * (stream_video_vln.py:353-407) = svln_encode_frames(pixels) ; new_window (the caller passed past_key_values=None): svln_kv_reset ;
 * new_episode (curr_t == 0) and the env holds rows: svln_reset_env ; svln_append_turn(ids, n_memory) ; svln_generate.  *kv_len (optional)
 * = the env's cache length afterwards (the KV handle the Python class hands back).  Same results and errors as the five calls. */
typedef struct svln_turn_args {
    const float* pixels; int32_t n_frames; int32_t pixels_on_device;     /* fp32 [n_frames,3,S,S] */
    int32_t env;
    const int64_t* ids; int32_t n_ids; int32_t n_memory;                /* text ids + sentinels; first n_memory frames = <memory> block */
    int32_t new_window, new_episode;
    int32_t max_new_tokens;
    const int64_t* eos_ids; int32_t n_eos;
} svln_turn_args;
int svln_turn(svln_engine* h, const svln_turn_args* a, int64_t* out_ids, int out_cap, int32_t* n_out, int32_t* kv_len);
/* generation_config.repetition_penalty of a checkpoint: transformers' RepetitionPenaltyLogitsProcessor -- applied by
 * GenerationMixin under greedy decoding too -- on the fp32 logits of every step, over the ids generated so far in the turn (the prompt is
 * passed as inputs_embeds, so it has no ids): logit < 0 ? logit * penalty : logit / penalty.  1 = off (default).  Applies to svln_generate,
 * svln_generate_batch and the scheduler; cannot change while scheduler turns are in flight. */
int svln_set_repetition_penalty(svln_engine* h, float penalty);
/* perf harness variant: decode exactly n_tokens regardless of EOS */
int svln_generate_fixed(svln_engine* h, int env, int n_tokens, int64_t* out_ids);

/* -- multi-env lockstep turns (SURVEY.md 8f-1 / BASELINE configs[4]; build-side extension, the reference runs batch 1):
 * svln_append_turn_at = svln_append_turn with the env's frames starting at `frame_base` of the last svln_encode_frames
 * call (several envs' frames encoded together); svln_generate_batch = svln_generate on each listed env (<= 8, distinct),
 * executed together: dense layers of all prefill rows at once, then batched decode steps that stream each weight matrix
 * once per step for all still-active envs.  out_ids is [n_envs][out_cap], n_out is [n_envs]. */
int svln_append_turn_at(svln_engine* h, int env, const int64_t* ids, int n_ids, int frame_base, int n_memory);
int svln_generate_batch(svln_engine* h, const int32_t* envs, int n_envs, int max_new_tokens, const int64_t* eos_ids, int n_eos,
                        int64_t* out_ids, int out_cap, int32_t* n_out);
/* -- the scheduler underneath svln_generate_batch, for callers whose envs' turns fall due at DIFFERENT times (a DAgger-style
 * collector mixing expert and model steps per env, streamvln_dagger.py:232-313): iteration-level batching.
 * svln_batch_submit: the env (turn already appended with svln_append_turn[_at]) joins the next iteration; *slot identifies the turn.
 * svln_batch_step: ONE pass over the weights carrying, for every turn in flight, either its prefill rows or the row of the token it
 * generated in the previous iteration (prefilling and decoding envs share the pass); *running = turns still in flight afterwards,
 * finished_slots[0 .. *n_finished) = turns that emitted EOS / max_new_tokens in this iteration (<= 8 entries).
 * svln_batch_result: ids of a finished turn (frees the slot).  Per-env results are exactly those of svln_generate. */
int svln_batch_submit(svln_engine* h, int env, int max_new_tokens, const int64_t* eos_ids, int n_eos, int32_t* slot);
int svln_batch_step(svln_engine* h, int32_t* running, int32_t* finished_slots, int32_t* n_finished);
int svln_batch_result(svln_engine* h, int slot, int32_t* env, int64_t* out_ids, int out_cap, int32_t* n_out);
/* Drop the turn in `slot` (slot < 0: every turn in flight), finished or not; its slot is free again.  svln_reset_env / svln_kv_reset drop
 * the env's turn themselves.  A svln_batch_step that fails part-way drops every turn in flight before it returns the error (a turn whose
 * env has no rows left to prefill is dropped alone), so the scheduler is always usable after an error. */
int svln_batch_cancel(svln_engine* h, int slot);
int svln_get_hidden_batch(svln_engine* h, int slot, float* host_out, int max_rows, int32_t* n_rows);   /* parity tap, <= 8 rows */

/* -- parity taps (test infrastructure reads these; not used by the product path) */
int svln_get_hidden(svln_engine* h, float* host_out, int max_rows, int32_t* n_rows);  /* final-norm hidden per generated token of the last generate */
int svln_get_embeds(svln_engine* h, int env, int start_row, int n_rows, float* host_out);
int svln_get_frame_feats(svln_engine* h, int start_row, int n_rows, float* host_out);
int svln_get_top2(svln_engine* h, float* host_out2);
/* prefill taps of svln_generate (single env): enable != 0 records the LAST row of the residual stream after every decoder layer of the
 * next prefills (svln_get_layer_taps: host_out [layers][hidden]); probe_layer >= 0 additionally records, for every row of the prefill,
 * the operands the products of that one layer actually saw (svln_get_layer_probe, which: 0 = x entering the layer, 1 = x leaving it,
 * 2 = attention output [q_heads * 128], 3 = x after the attention residual, 4 = post_attention_layernorm(x), 5 = silu(gate) * up
 * [inter], 6 = input_layernorm(x), 7 = the q | k | v rows after bias and RoPE [(q_heads + 2 kv_heads) * 128]; the others [hidden]),
 * so that each fused stage can be checked against the oracle on the engine's own inputs, at its own scale.  enable = 0 switches
 * both off. */
int svln_set_layer_taps(svln_engine* h, int enable, int probe_layer);
int svln_get_layer_taps(svln_engine* h, float* host_out);
int svln_get_layer_probe(svln_engine* h, int which, float* host_out, int64_t max_elems, int32_t* n_rows, int32_t* n_cols);

/* -- decode execution mode + timing probes (bench.py) */
int svln_set_decode_graph(svln_engine* h, int enable);     /* replay the per-token decode step as a hipGraph */
/* Execution form of the single-env decode step (same arithmetic per output, different summation order over K): enable != 0 runs, per
 * layer, the decode attention launch followed by ONE persistent launch (one workgroup per CU: LDS-DMA weight ring + consumer waves,
 * granule all-gathers between the products) for the merge of the attention partials, o_proj, gate/up + SwiGLU, down_proj and the next
 * layer's q|k|v, instead of six launches.  Needs the whole GPU (every workgroup must be resident); a hand-off that times out makes
 * svln_generate / svln_turn fail.  Refused for shapes it does not cover.  With svln_set_fp8_decode on, the launched GEMVs are kept. */
int svln_set_decode_persistent(svln_engine* h, int enable);
/* diagnostic (tools/persist_probe.py): one persistent launch of `layer` with per-workgroup phase stamps, out [n_wgs][16] ticks of the
 * 100 MHz wall clock: [0] merge done, [1] edge 0 gathered, [2] o_proj done, [3] edge 1 gathered, [4] gate/up done, [5] edge 2 gathered,
 * [6] down_proj done, [7] edge 3 gathered, [8] q|k|v done (consumer wave 0); [10..14] the loader at the start / after each product's
 * stream.  Runs on whatever the engine's buffers hold (after a decode step); overwrites the residual row and the q|k|v buffer. */
int svln_probe_decode_layer(svln_engine* h, int layer, unsigned long long* out, int max_wgs, int32_t* n_wgs);
/* Opt-in, no reference counterpart: the single-env decode step and the lm_head stream OCP e4m3 copies of the LLM
 * weights (one fp32 scale per output row, quantised on the device from the loaded tensors at the first enable) instead of the bf16
 * ones -- half the HBM bytes per generated token.  bf16 engines only; prefill, vision and svln_generate_batch keep bf16 weights. */
int svln_set_fp8_decode(svln_engine* h, int enable);
/* Opt-in, no reference counterpart (BASELINE configs[4] "fp8 MFMA on QKV/MLP GEMMs"): the LLM's dense products with more
 * than one row -- prefill, and the decode steps of >= 4 envs batched by svln_generate_batch / svln_batch_step -- run as e4m3 x e4m3 MFMA
 * products (fp32 accumulate, bf16 out) on the e4m3 weight copies above with per-row activation scales computed on the fly.  bf16 engines
 * only; vision, attention, norms, lm_head and the batch-1 decode GEMVs are unaffected (the latter have svln_set_fp8_decode). */
int svln_set_fp8_gemm(svln_engine* h, int enable);
/* Opt-in slow-memory pruning (BASELINE configs[3]; the reference has NO counterpart -- its memory is all num_history x 196 pooled
 * tokens, streamvln_eval.py:313-321 -- so this is pinned only by the project's own CPU restatement, oracle: prune_memory_tokens):
 * with keep_tokens > 0 a `<memory>` sentinel expands to the keep_tokens memory tokens least similar (cosine) to the mean memory
 * token, in their original order (ties: lower index).  0 (default) = the reference behaviour. */
int svln_set_memory_prune(svln_engine* h, int keep_tokens);
int svln_probe_reset(svln_engine* h);
int svln_probe_read(svln_engine* h, double* total_ms, int64_t* launches, double* bytes_per_launch);
/* second probe armed by svln_probe_reset: the layer-0 gate/up product of every steady prefill (<= 256 rows) between two stream events:
 * total ms, how many, their mean row count, flops of one (2 * rows * 2 * inter * hidden) and its weight bytes */
int svln_probe_read_prefill(svln_engine* h, double* total_ms, int64_t* count, double* mean_rows, double* flops, double* weight_bytes);
int svln_phase_times(svln_engine* h, double* vision_ms, double* prefill_ms, double* decode_ms, int reset);

/* -- optional memoisation of pooled frame features keyed by a 128-bit content hash of the pixels:
 * the <memory> frames of a window restart were all encoded earlier as "current" frames, so with the cache on they
 * skip the ViT.  capacity_frames = 0 (default) disables it: every frame is re-encoded, as the reference does
 * (stream_video_vln.py:104). */
int svln_set_feature_cache(svln_engine* h, int capacity_frames);
int svln_feature_cache_stats(svln_engine* h, int64_t* hits, int64_t* misses);

/* -- single-kernel entry points (device pointers in the engine dtype) for the op-level parity tests */
/* force_cfg (op tests and A/B runs only; the engine always passes 0): low 12 bits = tile configuration, 0 = heuristic, 128 = 128x128,
 * 129 = 128x128 with two in-workgroup K groups, 256 = 256x256 (bf16: 8-phase schedule), 258 = 256x256 with two K slices, 264 = 256x64
 * (M <= 256), 64 = 64x64, 32 = 32x128 (M <= 32); flag bits: 0x1000 row tiles fastest in the workgroup order, 0x10000 column tiles
 * fastest, 0x4000 stage-ring kernel for the 256x256 tile, 0x8000 32x32x16 form of the 8-phase schedule, 0x20000 direct 2-byte stores in the 8-phase
 * epilogue instead of the LDS-staged 16-byte row chunks.
 * force_split: 0 = heuristic, S >= 1 = 256x128 tiles (256x64 with force_cfg 264) with S K-splits */
int svln_op_gemm(svln_engine* h, const void* A, int lda, const void* W, int ldw, void* C, int ldc, const void* bias, const void* res,
                 int ldr, int res_mod, int M, int N, int K, int epi, int force_cfg, int force_split);
/* C = A . W^T + bias + res, and -- when the product takes the split-K path (few rows, N <= 4096) -- norm_out = norm(C) from the same slab
 * reduce (*fused = 1): RMSNorm with weight norm_w when norm_b is null (Qwen2 o_proj -> post_attention_layernorm, down_proj ->
 * input_layernorm, modeling_qwen2.py:269-299), LayerNorm with weight norm_w and bias norm_b otherwise (SigLIP out_proj -> layer_norm2,
 * fc2 -> next layer_norm1, siglip_encoder.py:269-305).  Otherwise norm_out is left untouched (*fused = 0) and the caller runs
 * svln_op_rmsnorm / svln_op_layernorm. */
int svln_op_gemm_norm(svln_engine* h, const void* A, int lda, const void* W, int ldw, void* C, int ldc, const void* bias, const void* res, int ldr,
                      const void* norm_w, const void* norm_b, void* norm_out, float eps, int M, int N, int K, int force_split, int* fused);
/* the RMSNorm form with the e4m3 copy of the normalised rows (opt-in fp8 products: the reduce that emits the norm also quantises it):
 * q8 [M][N] bytes, q8_scale [M] = max |norm_out row| / 448 */
int svln_op_gemm_norm_q8(svln_engine* h, const void* A, int lda, const void* W, int ldw, void* C, int ldc, const void* res, int ldr,
                         const void* norm_w, void* norm_out, float eps, int M, int N, int K, int force_split, void* q8, float* q8_scale, int* fused);
int svln_op_gemv(svln_engine* h, const void* W, int ldw, const void* x, const void* norm_w, float eps, const void* bias, const void* res,
                 void* y, int N, int K, int epi, int32_t* host_token);
/* the product behind svln_set_fp8_gemm: C [M][N] (bf16) = epi(a_scale[m] * w_scale[n] * (A8 [M][K] . W8 [N][K]^T) + bias) + res, e4m3 operands
 * (svln_op_quant_fp8 makes them), epi = EPI_NONE or EPI_SWIGLU, K % 16 == 0 */
int svln_op_gemm_fp8(svln_engine* h, const void* A8, const float* a_scale, int lda, const void* W8, const float* w_scale, int ldw, void* C, int ldc,
                     const void* bias, const void* res, int ldr, int M, int N, int K, int epi, int force_cfg, int force_split);
/* B (1, 2, 4 or 8) activation vectors x [B][ldx] against one weight stream (the decode step of svln_generate_batch / svln_batch_step at
 * B <= 2, and its lm_head at every B): y [B][ldy], res [B][ldr]; EPI_ARGMAX writes one token per vector to host_tokens[B] */
int svln_op_gemv_batched(svln_engine* h, const void* W, int ldw, const void* x, int ldx, const void* norm_w, float eps, const void* bias,
                         const void* res, int ldr, void* y, int ldy, int N, int K, int epi, int B, int32_t* host_tokens);
/* the selection step of svln_set_memory_prune on mem [n_rows][hidden] (engine dtype, device): out_idx[keep] ascending row indices
 * (host), out_score [n_rows] cosine scores (host, optional) */
int svln_op_memory_prune(svln_engine* h, const void* mem, int n_rows, int keep, int32_t* out_idx, float* out_score);
/* fp8 weight-only pieces of svln_set_fp8_decode: per-row e4m3 quantisation of a bf16 matrix [rows][cols] (cols % 16 == 0,
 * scale[r] = max|W[r]| / 448, round to nearest even), and the GEMV over such a matrix (same epilogues as svln_op_gemv) */
int svln_op_quant_fp8(svln_engine* h, const void* w_bf16, int64_t rows, int cols, void* w8, float* scale);
int svln_op_gemv_fp8(svln_engine* h, const void* w8, const float* scale, int ldw, const void* x, const void* norm_w, float eps, const void* bias,
                     const void* res, void* y, int N, int K, int epi, int32_t* host_token);
int svln_op_rmsnorm(svln_engine* h, const void* x, const void* g, void* y, int rows, int n, float eps);
int svln_op_layernorm(svln_engine* h, const void* x, const void* g, const void* b, void* y, int rows, int n, float eps);
/* attention over caller-provided q [T][q_stride] and k/v [S][kv_stride] (engine packs them into pages):
 * llm: rope = 1 applies RoPE at positions P + i to q and k (head_dim 128, GQA G = nq/nkv, causal)
 * vit: head_dim 72, non-causal, frames * heads */
int svln_op_attention_llm(svln_engine* h, void* qkv, int ld, int T, int P, const void* ctx_qkv, int ctx_T, void* out, int o_stride,
                          int nsplit);
int svln_op_attention_vit(svln_engine* h, const void* qkv, int ld, int F, void* out, int o_stride);
int svln_op_pool(svln_engine* h, const void* in, void* out, int F);
int svln_op_patchify(svln_engine* h, const float* pix, void* out, int F);

#ifdef __cplusplus
}
#endif
#endif
