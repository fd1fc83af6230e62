// StreamVLN streaming-inference engine: weights, workspaces, paged KV cache, per-env session state,
// the vision / prefill / decode schedules, and the C ABI of include/streamvln_hip.h.
//
// Replaces (reference file:line under /root/reference):
//   encode_rgbd + vision tower + projector + get_2dPool   streamvln/model/stream_video_vln.py:53-142
//   prepare_inputs_labels_for_multimodal (splice)         stream_video_vln.py:144-291
//   generate / prepare_inputs_for_generation / reset*     stream_video_vln.py:353-479
//   Qwen2 decoder stack + DynamicCache + greedy _sample   transformers 4.45.1 (requirements.txt:140)
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <stdexcept>
#include <string>
#include <unordered_map>
#include <vector>

#include "../../include/streamvln_hip.h"
#include "common.h"
#include "kernels.h"

namespace svln {

static thread_local std::string g_err;

#define HIP_CHECK(x)                                                                                   \
    do {                                                                                               \
        hipError_t e_ = (x);                                                                           \
        if (e_ != hipSuccess)                                                                          \
            throw std::runtime_error(std::string(#x) + ": " + hipGetErrorString(e_) + " at " + __FILE__ + ":" + std::to_string(__LINE__)); \
    } while (0)
#define REQUIRE(c, msg)                                          \
    do {                                                         \
        if (!(c)) throw std::runtime_error(std::string(msg));    \
    } while (0)

// The engine's buffers, stream and launches belong to ONE device; every API call runs with that device current and restores the
// caller's (torch's) current device on the way out.
struct DeviceGuard {
    int prev = -1, dev;
    explicit DeviceGuard(int d) : dev(d) {
        if (hipGetDevice(&prev) != hipSuccess) prev = -1;
        if (prev != dev) HIP_CHECK(hipSetDevice(dev));
    }
    ~DeviceGuard() { if (prev >= 0 && prev != dev) (void)hipSetDevice(prev); }
};
// launches are asynchronous: a bad configuration (LDS size, grid) only shows up in hipGetLastError
#define LAUNCH_CHECK(what)                                                                              \
    do {                                                                                               \
        hipError_t e_ = hipGetLastError();                                                             \
        if (e_ != hipSuccess) throw std::runtime_error(std::string(what) + ": kernel launch failed: " + hipGetErrorString(e_)); \
    } while (0)

constexpr int IMAGE_TOKEN = -200, MEMORY_TOKEN = -300;   // streamvln/utils/utils.py:9,15
constexpr int PAGE = 64;                                 // keys per KV page
constexpr int HID_TAP_ROWS = 64;
constexpr int PREFILL_SPLIT_ROWS = 2048;                 // split-KV prefill only when T * G fits this many rows

struct Slot {            // canonical tensor -> where its rows live in the packed device layout
    void* dst; int ld; int64_t rows; int cols; RowMap map; bool filled;
};

struct EngineBase {
    virtual ~EngineBase() {}
    virtual void synth_tensor(const char* name, uint64_t seed_t, float hw, float base) = 0;
    virtual void set_tensor(const char* name, const void* data, int dtype, int64_t numel, int on_device) = 0;
    virtual int weights_missing() = 0;
    virtual void get_tensor_f32(const char* name, float* out, int64_t numel) = 0;
    virtual void reset_env(int env) = 0;
    virtual void kv_reset(int env) = 0;
    virtual void env_state(int env, int32_t* n_embeds, int32_t* kv_len) = 0;
    virtual void encode_frames(const float* pixels, int F, int on_device) = 0;
    virtual void preprocess_frames(const uint8_t* rgb, int n, int height, int width, int on_device, float* out_dev, bool wait) = 0;
    virtual void* stream_handle() = 0;
    virtual void frame_ring(int slots, int height, int width, uint8_t** base, int64_t* stride) = 0;
    virtual void frame_ring_wait(int slot) = 0;
    virtual void turn(const svln_turn_args& a, int64_t* out, int cap, int32_t* n_out, int32_t* kv_len) = 0;
    virtual void preprocess_time(double* ms, int64_t* frames, int reset) = 0;
    virtual int device_id() const = 0;
    virtual void append_turn(int env, const int64_t* ids, int n, int frame_base, int n_memory) = 0;
    virtual void generate(int env, int max_new, const int64_t* eos, int n_eos, int64_t* out, int cap, int32_t* n_out, bool fixed) = 0;
    virtual void generate_batch(const int32_t* envs, int n_envs, int max_new, const int64_t* eos, int n_eos, int64_t* out, int cap, int32_t* n_out) = 0;
    virtual int batch_submit(int env, int max_new, const int64_t* eos, int n_eos) = 0;
    virtual int batch_step(int32_t* finished_slots, int32_t* n_finished) = 0;
    virtual void batch_result(int slot, int32_t* env, int64_t* out, int cap, int32_t* n_out) = 0;
    virtual void batch_cancel(int slot) = 0;
    virtual void set_turn_row_limit(int rows) = 0;
    virtual void set_repetition_penalty(float penalty) = 0;
    virtual void get_hidden_batch(int slot, float* out, int max_rows, int32_t* n_rows) = 0;
    virtual void get_hidden(float* out, int max_rows, int32_t* n_rows) = 0;
    virtual void set_layer_taps(int enable, int probe_layer) = 0;
    virtual void get_layer_taps(float* out) = 0;
    virtual void get_layer_probe(int which, float* out, int64_t max_elems, int32_t* n_rows, int32_t* n_cols) = 0;
    virtual void get_embeds(int env, int start, int n, float* out) = 0;
    virtual void get_feats(int start, int n, float* out) = 0;
    virtual void get_top2(float* out) = 0;
    virtual void sync() = 0;
    virtual void set_graph(int enable) = 0;
    virtual void set_decode_persistent(int enable) = 0;
    virtual void probe_decode_layer(int layer, unsigned long long* out, int max_wgs, int32_t* n_wgs) = 0;
    virtual void set_fp8_decode(int enable) = 0;
    virtual void set_fp8_gemm(int enable) = 0;
    virtual bool op_gemm_fp8(const GemmArgs& a) = 0;
    virtual void set_memory_prune(int keep) = 0;
    virtual void op_memory_prune(const void* m, int n_rows, int keep, int32_t* out_idx, float* out_score) = 0;
    virtual void probe_reset() = 0;
    virtual void probe_read(double* ms, int64_t* launches, double* bytes) = 0;
    virtual void phase_times(double* v, double* p, double* d, int reset) = 0;
    virtual void probe_read_prefill(double* ms, int64_t* n, double* rows, double* flops, double* wbytes) = 0;
    virtual void set_feature_cache(int cap) = 0;
    virtual void feature_cache_stats(int64_t* hits, int64_t* misses) = 0;
    virtual bool op_gemm(const GemmArgs& a) = 0;
    virtual void op_gemv(GemvArgs a, int32_t* host_token) = 0;
    virtual void op_gemv_batched(GemvBatchArgs a, int32_t* host_tokens) = 0;
    virtual void op_quant_fp8(const void* w, int64_t rows, int cols, void* w8, float* scale) = 0;
    virtual void op_rmsnorm(const void* x, const void* g, void* y, int rows, int n, float eps) = 0;
    virtual void op_layernorm(const void* x, const void* g, const void* b, void* y, int rows, int n, float eps) = 0;
    virtual void op_attention_llm(void* qkv, int ld, int T, int P, const void* ctx, int ctx_T, void* out, int o_stride, int nsplit) = 0;
    virtual void op_attention_vit(const void* qkv, int ld, int F, void* out, int o_stride) = 0;
    virtual void op_pool(const void* in, void* out, int F) = 0;
    virtual void op_patchify(const float* pix, void* out, int F) = 0;
};

template <typename T>
class Engine : public EngineBase {
public:
    svln_config c;
    int device;
    hipStream_t st = nullptr;
    std::vector<void*> allocs;
    std::unordered_map<std::string, Slot> slots;

    // derived dims
    int Hv, Iv, vheads, vhd, side, S, kp, oside, otok, H, I, nq, nkv, qkv_dim, V;
    int vhdp, vvrows, vtiles;         // ViT attention geometry
    int pages_per_env, pages_total;

    struct VLayer { T *ln1_w, *ln1_b, *qkv_w, *qkv_b, *out_w, *out_b, *ln2_w, *ln2_b, *fc1_w, *fc1_b, *fc2_w, *fc2_b; };
    struct Q8 { uint8_t* q = nullptr; float* s = nullptr; };      // fp8 (e4m3) copy of a weight matrix + per-row scales (opt-in decode mode)
    struct LLayer { T *in_norm, *qkv_w, *qkv_b, *o_w, *post_norm, *gu_w, *down_w, *kpool, *vpool; Q8 qkv8, o8, gu8, down8; };
    Q8 lm_head8; bool fp8_on = false, fp8_built = false;
    bool fp8_gemm_on = false; uint8_t* act8 = nullptr; float* act8_scale = nullptr;      // opt-in fp8 MFMA products: quantised activation rows
    T *patch_w, *patch_b, *pos_emb, *proj0_w, *proj0_b, *proj2_w, *proj2_b, *embed, *final_norm, *lm_head;
    std::vector<VLayer> vl;
    std::vector<LLayer> ll;

    // vision workspaces
    float* pix = nullptr;
    T *patches, *vx, *vn, *vqkv, *vattn, *vh, *vkpool, *vvpool, *proj_h, *proj_o, *feats;
    int n_feat_frames = 0;
    int* tap_idx; float* tap_w;
    // llm workspaces
    T *x, *xn, *qkv, *attn, *hbuf, *hid_tap, *head_xn;
    float* gemm_ws = nullptr; size_t gemm_ws_elems = 0; void* zero_line = nullptr;
    float* inv_freq; float* rope_tab;
    float* attn_part; size_t attn_part_elems = 0; int nsplit_max, tiles_per_split;
    float* part_val; int* part_idx; int* d_token; float* d_top2;
    GenCtl* d_ctl;               // device-side generation state (position / kv length of the next decode step, done flag, emitted count)
    GenCtl* h_ctl = nullptr;     // pinned
    int *d_eos = nullptr, *d_out_ids = nullptr, *h_out_ids = nullptr; std::vector<int> eos_cached; bool eos_valid = false;
    static constexpr int RUN_AHEAD = 4;     // decode steps enqueued per host synchronisation (a typical turn: 4 action tokens + EOS)
    // batched (multi-env lockstep) decode
    static constexpr int MAXB = 8;
    static constexpr int batched_mfma_min = 4;    // measured at B = 8: 340 vs 319 action-steps/s (gate/up 58 vs 97 us per launch)
    DecodeSlot* d_slots = nullptr; DecodeSlot* h_slots = nullptr; int* d_tok_b = nullptr; int* h_tok_b = nullptr;
    float* part_val_b = nullptr; int* part_idx_b = nullptr; T* last_rows = nullptr; int n_generated_b[MAXB] = {0};
    int* d_src; int* h_src;      // splice descriptors
    // opt-in slow-memory pruning (no reference counterpart): `<memory>` expands to the prune_keep least-average tokens
    int prune_keep = 0; float *prune_partial = nullptr, *prune_mean = nullptr, *prune_score = nullptr; int *d_sel = nullptr, *h_sel = nullptr;
    int* h_token;                // pinned
    float* h_top2;

    struct Env {
        T* embeds = nullptr; int n_embeds = 0; int kv_len = 0;
        int* d_pages = nullptr; std::vector<int> pages; int n_pages = 0;
    };
    std::vector<Env> envs;
    std::vector<int> free_pages;
    int n_generated = 0;

    // decode graph + probes.  The step graph holds the env's page-table pointer, so there is one set per env (a round-robin over several
    // envs through svln_generate replays instead of re-capturing); [0] = the whole step, [1] / [2] = the halves around the probed launch
    // (captured only while the roofline probe is on); a run-ahead batch of several steps is one graph.
    bool use_graph = false;
    struct GraphSet { std::unordered_map<int, hipGraphExec_t> ex; };     // key: steps (whole batch) | 1000 (head of a probed step) | 2000 + steps (its tail)
    std::vector<GraphSet> graphs;
    std::unordered_map<int, hipGraphExec_t> bgraphs;          // batched decode step: key = B | penalty << 8 | fp8 gemm << 9
    // per-turn truncation of the spliced rows (the reference's config.tokenizer_model_max_length, stream_video_vln.py:241-244); 0 = none
    int turn_row_limit = 0;
    // HF repetition penalty of the checkpoint's generation_config (1 = off): flags of the tokens generated in the current turn
    float rep_penalty = 1.0f; uint8_t* pen_flags = nullptr; uint8_t* pen_flags_b = nullptr; int* d_pen_rows = nullptr; int* h_pen_rows = nullptr;
    int* d_pen_ids = nullptr; int* h_pen_ids = nullptr;
    std::vector<hipEvent_t> probe_ev; size_t probe_used = 0; bool probe_on = false; double probe_bytes = 0;
    // second probe: the layer-0 gate/up product of every steady prefill (M <= 256 rows: main launch + K-split tail + reduce) between two events
    std::vector<hipEvent_t> pprobe_ev; size_t pprobe_used = 0; double pprobe_rows = 0;
    hipEvent_t ph_ev[5]; double ph_ms[3] = {0, 0, 0}; bool vision_pending = false;   // v0 v1 p0 p1 d1

    template <typename U> U* dalloc(size_t n, bool zero = false) {
        void* p = nullptr;
        HIP_CHECK(hipMalloc(&p, n * sizeof(U) + 256));
        if (zero) HIP_CHECK(hipMemsetAsync(p, 0, n * sizeof(U), st));
        allocs.push_back(p);
        return (U*)p;
    }
    void add_slot(const std::string& name, void* dst, int ld, int64_t rows, int cols, RowMap m = RowMap{1 << 30, 1, 0}) {
        slots[name] = Slot{dst, ld, rows, cols, m, false};
    }
    void add_linear(const std::string& name, T* w, T* b, int out_f, int in_f) {
        add_slot(name + ".weight", w, in_f, out_f, in_f);
        if (b) add_slot(name + ".bias", b, out_f, 1, out_f);
    }

    int device_id() const override { return device; }

    Engine(const svln_config& cfg, int dev) : c(cfg), device(dev) {
        DeviceGuard guard(dev);
        HIP_CHECK(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
        Hv = c.v_hidden; Iv = c.v_inter; vheads = c.v_heads; vhd = Hv / vheads; side = c.v_image / c.v_patch; S = side * side;
        kp = ((3 * c.v_patch * c.v_patch + 7) / 8) * 8;
        oside = (side + 1) / 2; otok = oside * oside;
        H = c.hidden; I = c.inter; nq = c.q_heads; nkv = c.kv_heads; qkv_dim = (nq + 2 * nkv) * c.head_dim; V = c.vocab;
        REQUIRE(c.head_dim == 128, "LLM head_dim must be 128");
        REQUIRE(vhd == 72, "vision head_dim must be 72");
        REQUIRE(nq % nkv == 0, "q_heads must be a multiple of kv_heads");
        REQUIRE(nq / nkv <= 32, "at most 32 q heads per kv head (decode attention keeps one GQA group in a 32-row tile)");
        REQUIRE(H % 8 == 0 && I % 32 == 0 && Hv % 8 == 0 && Iv % 8 == 0, "dims must be multiples of 8 (inter of 32)");
        REQUIRE(c.max_positions % PAGE == 0, "max_positions must be a multiple of 64");
        constexpr int EPC = Elt<T>::PER_CHUNK;
        vhdp = ((((vhd + EPC - 1) / EPC) + 1) & ~1) * EPC;
        vvrows = ((vhd + 31) / 32) * 32;
        vtiles = (S + PAGE - 1) / PAGE;

        const std::string VT = "model.vision_tower.vision_tower.vision_model.";
        patch_w = dalloc<T>((size_t)Hv * kp, true); patch_b = dalloc<T>(Hv); pos_emb = dalloc<T>((size_t)S * Hv);
        add_slot(VT + "embeddings.patch_embedding.weight", patch_w, kp, Hv, 3 * c.v_patch * c.v_patch);
        add_slot(VT + "embeddings.patch_embedding.bias", patch_b, Hv, 1, Hv);
        add_slot(VT + "embeddings.position_embedding.weight", pos_emb, Hv, S, Hv);
        vl.resize(c.v_layers);
        for (int i = 0; i < c.v_layers; ++i) {
            VLayer& L = vl[i];
            const std::string P = VT + "encoder.layers." + std::to_string(i) + ".";
            L.ln1_w = dalloc<T>(Hv); L.ln1_b = dalloc<T>(Hv); L.ln2_w = dalloc<T>(Hv); L.ln2_b = dalloc<T>(Hv);
            L.qkv_w = dalloc<T>((size_t)3 * Hv * Hv); L.qkv_b = dalloc<T>(3 * Hv);
            L.out_w = dalloc<T>((size_t)Hv * Hv); L.out_b = dalloc<T>(Hv);
            L.fc1_w = dalloc<T>((size_t)Iv * Hv); L.fc1_b = dalloc<T>(Iv);
            L.fc2_w = dalloc<T>((size_t)Hv * Iv); L.fc2_b = dalloc<T>(Hv);
            add_slot(P + "layer_norm1.weight", L.ln1_w, Hv, 1, Hv); add_slot(P + "layer_norm1.bias", L.ln1_b, Hv, 1, Hv);
            add_slot(P + "layer_norm2.weight", L.ln2_w, Hv, 1, Hv); add_slot(P + "layer_norm2.bias", L.ln2_b, Hv, 1, Hv);
            add_linear(P + "self_attn.q_proj", L.qkv_w, L.qkv_b, Hv, Hv);
            add_linear(P + "self_attn.k_proj", L.qkv_w + (size_t)Hv * Hv, L.qkv_b + Hv, Hv, Hv);
            add_linear(P + "self_attn.v_proj", L.qkv_w + (size_t)2 * Hv * Hv, L.qkv_b + 2 * Hv, Hv, Hv);
            add_linear(P + "self_attn.out_proj", L.out_w, L.out_b, Hv, Hv);
            add_linear(P + "mlp.fc1", L.fc1_w, L.fc1_b, Iv, Hv);
            add_linear(P + "mlp.fc2", L.fc2_w, L.fc2_b, Hv, Iv);
        }
        proj0_w = dalloc<T>((size_t)H * Hv); proj0_b = dalloc<T>(H); proj2_w = dalloc<T>((size_t)H * H); proj2_b = dalloc<T>(H);
        add_linear("model.mm_projector.0", proj0_w, proj0_b, H, Hv);
        add_linear("model.mm_projector.2", proj2_w, proj2_b, H, H);
        embed = dalloc<T>((size_t)V * H);
        add_slot("model.embed_tokens.weight", embed, H, V, H);

        pages_per_env = c.max_positions / PAGE;
        pages_total = pages_per_env * c.max_envs;
        ll.resize(c.layers);
        const int qd = nq * 128, kd = nkv * 128;
        for (int i = 0; i < c.layers; ++i) {
            LLayer& L = ll[i];
            const std::string P = "model.layers." + std::to_string(i) + ".";
            L.in_norm = dalloc<T>(H); L.post_norm = dalloc<T>(H);
            L.qkv_w = dalloc<T>((size_t)qkv_dim * H); L.qkv_b = dalloc<T>(qkv_dim);
            L.o_w = dalloc<T>((size_t)H * qd); L.gu_w = dalloc<T>((size_t)2 * I * H); L.down_w = dalloc<T>((size_t)H * I);
            L.kpool = dalloc<T>((size_t)pages_total * nkv * PAGE * 128, true);
            L.vpool = dalloc<T>((size_t)pages_total * nkv * 128 * PAGE, true);
            add_slot(P + "input_layernorm.weight", L.in_norm, H, 1, H);
            add_slot(P + "post_attention_layernorm.weight", L.post_norm, H, 1, H);
            add_linear(P + "self_attn.q_proj", L.qkv_w, L.qkv_b, qd, H);
            add_linear(P + "self_attn.k_proj", L.qkv_w + (size_t)qd * H, L.qkv_b + qd, kd, H);
            add_linear(P + "self_attn.v_proj", L.qkv_w + (size_t)(qd + kd) * H, L.qkv_b + qd + kd, kd, H);
            add_linear(P + "self_attn.o_proj", L.o_w, nullptr, H, qd);
            add_slot(P + "mlp.gate_proj.weight", L.gu_w, H, I, H, RowMap{32, 2, 0});
            add_slot(P + "mlp.up_proj.weight", L.gu_w, H, I, H, RowMap{32, 2, 1});
            add_linear(P + "mlp.down_proj", L.down_w, nullptr, H, I);
        }
        final_norm = dalloc<T>(H); lm_head = dalloc<T>((size_t)V * H);
        add_slot("model.norm.weight", final_norm, H, 1, H);
        add_slot("lm_head.weight", lm_head, H, V, H);

        // vision workspaces
        const size_t rv = (size_t)c.max_frames * S;
        pix = dalloc<float>((size_t)c.max_frames * 3 * c.v_image * c.v_image);
        patches = dalloc<T>(rv * kp); vx = dalloc<T>(rv * Hv); vn = dalloc<T>(rv * Hv); vqkv = dalloc<T>(rv * 3 * Hv);
        vattn = dalloc<T>(rv * Hv); vh = dalloc<T>(rv * Iv);
        vkpool = dalloc<T>((size_t)vtiles * c.max_frames * vheads * PAGE * vhdp, true);
        vvpool = dalloc<T>((size_t)vtiles * c.max_frames * vheads * vvrows * PAGE, true);
        proj_h = dalloc<T>(rv * H); proj_o = dalloc<T>(rv * H);
        feats = dalloc<T>((size_t)c.max_frames * otok * H);
        {   // bilinear taps: F.interpolate(align_corners=False), scale = side/oside  (stream_video_vln.py:64-67)
            std::vector<int> ti(2 * oside); std::vector<float> tw(2 * oside);
            const double scale = (double)side / (double)oside;
            for (int d = 0; d < oside; ++d) {
                double src = (d + 0.5) * scale - 0.5; if (src < 0) src = 0;
                int i0 = (int)std::floor(src); if (i0 > side - 1) i0 = side - 1;
                int i1 = i0 + 1 < side ? i0 + 1 : side - 1;
                float l1 = (float)(src - i0);
                ti[2 * d] = i0; ti[2 * d + 1] = i1; tw[2 * d] = 1.0f - l1; tw[2 * d + 1] = l1;
            }
            tap_idx = dalloc<int>(2 * oside); tap_w = dalloc<float>(2 * oside);
            HIP_CHECK(hipMemcpy(tap_idx, ti.data(), ti.size() * sizeof(int), hipMemcpyHostToDevice));
            HIP_CHECK(hipMemcpy(tap_w, tw.data(), tw.size() * sizeof(float), hipMemcpyHostToDevice));
        }
        // llm workspaces
        const size_t rt = (size_t)c.max_positions;
        x = dalloc<T>(rt * H); xn = dalloc<T>(rt * H); qkv = dalloc<T>(rt * qkv_dim); attn = dalloc<T>(rt * qd); hbuf = dalloc<T>(rt * I);
        hid_tap = dalloc<T>((size_t)HID_TAP_ROWS * H);
        head_xn = dalloc<T>((size_t)H);
        {   // split-K slabs: enough for 8 splits of the widest skinny product (rows of one ViT frame batch or a 512-row prefill)
            size_t widest = (size_t)(qkv_dim > 3 * Hv ? qkv_dim : 3 * Hv);
            if ((size_t)Iv > widest) widest = Iv;
            size_t rows = (size_t)c.max_frames * S > 768 ? (size_t)c.max_frames * S : 768;
            gemm_ws_elems = 8 * rows * widest;
            if (gemm_ws_elems > (size_t)64 << 20) gemm_ws_elems = (size_t)64 << 20;
            gemm_ws = dalloc<float>(gemm_ws_elems);
            zero_line = dalloc<char>(256, true);
        }
        {
            std::vector<float> f(64);
            for (int j = 0; j < 64; ++j) f[j] = 1.0f / powf(c.rope_theta, (float)(2 * j) / 128.0f);     // modeling_qwen2.py:115
            inv_freq = dalloc<float>(64);
            HIP_CHECK(hipMemcpy(inv_freq, f.data(), 64 * sizeof(float), hipMemcpyHostToDevice));
            rope_tab = dalloc<float>((size_t)c.max_positions * 128);
            launch_rope_table(st, rope_tab, inv_freq, c.max_positions);
        }
        tiles_per_split = 1;                       // decode: one 64-key page per workgroup, <= 64 splits
        while ((pages_per_env + tiles_per_split - 1) / tiles_per_split > 64) ++tiles_per_split;
        nsplit_max = (pages_per_env + tiles_per_split - 1) / tiles_per_split;
        {   // partials: decode [nsplit_max][nkv][32][132], prefill split-KV [<= 8][nkv][PREFILL_SPLIT_ROWS][132]
            constexpr int PW = 128 + ATTN_PART_PAD;
            const size_t dec = (size_t)MAXB * nsplit_max * nkv * 32 * PW, pre = (size_t)8 * nkv * PREFILL_SPLIT_ROWS * PW;
            attn_part_elems = dec > pre ? dec : pre;
            attn_part = dalloc<float>(attn_part_elems);
        }
        part_val = dalloc<float>(2048); part_idx = dalloc<int>(2048);
        d_token = dalloc<int>(4, true); d_top2 = dalloc<float>(4, true);
        d_ctl = (GenCtl*)dalloc<int>(sizeof(GenCtl) / sizeof(int), true);
        d_eos = dalloc<int>((size_t)(V > 16 ? V : 16)); d_out_ids = dalloc<int>((size_t)c.max_positions + 8);
        HIP_CHECK(hipHostMalloc((void**)&h_ctl, sizeof(GenCtl)));
        HIP_CHECK(hipHostMalloc((void**)&h_out_ids, ((size_t)c.max_positions + 8) * sizeof(int)));
        d_slots = dalloc<DecodeSlot>(MAXB, true); d_tok_b = dalloc<int>(MAXB, true);
        part_val_b = dalloc<float>((size_t)MAXB * 2048); part_idx_b = dalloc<int>((size_t)MAXB * 2048);
        last_rows = dalloc<T>((size_t)MAXB * H);
        HIP_CHECK(hipHostMalloc((void**)&h_slots, MAXB * sizeof(DecodeSlot)));
        HIP_CHECK(hipHostMalloc((void**)&h_tok_b, MAXB * sizeof(int)));
        d_src = dalloc<int>(rt);
        HIP_CHECK(hipHostMalloc((void**)&h_src, rt * sizeof(int)));
        HIP_CHECK(hipHostMalloc((void**)&h_token, 16));
        HIP_CHECK(hipHostMalloc((void**)&h_top2, 16));
        envs.resize(c.max_envs);
        graphs.resize(c.max_envs);
        for (auto& e : envs) {
            e.embeds = dalloc<T>(rt * H);
            e.d_pages = dalloc<int>(pages_per_env, true);
            e.pages.assign(pages_per_env, 0);
        }
        for (int p = pages_total - 1; p >= 0; --p) free_pages.push_back(p);
        for (int i = 0; i < 5; ++i) HIP_CHECK(hipEventCreate(&ph_ev[i]));
        HIP_CHECK(init_kernel_attributes());
        HIP_CHECK(hipStreamSynchronize(st));
    }

    ~Engine() override {
        int prev = -1;
        (void)hipGetDevice(&prev);
        (void)hipSetDevice(device);
        (void)hipStreamSynchronize(st);
        drop_graphs();
        if (d_rgb) (void)hipFree(d_rgb);
        for (auto& t : rs_tabs) for (void* p : t.bufs) (void)hipFree(p);
        for (int i = 0; i < 2; ++i) {
            if (h_rgb[i]) (void)hipHostFree(h_rgb[i]);
            if (h2d_ev[i]) (void)hipEventDestroy(h2d_ev[i]);
        }
        for (auto& pr : pp_ring) { if (pr.a) (void)hipEventDestroy(pr.a); if (pr.b) (void)hipEventDestroy(pr.b); }
        if (src_ev) (void)hipEventDestroy(src_ev);
        for (auto ev : ring_ev) (void)hipEventDestroy(ev);
        if (ring_host) (void)hipHostFree(ring_host);
        for (auto e : probe_ev) (void)hipEventDestroy(e);
        for (auto e : pprobe_ev) (void)hipEventDestroy(e);
        for (int i = 0; i < 5; ++i) (void)hipEventDestroy(ph_ev[i]);
        (void)hipHostFree(h_ctl); (void)hipHostFree(h_out_ids);
        if (h_giveup) (void)hipHostFree(h_giveup);
        if (h_hash) (void)hipHostFree(h_hash);
        for (void* p : allocs) (void)hipFree(p);
        (void)hipHostFree(h_src); (void)hipHostFree(h_token); (void)hipHostFree(h_top2);
        if (h_sel) (void)hipHostFree(h_sel);
        (void)hipHostFree(h_slots); (void)hipHostFree(h_tok_b);
        if (h_pen_rows) (void)hipHostFree(h_pen_rows);
        if (h_pen_ids) (void)hipHostFree(h_pen_ids);
        (void)hipStreamDestroy(st);
        if (prev >= 0 && prev != device) (void)hipSetDevice(prev);
    }

    // ------------------------------------------------------------------------------- weights
    Slot& slot(const char* name) {
        auto it = slots.find(name);
        REQUIRE(it != slots.end(), std::string("unknown tensor: ") + name);
        return it->second;
    }
    void synth_tensor(const char* name, uint64_t seed_t, float hw, float base) override {
        Slot& s = slot(name);
        launch_synth<T>(st, s.dst, s.ld, s.rows, s.cols, s.map, seed_t, hw, base);
        s.filled = true;
    }
    void set_tensor(const char* name, const void* data, int dtype, int64_t numel, int on_device) override {
        Slot& s = slot(name);
        REQUIRE(numel == s.rows * s.cols, std::string("size mismatch for ") + name);
        const size_t bytes = (size_t)numel * (dtype == SVLN_F32 ? 4 : 2);
        const void* src = data;
        void* tmp = nullptr;
        if (!on_device) {
            HIP_CHECK(hipMalloc(&tmp, bytes));
            HIP_CHECK(hipMemcpyAsync(tmp, data, bytes, hipMemcpyHostToDevice, st));
            src = tmp;
        }
        launch_convert<T>(st, s.dst, s.ld, s.rows, s.cols, s.map, src, dtype == SVLN_F32);
        HIP_CHECK(hipStreamSynchronize(st));
        if (tmp) HIP_CHECK(hipFree(tmp));
        s.filled = true;
    }
    int weights_missing() override {
        int n = 0;
        for (auto& kv : slots) if (!kv.second.filled) { if (!n) g_err = "missing tensor: " + kv.first; ++n; }
        return n;
    }
    void get_tensor_f32(const char* name, float* out, int64_t numel) override {
        Slot& s = slot(name);
        REQUIRE(numel == s.rows * s.cols, "size mismatch");
        std::vector<T> row(s.cols);
        for (int64_t r = 0; r < s.rows; ++r) {
            const int64_t dr = (r / s.map.blk) * (int64_t)s.map.blk * s.map.nint + (int64_t)s.map.phase * s.map.blk + r % s.map.blk;
            HIP_CHECK(hipMemcpy(row.data(), (const T*)s.dst + dr * s.ld, s.cols * sizeof(T), hipMemcpyDeviceToHost));
            for (int k = 0; k < s.cols; ++k) out[r * s.cols + k] = host_to_f32(row[k]);
        }
    }
    static float host_to_f32(float v) { return v; }
    static float host_to_f32(bf16 v) {
        unsigned short b; std::memcpy(&b, &v, 2);
        uint32_t u = (uint32_t)b << 16; float f; std::memcpy(&f, &u, 4); return f;
    }

    // ------------------------------------------------------------------------------- session
    Env& env_at(int env) { REQUIRE(env >= 0 && env < (int)envs.size(), "env out of range"); return envs[env]; }
    void release_pages(Env& e) {
        for (int i = 0; i < e.n_pages; ++i) free_pages.push_back(e.pages[i]);
        e.n_pages = 0;
    }
    // A reset drops the env's turn in the multi-env scheduler, if it has one (submitted, running or finished but not collected): its rows
    // and KV pages are gone, so the job must not reach another batch_step.
    void reset_env(int env) override { Env& e = env_at(env); cancel_jobs_of(env); release_pages(e); e.n_embeds = 0; e.kv_len = 0; }
    void kv_reset(int env) override { Env& e = env_at(env); cancel_jobs_of(env); release_pages(e); e.kv_len = 0; }
    void env_state(int env, int32_t* n_embeds, int32_t* kv_len) override { Env& e = env_at(env); *n_embeds = e.n_embeds; *kv_len = e.kv_len; }
    void ensure_pages(Env& e, int positions) {      // pages covering [0, positions)
        const int need = (positions + PAGE - 1) / PAGE;
        REQUIRE(need <= pages_per_env, "sequence exceeds max_positions");
        if (need <= e.n_pages) return;
        const int first = e.n_pages;
        while (e.n_pages < need) {
            REQUIRE(!free_pages.empty(), "KV page pool exhausted");
            e.pages[e.n_pages++] = free_pages.back();
            free_pages.pop_back();
        }
        HIP_CHECK(hipMemcpyAsync(e.d_pages + first, e.pages.data() + first, (e.n_pages - first) * sizeof(int), hipMemcpyHostToDevice, st));
    }

    // ------------------------------------------------------------------------------- vision
    GemmArgs gemm_args(const void* A, int lda, const void* W, int ldw, void* C, int ldc, const void* bias, const void* res, int ldr,
                       int res_mod, int M, int N, int K, int epi) {
        GemmArgs a; a.A = A; a.lda = lda; a.W = W; a.ldw = ldw; a.C = C; a.ldc = ldc; a.bias = bias; a.res = res; a.ldr = ldr;
        a.res_mod = res_mod; a.M = M; a.N = N; a.K = K; a.epi = epi; a.ws = gemm_ws; a.ws_elems = gemm_ws_elems; a.nsplit = 1;
        a.zeros = zero_line; a.force_cfg = 0; a.force_split = 0; a.norm_w = nullptr; a.norm_out = nullptr; a.norm_eps = 0.0f; a.norm_b = nullptr;
        a.a_scale = nullptr; a.w_scale = nullptr; a.rope = nullptr; a.vitpack = nullptr; return a;
    }
    AttnArgs vit_attn_args(const void* q, int ld, int F, void* out, int o_stride) {
        AttnArgs a; std::memset(&a, 0, sizeof(a));
        a.Q = q; a.q_stride = ld; a.O = out; a.o_stride = o_stride; a.Kpool = vkpool; a.Vpool = vvpool; a.page_table = nullptr;
        a.n_kv_total = F * vheads; a.hpf = vheads; a.G = 1; a.T = S; a.P = 0; a.kv_len = S; a.dyn_kv_len = nullptr;
        a.scale = 1.0f / sqrtf((float)vhd); a.causal = 0; a.nsplit = 1; a.tiles_per_split = vtiles; a.part = attn_part; a.rows_pad = 0;
        // one frame = 6 row blocks x 16 heads = 96 workgroups of the plain kernel: too few to fill the chip
        const int wgs = ((S + 127) / 128) * F * vheads, rows_pad = ((S + 127) / 128) * 128;
        // (measured: 6 splits are slower than 3 -- the fp32 partials double -- and unsplit is 35 us against 18 + 8 for split + combine)
        // one frame, head_dim 72: the key split stays inside the workgroup (key groups merged through LDS): 14.3 us against 15.6 + 7.4
        if (wgs < 192 && vtiles >= 6 && vhd == 72) { a.key_groups = attn_key_groups<T>(72); return a; }
        // (other head dims)
        if (wgs < 192 && vtiles >= 6 && (size_t)3 * F * vheads * rows_pad * (vhd + ATTN_PART_PAD) <= attn_part_elems) {
            a.nsplit = 3; a.tiles_per_split = (vtiles + 2) / 3; a.rows_pad = rows_pad;
        }
        return a;
    }
    void vit_attention(const void* qkv_buf, int ld, int F, void* out, int o_stride, bool packed = false) {
        if (!packed) launch_vit_kv_pack<T>(st, qkv_buf, ld, vkpool, vvpool, F, S, vheads, vhd);
        AttnArgs a = vit_attn_args(qkv_buf, ld, F, out, o_stride);
        launch_attention<T>(st, a, vhd, 4);
        if (a.nsplit > 1) launch_attention_combine<T>(st, a, vhd);
    }
    // SigLIP tower + projector + pool on F frames of `pixbuf` (fp32 [F,3,S,S]) -> dst [F*otok][H]
    void run_vit(const float* pixbuf, int F, T* dst) {
        const int M = F * S;
        // SigLipVisionEmbeddings (siglip_encoder.py:169-174): patch GEMM + bias + position embedding
        launch_patchify<T>(st, pixbuf, patches, F, c.v_image, c.v_patch, kp);
        launch_gemm<T>(st, gemm_args(patches, kp, patch_w, kp, vx, Hv, patch_b, pos_emb, Hv, S, M, Hv, kp, EPI_NONE));
        bool vn_ready = false;                      // vn already holds ln1(vx) (written by the previous layer's fc2 epilogue)
        for (int i = 0; i < c.v_layers; ++i) {      // SigLipEncoderLayer (siglip_encoder.py:269-305)
            const VLayer& L = vl[i];
            if (!vn_ready) launch_layernorm<T>(st, vx, L.ln1_w, L.ln1_b, vn, M, Hv, c.v_eps);
            // (one frame: the product runs split-K and its slab reduce also packs the K / V^T pages of the attention)
            GemmArgs aq = gemm_args(vn, Hv, L.qkv_w, Hv, vqkv, 3 * Hv, L.qkv_b, nullptr, 0, 0, M, 3 * Hv, Hv, EPI_NONE);
            const VitPackArgs vpk{vkpool, vvpool, F, S, vheads, vhd};
            aq.vitpack = &vpk;
            const bool packed = launch_gemm<T>(st, aq);
            vit_attention(vqkv, 3 * Hv, F, vattn, Hv, packed);
            // out_proj / fc2 run split-K at one frame: their slab reduce also emits the following LayerNorm
            GemmArgs ao = gemm_args(vattn, Hv, L.out_w, Hv, vx, Hv, L.out_b, vx, Hv, 0, M, Hv, Hv, EPI_NONE);
            ao.norm_w = L.ln2_w; ao.norm_b = L.ln2_b; ao.norm_out = vn; ao.norm_eps = c.v_eps;
            if (!launch_gemm<T>(st, ao)) launch_layernorm<T>(st, vx, L.ln2_w, L.ln2_b, vn, M, Hv, c.v_eps);
            launch_gemm<T>(st, gemm_args(vn, Hv, L.fc1_w, Hv, vh, Iv, L.fc1_b, nullptr, 0, 0, M, Iv, Hv, EPI_GELU_TANH));
            GemmArgs a2 = gemm_args(vh, Iv, L.fc2_w, Iv, vx, Hv, L.fc2_b, vx, Hv, 0, M, Hv, Iv, EPI_NONE);
            if (i + 1 < c.v_layers) { a2.norm_w = vl[i + 1].ln1_w; a2.norm_b = vl[i + 1].ln1_b; a2.norm_out = vn; a2.norm_eps = c.v_eps; }
            vn_ready = launch_gemm<T>(st, a2);
        }
        // mm_projector (builder.py:41-48) then get_2dPool bilinear 27x27 -> 14x14 (stream_video_vln.py:53-73)
        launch_gemm<T>(st, gemm_args(vx, Hv, proj0_w, Hv, proj_h, H, proj0_b, nullptr, 0, 0, M, H, Hv, EPI_GELU_ERF));
        launch_gemm<T>(st, gemm_args(proj_h, H, proj2_w, H, proj_o, H, proj2_b, nullptr, 0, 0, M, H, H, EPI_NONE));
        launch_pool<T>(st, proj_o, dst, tap_idx, tap_w, F, side, oside, H);
    }

    // Optional memoisation of pooled frame features: the reference re-encodes the
    // num_history memory frames through the ViT at every window restart (stream_video_vln.py:104); the same pixels through
    // the same frozen weights give the same features, so frames are keyed by a 128-bit content hash and reused.  OFF by
    // default (svln_set_feature_cache).
    struct CacheEntry { unsigned long long a, b; uint64_t stamp; bool used; };
    int fc_cap = 0; T* fc_store = nullptr; std::vector<CacheEntry> fc_entries; uint64_t fc_clock = 0;
    int64_t fc_hits = 0, fc_misses = 0;
    float* pix2 = nullptr; T* feats_stage = nullptr; unsigned long long* d_hash = nullptr; unsigned long long* h_hash = nullptr;
    void set_feature_cache(int cap) override {
        REQUIRE(cap >= 0, "capacity must be >= 0");
        if (cap > 0 && cap != (int)fc_entries.size()) {
            REQUIRE(fc_store == nullptr, "feature cache capacity can be set once");
            fc_store = dalloc<T>((size_t)cap * otok * H);
            fc_entries.assign(cap, CacheEntry{0, 0, 0, false});
            pix2 = dalloc<float>((size_t)c.max_frames * 3 * c.v_image * c.v_image);
            feats_stage = dalloc<T>((size_t)c.max_frames * otok * H);
            d_hash = dalloc<unsigned long long>(2 * c.max_frames);
            HIP_CHECK(hipHostMalloc((void**)&h_hash, 2 * c.max_frames * sizeof(unsigned long long)));
        }
        fc_cap = cap;
        for (auto& e : fc_entries) e.used = false;
    }
    void feature_cache_stats(int64_t* hits, int64_t* misses) override { *hits = fc_hits; *misses = fc_misses; }

    void encode_frames(const float* pixels, int F, int on_device) override {
        REQUIRE(F >= 1 && F <= c.max_frames, "n_frames out of range");
        REQUIRE(weights_missing() == 0, g_err);
        const size_t fwords = (size_t)3 * c.v_image * c.v_image;
        HIP_CHECK(hipEventRecord(ph_ev[0], st));
        HIP_CHECK(hipMemcpyAsync(pix, pixels, F * fwords * sizeof(float), on_device ? hipMemcpyDeviceToDevice : hipMemcpyHostToDevice, st));
        if (fc_cap <= 0) {
            run_vit(pix, F, feats);
        } else {
            HIP_CHECK(hipMemsetAsync(d_hash, 0, 2 * F * sizeof(unsigned long long), st));
            launch_frame_hash(st, pix, F, fwords, d_hash);
            HIP_CHECK(hipMemcpyAsync(h_hash, d_hash, 2 * F * sizeof(unsigned long long), hipMemcpyDeviceToHost, st));
            HIP_CHECK(hipStreamSynchronize(st));
            std::vector<int> slot(F, -1), miss;
            for (int f = 0; f < F; ++f) {
                for (int k = 0; k < fc_cap; ++k)
                    if (fc_entries[k].used && fc_entries[k].a == h_hash[2 * f] && fc_entries[k].b == h_hash[2 * f + 1]) { slot[f] = k; break; }
                if (slot[f] >= 0) { fc_entries[slot[f]].stamp = ++fc_clock; ++fc_hits; }
                else { miss.push_back(f); ++fc_misses; }
            }
            const size_t fbytes = (size_t)otok * H * sizeof(T);
            if (!miss.empty()) {
                for (size_t j = 0; j < miss.size(); ++j)
                    HIP_CHECK(hipMemcpyAsync(pix2 + j * fwords, pix + (size_t)miss[j] * fwords, fwords * sizeof(float), hipMemcpyDeviceToDevice, st));
                run_vit(pix2, (int)miss.size(), feats_stage);
                for (size_t j = 0; j < miss.size(); ++j) {
                    const int f = miss[j];
                    int victim = -1;                                           // free slot, else least recently used
                    for (int k = 0; k < fc_cap; ++k) {
                        bool taken = false;
                        for (int g = 0; g < F; ++g) taken |= slot[g] == k;     // never evict a slot this call still needs
                        if (taken) continue;
                        if (!fc_entries[k].used) { victim = k; break; }
                        if (victim < 0 || fc_entries[k].stamp < fc_entries[victim].stamp) victim = k;
                    }
                    HIP_CHECK(hipMemcpyAsync((char*)feats + (size_t)f * fbytes, (char*)feats_stage + j * fbytes, fbytes, hipMemcpyDeviceToDevice, st));
                    if (victim >= 0) {
                        HIP_CHECK(hipMemcpyAsync((char*)fc_store + (size_t)victim * fbytes, (char*)feats_stage + j * fbytes, fbytes, hipMemcpyDeviceToDevice, st));
                        fc_entries[victim] = CacheEntry{h_hash[2 * f], h_hash[2 * f + 1], ++fc_clock, true};
                        slot[f] = victim;
                    }
                }
            }
            for (int f = 0; f < F; ++f) {
                bool was_miss = false;
                for (int g : miss) was_miss |= g == f;
                if (!was_miss)
                    HIP_CHECK(hipMemcpyAsync((char*)feats + (size_t)f * fbytes, (char*)fc_store + (size_t)slot[f] * fbytes, fbytes, hipMemcpyDeviceToDevice, st));
            }
        }
        HIP_CHECK(hipEventRecord(ph_ev[1], st));
        vision_pending = true;
        n_feat_frames = F;
    }

    // ------------------------------------------------------------------------------- a-1: image preprocess on the GPU
    // SigLipImageProcessor.preprocess (siglip_encoder.py:47-67), bit-exact with Pillow's bicubic (preprocess.hip).  Coefficient tables
    // are built on the host per frame geometry (double precision, Pillow's operation order) and cached on the device.
    struct ResampleTabs { int H, W; ResampleDev dev; uint64_t stamp = 0; std::vector<void*> bufs; };
    std::vector<ResampleTabs> rs_tabs;
    uint8_t* d_rgb = nullptr; size_t d_rgb_cap = 0; float* d_lut = nullptr;
    // pinned staging, double-buffered: call i + 1 copies its frame in while the DMA of call i may still be reading the other buffer
    uint8_t* h_rgb[2] = {nullptr, nullptr}; uint8_t* h_rgb_dev[2] = {nullptr, nullptr}; size_t h_rgb_cap[2] = {0, 0}; hipEvent_t h2d_ev[2] = {nullptr, nullptr}; int h_cur = 0;
    // GPU time of the preprocess calls: a fixed ring of (begin, end) event pairs created with the first call and read back lazily (when the
    // ring wraps, or when the totals are asked for), so that an enqueue-only call neither waits nor creates events -- event creation on
    // the hot path stalls for milliseconds whenever the runtime grows its signal pool
    static constexpr int PP_RING = 16;
    struct EvPair { hipEvent_t a = nullptr, b = nullptr; int frames = 0; };
    EvPair pp_ring[PP_RING]; size_t pp_head = 0, pp_tail = 0;      // pairs [pp_tail, pp_head) are pending
    double pp_ms = 0; int64_t pp_frames = 0;
    void pp_resolve_oldest() {
        EvPair& pr = pp_ring[pp_tail++ % PP_RING];
        float t = 0.f;
        HIP_CHECK(hipEventSynchronize(pr.b));
        HIP_CHECK(hipEventElapsedTime(&t, pr.a, pr.b));
        pp_ms += t; pp_frames += pr.frames;
    }
    void pp_collect() { while (pp_tail < pp_head) pp_resolve_oldest(); }
    void* stream_handle() override { return (void*)st; }
    // Coefficient tables per frame geometry: at most RS_CACHE geometries stay on the device (least recently used one evicted, its buffers
    // reused by hipFree after a stream sync), and a geometry the kernel cannot take is rejected BEFORE anything is uploaded.
    static constexpr int RS_CACHE = 4;
    uint64_t rs_clock = 0;
    const ResampleDev& resample_tabs(int Hh, int Ww) {
        for (auto& t : rs_tabs) if (t.H == Hh && t.W == Ww) { t.stamp = ++rs_clock; return t.dev; }
        const int Sx = c.v_image;
        ResampleAxis ah, av;
        build_resample_table(Ww, Sx, ah);
        build_resample_table(Hh, Sx, av);
        REQUIRE(av.ksize <= 40, "frame too large for the GPU preprocess kernel (more than 40 source rows per output row)");
        REQUIRE(preprocess_lds_bytes(Ww, Sx, av.ksize, ah.ksize) <= (size_t)160 * 1024,
                "frame too large for the GPU preprocess kernel (its source rows of one output row must fit the 160 KiB LDS)");
        if ((int)rs_tabs.size() >= RS_CACHE) {
            HIP_CHECK(hipStreamSynchronize(st));       // no launch may still be reading the evicted tables
            size_t v = 0;
            for (size_t k = 1; k < rs_tabs.size(); ++k) if (rs_tabs[k].stamp < rs_tabs[v].stamp) v = k;
            for (void* p : rs_tabs[v].bufs) (void)hipFree(p);
            rs_tabs.erase(rs_tabs.begin() + v);
        }
        ResampleTabs t; t.H = Hh; t.W = Ww; t.stamp = ++rs_clock;
        auto up = [&](const std::vector<int>& v) {
            int* d = nullptr;
            HIP_CHECK(hipMalloc((void**)&d, v.size() * sizeof(int) + 256));
            t.bufs.push_back(d);
            HIP_CHECK(hipMemcpyAsync(d, v.data(), v.size() * sizeof(int), hipMemcpyHostToDevice, st));
            return d;
        };
        t.dev.hmin = up(ah.xmin); t.dev.hcnt = up(ah.cnt); t.dev.hk = up(ah.k); t.dev.ks_h = ah.ksize;
        t.dev.vmin = up(av.xmin); t.dev.vcnt = up(av.cnt); t.dev.vk = up(av.k); t.dev.ks_v = av.ksize;
        HIP_CHECK(hipStreamSynchronize(st));          // the host vectors go out of scope
        rs_tabs.push_back(t);
        return rs_tabs.back().dev;
    }
    // Engine-owned pinned frame ring (svln_frame_ring): `slots` frames of height x width x 3 bytes in pinned, device-mapped host memory.
    // A frame handed to svln_preprocess_frames* that lies INSIDE the ring is read by the upload kernel where it is -- no staging copy on
    // the host (921 KB per 640x480 frame).  The camera / simulator side writes frames into the slots; a slot may be rewritten once the
    // upload that last read it has run (svln_frame_ring_wait).
    uint8_t *ring_host = nullptr, *ring_dev = nullptr; size_t ring_stride = 0, ring_frame_bytes = 0; int ring_slots = 0;
    std::vector<hipEvent_t> ring_ev; std::vector<char> ring_pending;
    void frame_ring(int slots, int height, int width, uint8_t** base, int64_t* stride) override {
        REQUIRE(slots >= 1 && height >= 1 && width >= 1, "frame ring: bad geometry");
        REQUIRE(base && stride, "null output pointer");
        HIP_CHECK(hipStreamSynchronize(st));
        for (auto ev : ring_ev) (void)hipEventDestroy(ev);
        ring_ev.clear(); ring_pending.clear();
        if (ring_host) { (void)hipHostFree(ring_host); ring_host = nullptr; ring_dev = nullptr; }
        ring_frame_bytes = (size_t)height * width * 3;
        ring_stride = (ring_frame_bytes + 255) / 256 * 256;            // slots start 256-byte aligned (the upload kernel moves 16-byte granules)
        ring_slots = slots;
        HIP_CHECK(hipHostMalloc((void**)&ring_host, ring_stride * slots + 256));
        HIP_CHECK(hipHostGetDevicePointer((void**)&ring_dev, ring_host, 0));
        ring_ev.resize(slots); ring_pending.assign(slots, 0);
        for (auto& ev : ring_ev) HIP_CHECK(hipEventCreateWithFlags(&ev, hipEventDisableTiming));
        *base = ring_host; *stride = (int64_t)ring_stride;
    }
    void frame_ring_wait(int slot) override {
        REQUIRE(slot >= 0 && slot < ring_slots, "frame ring: no such slot");
        if (ring_pending[slot]) { HIP_CHECK(hipEventSynchronize(ring_ev[slot])); ring_pending[slot] = 0; }
    }
    // device-visible address of a host frame batch that lies inside the ring (16-byte aligned, contiguous), else null
    const uint8_t* ring_lookup(const uint8_t* rgb, size_t bytes, int n) const {
        if (!ring_host || rgb < ring_host || rgb + bytes > ring_host + ring_stride * ring_slots) return nullptr;
        const size_t off = (size_t)(rgb - ring_host);
        if (off % 16 != 0) return nullptr;
        if (n > 1 && ring_stride != ring_frame_bytes) return nullptr;      // padded slots: a multi-frame batch is not contiguous
        return ring_dev + off;
    }
    void ring_mark(const uint8_t* rgb, size_t bytes) {                      // the slots this upload reads: busy until the event fires
        const int s0 = (int)((size_t)(rgb - ring_host) / ring_stride), s1 = (int)((size_t)(rgb - ring_host + bytes - 1) / ring_stride);
        for (int k = s0; k <= s1 && k < ring_slots; ++k) { HIP_CHECK(hipEventRecord(ring_ev[k], st)); ring_pending[k] = 1; }
    }
    // One model turn behind ONE call (SURVEY.md 8b `svln_turn`): encode_rgbd + the KV / embeds bookkeeping of StreamVLNForCausalLM.generate +
    // splice + greedy generation -- svln_encode_frames, svln_kv_reset / svln_reset_env, svln_append_turn, svln_generate in that order.
    void turn(const svln_turn_args& a, int64_t* out, int cap, int32_t* n_out, int32_t* kv_len) override {
        Env& e = env_at(a.env);
        REQUIRE(a.ids && a.n_ids >= 1 && out && n_out, "svln_turn: null / empty argument");
        encode_frames(a.pixels, a.n_frames, a.pixels_on_device);
        if (a.new_window) kv_reset(a.env);                      // past_key_values = None (streamvln_eval.py:349)
        if (a.new_episode && e.n_embeds != 0) reset_env(a.env); // curr_t == 0: the env's inputs_embeds start over (stream_video_vln.py:396-401)
        append_turn(a.env, a.ids, a.n_ids, 0, a.n_memory);
        generate(a.env, a.max_new_tokens, a.eos_ids, a.n_eos, out, cap, n_out, false);
        if (kv_len) *kv_len = e.kv_len;
    }
    void preprocess_frames(const uint8_t* rgb, int n, int Hh, int Ww, int on_device, float* out_dev, bool wait) override {
        REQUIRE(rgb && out_dev, "null frame / output pointer");
        REQUIRE(n >= 1 && Hh >= 1 && Ww >= 1, "bad frame geometry");
        REQUIRE((long long)Hh <= 100ll * Ww, "frames taller than 100 x their width are not supported (Pillow >= 12 resizes them vertical-first)");
        const int Sx = c.v_image;
        const size_t bytes = (size_t)n * Hh * Ww * 3;
        if (!d_lut) {
            float lut[256];
            build_normalize_lut(lut, 0.5f, 0.5f);      // image_mean = image_std = 0.5 (siglip_encoder.py:35)
            d_lut = dalloc<float>(256);
            HIP_CHECK(hipMemcpy(d_lut, lut, sizeof(lut), hipMemcpyHostToDevice));
            for (int i = 0; i < 2; ++i) HIP_CHECK(hipEventCreate(&h2d_ev[i]));
            for (auto& pr : pp_ring) { HIP_CHECK(hipEventCreate(&pr.a)); HIP_CHECK(hipEventCreate(&pr.b)); }
        }
        const ResampleDev tabs = resample_tabs(Hh, Ww);       // (by value: a later geometry may evict the cache entry)
        if (d_rgb_cap < bytes + 64) {
            HIP_CHECK(hipStreamSynchronize(st));
            if (d_rgb) HIP_CHECK(hipFree(d_rgb));
            d_rgb = nullptr; d_rgb_cap = 0;
            HIP_CHECK(hipMalloc((void**)&d_rgb, bytes + 256));
            d_rgb_cap = bytes + 256;
        }
        if (pp_head - pp_tail == PP_RING) pp_resolve_oldest();      // 16 calls old: finished long ago
        EvPair& pr = pp_ring[pp_head++ % PP_RING];
        pr.frames = n;
        HIP_CHECK(hipEventRecord(pr.a, st));
        if (on_device) {
            HIP_CHECK(hipMemcpyAsync(d_rgb, rgb, bytes, hipMemcpyDeviceToDevice, st));
        } else if (const uint8_t* rd = ring_lookup(rgb, bytes, n)) {
            launch_upload(st, rd, d_rgb, bytes);             // the frame sits in the engine's pinned ring: read it where it is
            ring_mark(rgb, bytes);
        } else {
            const int b = h_cur; h_cur ^= 1;
            if (h_rgb_cap[b] < bytes) {
                HIP_CHECK(hipStreamSynchronize(st));
                if (h_rgb[b]) HIP_CHECK(hipHostFree(h_rgb[b]));
                h_rgb[b] = nullptr; h_rgb_cap[b] = 0;
                HIP_CHECK(hipHostMalloc((void**)&h_rgb[b], bytes + 16));
                HIP_CHECK(hipHostGetDevicePointer((void**)&h_rgb_dev[b], h_rgb[b], 0));
                h_rgb_cap[b] = bytes;
            }
            HIP_CHECK(hipEventSynchronize(h2d_ev[b]));   // the DMA that last read this staging buffer (two calls ago) has finished
            std::memcpy(h_rgb[b], rgb, bytes);           // pinned, device-mapped staging: the upload below is one kernel reading it over PCIe
            launch_upload(st, h_rgb_dev[b], d_rgb, bytes);
            HIP_CHECK(hipEventRecord(h2d_ev[b], st));
        }
        launch_preprocess(st, d_rgb, out_dev, n, Hh, Ww, Sx, tabs, d_lut);
        LAUNCH_CHECK("preprocess");
        HIP_CHECK(hipEventRecord(pr.b, st));
        // wait: out_dev (the caller's tensor) is complete on return.  Otherwise the work is only enqueued on the engine's stream (the
        // frame bytes have been consumed): svln_encode_frames on the same engine is ordered behind it, any other stream must wait on
        // svln_engine_stream() itself.
        if (wait) HIP_CHECK(hipEventSynchronize(pr.b));
    }
    void preprocess_time(double* ms, int64_t* frames, int reset) override {
        pp_collect();
        *ms = pp_ms; *frames = pp_frames;
        if (reset) { pp_ms = 0; pp_frames = 0; }
    }

    // ------------------------------------------------------------------------------- splice
    hipEvent_t src_ev = nullptr; bool src_pending = false;      // h_src (pinned splice descriptor) is still being read by the last upload
    void append_turn(int env, const int64_t* ids, int n, int frame_base, int n_memory) override {
        Env& e = env_at(env);
        if (src_pending) { HIP_CHECK(hipEventSynchronize(src_ev)); src_pending = false; }
        REQUIRE(frame_base >= 0 && n_memory >= 0 && frame_base + n_memory <= n_feat_frames, "n_memory exceeds encoded frames");
        // rows of this turn, then the reference's per-turn truncation (new_input_embeds[:tokenizer_model_max_length], stream_video_vln.py:241-244),
        // then the capacity check of the accumulated sequence (max_positions is this engine's buffer size; the reference has no such cap)
        int img = frame_base + n_memory, mem_used = 0;
        const int cap = c.max_positions - e.n_embeds;
        const size_t lim = turn_row_limit > 0 ? (size_t)turn_row_limit : (size_t)1 << 30;
        std::vector<int> src;
        for (int k = 0; k < n && src.size() < lim; ++k) {
            const int64_t t = ids[k];
            if (t == IMAGE_TOKEN) {
                REQUIRE(img < n_feat_frames, "more <image> tokens than encoded frames");
                for (int j = 0; j < otok; ++j) src.push_back(-(1 + img * otok + j));
                ++img;
            } else if (t == MEMORY_TOKEN) {
                REQUIRE(n_memory > 0 && mem_used == 0, "<memory> token without (or with repeated) memory frames");
                const int nm = n_memory * otok;
                if (prune_keep > 0 && prune_keep < nm) {          // opt-in: keep the prune_keep least-average memory tokens, in order
                    run_memory_prune(feats + (size_t)frame_base * otok * H, nm, prune_keep);
                    for (int j = 0; j < prune_keep; ++j) src.push_back(-(1 + frame_base * otok + h_sel[j]));
                } else {
                    for (int j = 0; j < nm; ++j) src.push_back(-(1 + frame_base * otok + j));
                }
                mem_used = 1;
            } else {
                REQUIRE(t >= 0 && t < V, "token id out of range");
                src.push_back((int)t);
            }
        }
        if (src.size() > lim) src.resize(lim);
        REQUIRE((int)src.size() <= cap, "inputs_embeds exceeds max_positions");
        const int rows = (int)src.size();
        std::memcpy(h_src, src.data(), (size_t)rows * sizeof(int));
        HIP_CHECK(hipMemcpyAsync(d_src, h_src, rows * sizeof(int), hipMemcpyHostToDevice, st));
        launch_gather_rows<T>(st, d_src, embed, feats, e.embeds + (size_t)e.n_embeds * H, rows, H);
        if (!src_ev) HIP_CHECK(hipEventCreateWithFlags(&src_ev, hipEventDisableTiming));
        HIP_CHECK(hipEventRecord(src_ev, st));       // no stream synchronisation on the hot path: the next call waits for this upload only
        src_pending = true;
        e.n_embeds += rows;
    }

    // ------------------------------------------------------------------------------- LLM
    AttnArgs llm_attn_args(const LLayer& L, const Env& e, const void* q, int ld, void* out, int o_stride, int Tn, int P, int kv_len,
                           bool decode) {
        AttnArgs a; std::memset(&a, 0, sizeof(a));
        a.Q = q; a.q_stride = ld; a.O = out; a.o_stride = o_stride; a.Kpool = L.kpool; a.Vpool = L.vpool; a.page_table = e.d_pages;
        a.n_kv_total = nkv; a.hpf = nkv; a.G = nq / nkv; a.T = Tn; a.P = P; a.kv_len = kv_len; a.scale = 1.0f / sqrtf(128.0f);
        a.part = attn_part; a.rows_pad = 32;
        if (decode) {
            a.causal = 0; a.dyn_kv_len = &d_ctl->kv_len; a.nsplit = nsplit_max; a.tiles_per_split = tiles_per_split;
            a.fuse_rope_append = 1; a.rope_tab = rope_tab; a.dyn_pos = &d_ctl->pos; a.nq_heads = nq; a.skip = &d_ctl->done;
        } else {
            a.causal = 1; a.dyn_kv_len = nullptr; a.nsplit = 1; a.tiles_per_split = pages_per_env;
            // few row blocks (steady turn: 12 x nkv workgroups): split the keys as well so the chip is filled
            const int rows = Tn * a.G, wgs = ((rows + 127) / 128) * nkv, tiles = (kv_len + PAGE - 1) / PAGE;
            // (the ViT's in-workgroup key groups were tried here as well -- 96 workgroups of 4 groups x 2 waves under a 2-way grid split, K
            //  prefetched in registers: steady prefill 9.00 against 9.01-9.03 ms per turn in two alternating rounds, +0.15-0.19 without the
            //  grid split; not kept)
            if (rows <= PREFILL_SPLIT_ROWS && wgs < 128 && tiles >= 4) {
                int ns = (256 + wgs - 1) / wgs;
                if (ns > 8) ns = 8;
                if (ns > tiles / 2) ns = tiles / 2;
                if (ns > 1) { a.nsplit = ns; a.tiles_per_split = (tiles + ns - 1) / ns; a.rows_pad = PREFILL_SPLIT_ROWS; }
            }
        }
        return a;
    }

    struct Seg { Env* e; int P, Tn, off; };       // rows [off, off + Tn) of the prefill batch belong to env e at positions P..
    // Qwen2DecoderLayer stack (modeling_qwen2.py:269-299) over the concatenated new rows of one or several envs:
    // the dense products run once on all rows; RoPE + KV append and attention run per env (own pages / positions).
    // n_dec > 0 (mixed iteration of the multi-env scheduler): rows [0, n_dec) are single-token decode rows of n_dec other envs
    // (x already holds their token embeddings, d_slots their page tables / positions); they share every dense product with the
    // prefill rows and run the batched decode attention (fused RoPE + KV append) instead of the per-segment prefill attention.
    void prefill_rows(const std::vector<Seg>& segs, int M, int n_dec = 0) {
        const int qd = nq * 128;
        for (const Seg& g : segs)
            HIP_CHECK(hipMemcpyAsync(x + (size_t)g.off * H, g.e->embeds + (size_t)g.P * H, (size_t)g.Tn * H * sizeof(T), hipMemcpyDeviceToDevice, st));
        bool xn_ready = false;        // xn already holds rmsnorm(x) * in_norm (written by the previous layer's down_proj epilogue)
        const bool taps = layer_taps_on && segs.size() == 1 && n_dec == 0;
        for (int i = 0; i < c.layers; ++i) {
            const LLayer& L = ll[i];
            const bool probing = taps && i == probe_layer;
            if (probing) { probe_copy(0, x, M); probe_rows = M; }
            if (!xn_ready) launch_rmsnorm<T>(st, x, L.in_norm, xn, M, H, c.rms_eps);
            // one env's turn alone in the batch: the QKV product's split-K reduce also applies RoPE and appends k / v to its pages
            RopeKvArgs r0; r0.qkv = qkv; r0.ld = qkv_dim; r0.Kpool = L.kpool; r0.Vpool = L.vpool; r0.rope_tab = rope_tab;
            r0.nq = nq; r0.nkv = nkv; r0.dyn_pos = nullptr;
            GemmArgs aq = gemm_args(xn, H, L.qkv_w, H, qkv, qkv_dim, L.qkv_b, nullptr, 0, 0, M, qkv_dim, H, EPI_NONE);
            if (segs.size() == 1 && n_dec == 0) { r0.page_table = segs[0].e->d_pages; r0.T = segs[0].Tn; r0.P = segs[0].P; aq.rope = &r0; }
            if (probing) probe_copy(6, xn, M);
            const bool roped = llm_gemm(aq, L.qkv8);
            if (n_dec > 0) {
                AttnArgs a = batched_decode_attn_args(L, n_dec);
                launch_attention<T>(st, a, 128, 1);
                launch_attention_combine<T>(st, a, 128);
            }
            for (const Seg& g : segs) {
                T* q_g = qkv + (size_t)g.off * qkv_dim;
                RopeKvArgs r = r0; r.qkv = q_g; r.page_table = g.e->d_pages;
                r.T = g.Tn; r.P = g.P;
                if (!roped) launch_rope_kv<T>(st, r);
                AttnArgs a = llm_attn_args(L, *g.e, q_g, qkv_dim, attn + (size_t)g.off * qd, qd, g.Tn, g.P, g.P + g.Tn, false);
                launch_attention<T>(st, a, 128, 4);
                if (a.nsplit > 1) launch_attention_combine<T>(st, a, 128);
            }
            // the split-K reduce of o_proj / down_proj also emits the following RMSNorm when it can (T <= 256 rows)
            GemmArgs ao = gemm_args(attn, qd, L.o_w, qd, x, H, nullptr, x, H, 0, M, H, qd, EPI_NONE);
            ao.norm_w = L.post_norm; ao.norm_out = xn; ao.norm_eps = c.rms_eps;
            if (probing) { probe_copy(2, attn, M); probe_copy(7, qkv, M); }
            if (!llm_gemm(ao, L.o8)) launch_rmsnorm<T>(st, x, L.post_norm, xn, M, H, c.rms_eps);
            if (probing) { probe_copy(3, x, M); probe_copy(4, xn, M); }
            const bool pp = i == 0 && probe_on && M <= 256 && n_dec == 0 && pprobe_used + 2 <= pprobe_ev.size();
            if (pp) HIP_CHECK(hipEventRecord(pprobe_ev[pprobe_used], st));
            llm_gemm(gemm_args(xn, H, L.gu_w, H, hbuf, I, nullptr, nullptr, 0, 0, M, 2 * I, H, EPI_SWIGLU), L.gu8);
            if (pp) { HIP_CHECK(hipEventRecord(pprobe_ev[pprobe_used + 1], st)); pprobe_used += 2; pprobe_rows += M; }
            if (probing) probe_copy(5, hbuf, M);
            GemmArgs ad = gemm_args(hbuf, I, L.down_w, I, x, H, nullptr, x, H, 0, M, H, I, EPI_NONE);
            if (i + 1 < c.layers) { ad.norm_w = ll[i + 1].in_norm; ad.norm_out = xn; ad.norm_eps = c.rms_eps; }
            xn_ready = llm_gemm(ad, L.down8);
            if (taps) {
                HIP_CHECK(hipMemcpyAsync(layer_tap + (size_t)i * H, x + (size_t)(M - 1) * H, (size_t)H * sizeof(T), hipMemcpyDeviceToDevice, st));
                if (probing) probe_copy(1, x, M);
            }
        }
    }
    AttnArgs batched_decode_attn_args(const LLayer& L, int B) {
        AttnArgs a = llm_attn_args(L, envs[0], qkv, qkv_dim, attn, nq * 128, 1, 0, 0, true);
        a.page_table = nullptr; a.dyn_kv_len = nullptr; a.dyn_pos = nullptr; a.skip = nullptr;
        a.slots = d_slots; a.batch = B; a.part_bstride = (size_t)nsplit_max * nkv * 32 * (128 + ATTN_PART_PAD);
        return a;
    }
    void prefill(Env& e, int P, int Tn) {
        std::vector<Seg> segs{Seg{&e, P, Tn, 0}};
        prefill_rows(segs, Tn);
    }

    GemvArgs gemv_args(const void* W, int ldw, const void* xin, const void* norm_w, const void* bias, const void* res, void* y, int N, int K,
                       int epi) {
        GemvArgs a; a.W = W; a.ldw = ldw; a.x = xin; a.norm_w = norm_w; a.eps = c.rms_eps; a.bias = bias; a.res = res; a.y = y; a.N = N; a.K = K;
        a.epi = epi; a.part_val = part_val; a.part_idx = part_idx; a.w8 = nullptr; a.scale = nullptr; a.skip = nullptr; return a;
    }
    GemvArgs with8(GemvArgs a, const Q8& q) { if (fp8_on) { a.w8 = q.q; a.scale = q.s; } return a; }
    GemvArgs guarded(GemvArgs a) { a.skip = &d_ctl->done; return a; }      // decode-step launch: no-op once the generation is done
    // final norm -> hidden tap row -> lm_head arg-max -> d_token  (lm_head on the LAST position only).
    // `gen`: the arg-max also runs one step of the greedy loop on the device (GenCtl: append, EOS / max_new stop, advance the position)
    // and every launch is guarded by the done flag.
    // With `gen` the tap row is GenCtl.count on the device (= tap_row on the host: tokens emitted so far), so the three launches are the
    // same for every decode step and sit at the end of the captured step graph.
    void head(const T* xrow, int tap_row, bool gen) {
        T* tap = hid_tap + (size_t)(tap_row < HID_TAP_ROWS ? tap_row : HID_TAP_ROWS - 1) * H;
        const int* skip = gen ? &d_ctl->done : nullptr;
        if (gen) {
            launch_rmsnorm<T>(st, xrow, final_norm, head_xn, 1, H, c.rms_eps, skip, hid_tap, &d_ctl->count, HID_TAP_ROWS);
            tap = head_xn;
        } else {
            launch_rmsnorm<T>(st, xrow, final_norm, tap, 1, H, c.rms_eps, skip);
        }
        GemvArgs a = with8(gemv_args(lm_head, H, tap, nullptr, nullptr, nullptr, nullptr, V, H, EPI_ARGMAX), lm_head8);
        a.skip = skip;
        const bool pen = gen && rep_penalty != 1.0f;
        if (pen) { a.pen_flags = pen_flags; a.pen = rep_penalty; }
        launch_gemv<T>(st, a);
        if (gen) launch_argmax_step(st, part_val, part_idx, gemv_grid(V), d_token, d_top2, d_ctl, d_eos, d_out_ids, pen ? pen_flags : nullptr);
        else launch_argmax_final(st, part_val, part_idx, gemv_grid(V), d_token, d_top2);
    }
    // One decode step as a fixed op sequence; every run-time scalar is read from device memory (d_ctl,
    // d_token) so any sub-range [lo, hi) of the sequence can be captured once and graph-replayed.
    // Op ids: 0 = embedding gather, then 6 per layer (qkv GEMV, attention with fused RoPE + KV append, combine, o GEMV,
    // gate/up GEMV, down GEMV); the layer-0 gate/up GEMV is op probe_op().
    static constexpr int OPS_PER_LAYER = 6;
    // Persistent decode layer (svln_set_decode_persistent; decode_layer.hip): per layer the attention launch + ONE launch for everything
    // behind it (merge of the partials, o_proj, gate/up, down_proj, the next layer's q|k|v), instead of six launches.
    bool persistent_on = false; int n_cus = 0;
    unsigned long long* dl_gran[4] = {nullptr, nullptr, nullptr, nullptr}; unsigned* dl_seq = nullptr; unsigned* dl_giveup = nullptr; unsigned* h_giveup = nullptr;
    DecodeLayerArgs layer_args(int i) {
        const LLayer& L = ll[i];
        DecodeLayerArgs a; std::memset(&a, 0, sizeof(a));
        a.part = attn_part; a.nsplit = nsplit_max; a.tiles_per_split = tiles_per_split; a.dyn_kv_len = &d_ctl->kv_len; a.n_kv = nkv; a.Gq = nq / nkv;
        a.o_w = L.o_w; a.post_norm = L.post_norm; a.gu_w = L.gu_w; a.down_w = L.down_w;
        if (i + 1 < c.layers) { a.next_norm = ll[i + 1].in_norm; a.next_qkv_w = ll[i + 1].qkv_w; a.next_qkv_b = ll[i + 1].qkv_b; }
        a.x = x; a.qkv_out = qkv; a.H = H; a.I = I; a.qd = nq * 128; a.qkv_dim = qkv_dim; a.eps = c.rms_eps;
        for (int k = 0; k < 4; ++k) a.gran[k] = dl_gran[k];
        a.seq = dl_seq; a.giveup = dl_giveup; a.skip = &d_ctl->done;
        return a;
    }
    void set_decode_persistent(int enable) override {
        HIP_CHECK(hipStreamSynchronize(st));
        if (!enable) { if (persistent_on) drop_graphs(); persistent_on = false; return; }
        if (!n_cus) { hipDeviceProp_t pr; HIP_CHECK(hipGetDeviceProperties(&pr, device)); n_cus = pr.multiProcessorCount; }
        DecodeLayerArgs a = layer_args(0);
        REQUIRE(decode_layer_supported<T>(a, n_cus), "persistent decode layer: this configuration is not supported (per-CU slices of hidden / inter / q|k|v "
                                                   "must be whole, rows whole KiB, LDS <= 160 KiB)");
        if (!dl_seq) {
            const size_t ng[4] = {(size_t)nq * 128, (size_t)H, (size_t)I, (size_t)H};
            for (int k = 0; k < 4; ++k) dl_gran[k] = dalloc<unsigned long long>(ng[k], true);
            dl_seq = dalloc<unsigned>(64, true);          // [0] = launch counter, [16] = give-up word (own cache lines)
            dl_giveup = dl_seq + 16;
            HIP_CHECK(hipHostMalloc((void**)&h_giveup, 64));
            h_giveup[0] = 0;
            HIP_CHECK(hipStreamSynchronize(st));
        }
        if (!persistent_on) drop_graphs();
        persistent_on = true;
    }
    // diagnostic: ONE persistent launch of `layer` on whatever the buffers hold after the last decode step, with phase stamps per workgroup
    void probe_decode_layer(int layer, unsigned long long* out, int max_wgs, int32_t* n_wgs) override {
        REQUIRE(persistent_on && layer >= 0 && layer < c.layers, "persistent decode layer is off / no such layer");
        unsigned long long* d = nullptr;
        HIP_CHECK(hipMalloc((void**)&d, (size_t)n_cus * 16 * 8));
        HIP_CHECK(hipMemsetAsync(d, 0, (size_t)n_cus * 16 * 8, st));
        DecodeLayerArgs a = layer_args(layer);
        a.skip = nullptr; a.dbg = d;
        launch_decode_layer<T>(st, a, n_cus);
        const int n = n_cus < max_wgs ? n_cus : max_wgs;
        HIP_CHECK(hipMemcpyAsync(out, d, (size_t)n * 16 * 8, hipMemcpyDeviceToHost, st));
        HIP_CHECK(hipStreamSynchronize(st));
        HIP_CHECK(hipFree(d));
        *n_wgs = n;
    }
    bool persistent_active() const { return persistent_on && !fp8_on; }       // (the e4m3 decode weights keep the launched GEMVs)
    int probe_op() const { return persistent_active() ? 3 : 1 + 4; }          // the launch the roofline probe times (layer 0)
    int total_ops() const { return persistent_active() ? 2 + 2 * c.layers : 1 + c.layers * OPS_PER_LAYER; }
    void decode_ops(Env& e, int lo, int hi) {
        const int qd = nq * 128;
        int op = 0;
        auto on = [&](void) { const bool r = op >= lo && op < hi; ++op; return r; };
        if (on()) launch_gather_rows<T>(st, d_token, embed, feats, x, 1, H, &d_ctl->done);
        if (persistent_active()) {
            // op 1: layer 0's q|k|v rows; then per layer: decode attention (RoPE, KV append, per-page partials), the persistent layer
            if (on()) launch_gemv<T>(st, guarded(gemv_args(ll[0].qkv_w, H, x, ll[0].in_norm, ll[0].qkv_b, nullptr, qkv, qkv_dim, H, EPI_NONE)));
            for (int i = 0; i < c.layers; ++i) {
                AttnArgs a = llm_attn_args(ll[i], e, qkv, qkv_dim, attn, qd, 1, 0, 0, true);
                if (on()) launch_attention<T>(st, a, 128, 1);
                if (on()) launch_decode_layer<T>(st, layer_args(i), n_cus);
            }
            return;
        }
        for (int i = 0; i < c.layers; ++i) {
            const LLayer& L = ll[i];
            if (on()) launch_gemv<T>(st, guarded(with8(gemv_args(L.qkv_w, H, x, L.in_norm, L.qkv_b, nullptr, qkv, qkv_dim, H, EPI_NONE), L.qkv8)));
            AttnArgs a = llm_attn_args(L, e, qkv, qkv_dim, attn, qd, 1, 0, 0, true);
            if (on()) launch_attention<T>(st, a, 128, 1);
            if (on()) launch_attention_combine<T>(st, a, 128);
            if (on()) launch_gemv<T>(st, guarded(with8(gemv_args(L.o_w, qd, attn, nullptr, nullptr, x, x, H, qd, EPI_NONE), L.o8)));
            if (on()) launch_gemv<T>(st, guarded(with8(gemv_args(L.gu_w, H, x, L.post_norm, nullptr, nullptr, hbuf, 2 * I, H, EPI_SWIGLU), L.gu8)));
            if (on()) launch_gemv<T>(st, guarded(with8(gemv_args(L.down_w, I, hbuf, nullptr, nullptr, x, x, H, I, EPI_NONE), L.down8)));
        }
    }

    // layer-0 gate/up SwiGLU GEMV with the kernel's own begin/end timestamps
    void probe_launch(Env&) {
        const LLayer& L = ll[0];
        if (persistent_active()) launch_decode_layer<T>(st, layer_args(0), n_cus, probe_ev[probe_used], probe_ev[probe_used + 1]);
        else launch_gemv_timed<T>(st, guarded(with8(gemv_args(L.gu_w, H, x, L.post_norm, nullptr, nullptr, hbuf, 2 * I, H, EPI_SWIGLU), L.gu8)), probe_ev[probe_used],
                                  probe_ev[probe_used + 1]);
        probe_used += 2;
    }
    // graph of: ops [lo, hi) of one decode step (+ the head when hi is the end of the step), then `more` further whole steps
    hipGraphExec_t capture(Env& e, int lo, int hi, int more = 0) {
        return capture_graph([&] {
            decode_ops(e, lo, hi);
            if (hi == total_ops()) head(x, 0, true);       // (the tap row comes from the device)
            for (int k = 0; k < more; ++k) { decode_ops(e, 0, total_ops()); head(x, 0, true); }
        });
    }
    // Capture `body` (launches on `st` only) into an executable graph.  An exception thrown inside the capture (REQUIRE / HIP_CHECK /
    // LAUNCH_CHECK of a launcher) must not leave the stream in capture mode: the capture is ended, the partial graph destroyed and the
    // host-side state a captured product may have left behind (act8_src) reset before the exception travels on.
    template <typename F> hipGraphExec_t capture_graph(F&& body) {
        hipGraph_t g = nullptr; hipGraphExec_t ex = nullptr;
        HIP_CHECK(hipStreamBeginCapture(st, hipStreamCaptureModeThreadLocal));
        try {
            body();
        } catch (...) {
            (void)hipStreamEndCapture(st, &g);
            if (g) (void)hipGraphDestroy(g);
            (void)hipGetLastError();
            act8_src = nullptr;
            throw;
        }
        HIP_CHECK(hipStreamEndCapture(st, &g));
        const hipError_t inst = hipGraphInstantiate(&ex, g, nullptr, nullptr, 0);
        (void)hipGraphDestroy(g);
        HIP_CHECK(inst);
        return ex;
    }
    void drop_graphs() {
        for (auto& gs : graphs) {
            for (auto& kv : gs.ex) if (kv.second) (void)hipGraphExecDestroy(kv.second);
            gs.ex.clear();
        }
        for (auto& kv : bgraphs) if (kv.second) (void)hipGraphExecDestroy(kv.second);
        bgraphs.clear();
    }
    // Enqueue `steps` decode steps (position, kv length and the fed token live in device memory: GenCtl / d_token); nothing here waits
    // for the GPU.  With hipGraph replay the whole run-ahead batch is ONE graph launch (every launch of a ~175-node graph left a
    // 50-70 us bubble in the kernel trace of round 2: four launches per turn).  The roofline probe times the layer-0 gate/up GEMV of
    // the FIRST decode step of every turn: that step is split around one plain timed launch (same kernel, grid and bytes as the other
    // layers' gate/up GEMVs inside the graph), the remaining steps ride in the second graph.
    void decode_steps(Env& e, int env, int first_tap_row, int steps) {
        if (steps <= 0) return;
        const int n_ops = total_ops();
        const bool probing = probe_on && first_tap_row == 1 && probe_used + 2 <= probe_ev.size();
        if (use_graph) {
            GraphSet& gs = graphs[env];
            auto get = [&](int key, int lo, int hi, int more) {
                auto it = gs.ex.find(key);
                if (it == gs.ex.end()) it = gs.ex.emplace(key, capture(e, lo, hi, more)).first;
                return it->second;
            };
            if (!probing) {
                HIP_CHECK(hipGraphLaunch(get(steps, 0, n_ops, steps - 1), st));
            } else {
                HIP_CHECK(hipGraphLaunch(get(1000, 0, probe_op(), 0), st));
                probe_launch(e);
                HIP_CHECK(hipGraphLaunch(get(2000 + steps, probe_op() + 1, n_ops, steps - 1), st));
            }
            return;
        }
        for (int k = 0; k < steps; ++k) {
            if (!(probing && k == 0)) {
                decode_ops(e, 0, n_ops);
            } else {
                decode_ops(e, 0, probe_op());
                probe_launch(e);
                decode_ops(e, probe_op() + 1, n_ops);
            }
            head(x, first_tap_row + k, true);
        }
    }

    // ------------------------------------------------------------------------------- multi-env lockstep
    GemvBatchArgs gemvb_args(const void* W, int ldw, const void* xin, int ldx, const void* norm_w, const void* bias, const void* res, int ldr,
                             void* y, int ldy, int N, int K, int epi, int B) {
        GemvBatchArgs a; a.W = W; a.ldw = ldw; a.x = xin; a.ldx = ldx; a.norm_w = norm_w; a.eps = c.rms_eps; a.bias = bias; a.res = res; a.ldr = ldr;
        a.y = y; a.ldy = ldy; a.N = N; a.K = K; a.epi = epi; a.B = B; a.part_val = part_val_b; a.part_idx = part_idx_b; return a;
    }
    // final norm of rows[0..B) -> lm_head once for all B envs -> d_tok_b[0..B)
    // pen: the repetition penalty is on and d_pen_rows[0..B) holds the job slot (= flag row) of every batch row
    // arg-max over W [N][K] . x_b for B rows -> d_tok_b.  B >= 4: one pass of 32-row MFMA tiles with the arg-max in the epilogue (the
    // batched GEMV is dot-product-issue bound from B = 4: 386 us at B = 8 on the full vocabulary); B <= 2: the batched GEMV.
    void argmax_rows(const void* W, int ldw, const T* xrows, int ldx, int N, int K, int B, bool pen) {
        if (B >= batched_mfma_min && K % Elt<T>::PER_CHUNK == 0 && (N + 127) / 128 <= 2048) {
            GemmArgs a = gemm_args(xrows, ldx, W, ldw, nullptr, 0, nullptr, nullptr, 0, 0, B, N, K, EPI_ARGMAX);
            a.part_val = part_val_b; a.part_idx = part_idx_b;
            if (pen) { a.pen_flags = pen_flags_b; a.pen_rows = d_pen_rows; a.pen = rep_penalty; }
            const int n = launch_gemm_argmax<T>(st, a);
            launch_argmax_final_batched(st, part_val_b, part_idx_b, n, B, d_tok_b);
            return;
        }
        GemvBatchArgs hb = gemvb_args(W, ldw, xrows, ldx, nullptr, nullptr, nullptr, 0, nullptr, 0, N, K, EPI_ARGMAX, B);
        if (pen) { hb.pen_flags = pen_flags_b; hb.pen_rows = d_pen_rows; hb.pen = rep_penalty; }
        launch_gemv_batched<T>(st, hb);
        launch_argmax_final_batched(st, part_val_b, part_idx_b, gemv_batched_grid(N, EPI_ARGMAX, B), B, d_tok_b);
    }
    void head_batched(const T* rows, int B, bool pen = false) {
        launch_rmsnorm<T>(st, rows, final_norm, xn, B, H, c.rms_eps);
        argmax_rows(lm_head, H, xn, H, V, H, B, pen);
    }
    // One decode step for B envs (B in {1,2,4,8}; d_slots / d_tok_b already set): every weight matrix is streamed once.
    // B >= 4: the projections run as 32-row MFMA products (gemm.hip CfgSkinny: the weight stream goes through LDS-DMA, the B rows
    // ride along; the batched GEMV is dot-product-issue bound from B = 4 up) and the split-K reduce of o_proj / down_proj also emits
    // the following RMSNorm.  B <= 2: the batched GEMV (HBM-bound there).
    void decode_ops_batched(int B, bool pen = false) {
        const int qd = nq * 128;
        const bool mfma = B >= batched_mfma_min;
        launch_gather_rows<T>(st, d_tok_b, embed, feats, x, B, H);
        bool xn_ready = false;
        for (int i = 0; i < c.layers; ++i) {
            const LLayer& L = ll[i];
            // RMSNorm as its own tiny launch in the batched step (amortised over B envs)
            if (!xn_ready) launch_rmsnorm<T>(st, x, L.in_norm, xn, B, H, c.rms_eps);
            if (mfma) llm_gemm(gemm_args(xn, H, L.qkv_w, H, qkv, qkv_dim, L.qkv_b, nullptr, 0, 0, B, qkv_dim, H, EPI_NONE), L.qkv8);
            else launch_gemv_batched<T>(st, gemvb_args(L.qkv_w, H, xn, H, nullptr, L.qkv_b, nullptr, 0, qkv, qkv_dim, qkv_dim, H, EPI_NONE, B));
            AttnArgs a = batched_decode_attn_args(L, B);
            launch_attention<T>(st, a, 128, 1);
            launch_attention_combine<T>(st, a, 128);
            if (mfma) {
                GemmArgs ao = gemm_args(attn, qd, L.o_w, qd, x, H, nullptr, x, H, 0, B, H, qd, EPI_NONE);
                ao.norm_w = L.post_norm; ao.norm_out = xn; ao.norm_eps = c.rms_eps;
                if (!llm_gemm(ao, L.o8)) launch_rmsnorm<T>(st, x, L.post_norm, xn, B, H, c.rms_eps);
                llm_gemm(gemm_args(xn, H, L.gu_w, H, hbuf, I, nullptr, nullptr, 0, 0, B, 2 * I, H, EPI_SWIGLU), L.gu8);
                GemmArgs ad = gemm_args(hbuf, I, L.down_w, I, x, H, nullptr, x, H, 0, B, H, I, EPI_NONE);
                if (i + 1 < c.layers) { ad.norm_w = ll[i + 1].in_norm; ad.norm_out = xn; ad.norm_eps = c.rms_eps; }
                xn_ready = llm_gemm(ad, L.down8);
            } else {
                launch_gemv_batched<T>(st, gemvb_args(L.o_w, qd, attn, qd, nullptr, nullptr, x, H, x, H, H, qd, EPI_NONE, B));
                launch_rmsnorm<T>(st, x, L.post_norm, xn, B, H, c.rms_eps);
                launch_gemv_batched<T>(st, gemvb_args(L.gu_w, H, xn, H, nullptr, nullptr, nullptr, 0, hbuf, I, 2 * I, H, EPI_SWIGLU, B));
                launch_gemv_batched<T>(st, gemvb_args(L.down_w, I, hbuf, I, nullptr, nullptr, x, H, x, H, H, I, EPI_NONE, B));
            }
        }
        head_batched(x, B, pen);
    }
    bool taps_on = true;
    void tap_copy(const T* row, int token_idx, int slot) {       // parity tap: hid_tap[min(token,7)][slot]
        if (!taps_on) return;
        const int t = token_idx < 8 ? token_idx : 7;
        HIP_CHECK(hipMemcpyAsync(hid_tap + (size_t)(t * MAXB + slot) * H, row, (size_t)H * sizeof(T), hipMemcpyDeviceToDevice, st));
    }

    // ---- multi-env scheduler (caller = a DAgger-style collector whose envs mix expert and model steps, so their model
    // turns fall due at different times: streamvln_dagger.py:232-313).  Per-env semantics are those of svln_generate; execution is
    // iteration-level batching: every iteration is ONE pass over the weights that carries, for each active env, either the rows of
    // its new turn (prefill) or the single row of the token it generated in the previous iteration (decode).  Envs join at any
    // iteration (svln_batch_submit) and leave when they emit EOS / max_new_tokens, so prefilling and decoding envs share passes.
    struct Job {
        bool used = false, finished = false, prefill = true;
        int env = -1, max_new = 0, count = 0, last_tok = -1;
        std::vector<int64_t> out, eos;
    };
    Job jobs[MAXB];
    // forget a job: its slot becomes free, its repetition-penalty flags are cleared (the stream is idle between scheduler calls)
    void drop_job(int k) {
        Job& j = jobs[k];
        if (!j.used) return;
        if (pen_flags_b && !j.out.empty()) {
            const int n = (int)j.out.size() < c.max_positions ? (int)j.out.size() : c.max_positions;
            for (int q = 0; q < n; ++q) h_pen_ids[q] = (int)j.out[q];
            (void)hipMemcpyAsync(d_pen_ids, h_pen_ids, (size_t)n * sizeof(int), hipMemcpyHostToDevice, st);
            launch_set_flags(st, pen_flags_b + (size_t)k * V, d_pen_ids, nullptr, n, 0);
            (void)hipStreamSynchronize(st);          // h_pen_ids is reused by the next drop
        }
        j = Job();
    }
    void cancel_jobs_of(int env) {
        for (int k = 0; k < MAXB; ++k) if (jobs[k].used && jobs[k].env == env) drop_job(k);
    }
    // svln_batch_cancel: slot >= 0 drops that turn, slot < 0 every turn in flight.  The envs keep whatever rows / KV the dropped turns
    // had already written (a cancelled prefill leaves kv_len behind n_embeds: reset the env, or submit it again to finish the prefill).
    void batch_cancel(int slot) override {
        REQUIRE(slot < MAXB, "no such scheduler slot");
        HIP_CHECK(hipStreamSynchronize(st));
        if (slot >= 0) drop_job(slot);
        else for (int k = 0; k < MAXB; ++k) drop_job(k);
    }
    void ensure_pen_buffers() {
        if (pen_flags) return;
        pen_flags = dalloc<uint8_t>((size_t)V, true);
        pen_flags_b = dalloc<uint8_t>((size_t)MAXB * V, true);
        d_pen_rows = dalloc<int>(MAXB, true);
        d_pen_ids = dalloc<int>((size_t)c.max_positions + 8);
        HIP_CHECK(hipHostMalloc((void**)&h_pen_rows, MAXB * sizeof(int)));
        HIP_CHECK(hipHostMalloc((void**)&h_pen_ids, ((size_t)c.max_positions + 8) * sizeof(int)));
        HIP_CHECK(hipStreamSynchronize(st));
    }
    // generation_config.repetition_penalty of a checkpoint (HF RepetitionPenaltyLogitsProcessor over the tokens generated in the turn;
    // transformers 4.45.1 applies it under greedy decoding too).  1 = off (the default: no kernel sees a flag pointer).
    void set_repetition_penalty(float penalty) override {
        REQUIRE(penalty > 0.0f, "repetition_penalty must be > 0");
        for (int k = 0; k < MAXB; ++k) REQUIRE(!jobs[k].used, "repetition_penalty cannot change while turns are in flight");
        HIP_CHECK(hipStreamSynchronize(st));
        if (penalty != 1.0f) ensure_pen_buffers();
        if (penalty != rep_penalty) {
            drop_graphs();           // the captured lm_head launch holds the flag pointer and the factor
            // flags of the last penalised turn must not survive a change of the factor: an unpenalised generate() in between rewrites
            // d_out_ids / GenCtl.count without touching them, so the next penalised turn would clear the wrong ids (P -> 1 -> P)
            if (pen_flags) {
                HIP_CHECK(hipMemsetAsync(pen_flags, 0, (size_t)V, st));
                HIP_CHECK(hipMemsetAsync(pen_flags_b, 0, (size_t)MAXB * V, st));
                HIP_CHECK(hipStreamSynchronize(st));
            }
        }
        rep_penalty = penalty;
    }
    void set_turn_row_limit(int rows) override { REQUIRE(rows >= 0, "row limit must be >= 0 (0 = none)"); turn_row_limit = rows; }
    int batch_submit(int env, int max_new, const int64_t* eos, int n_eos) override {
        Env& e = env_at(env);
        REQUIRE(weights_missing() == 0, g_err);
        REQUIRE(max_new >= 1, "max_new_tokens must be >= 1");
        REQUIRE(e.n_embeds - e.kv_len >= 1, "nothing to prefill for this env (append its turn first)");
        int slot = -1;
        for (int k = 0; k < MAXB; ++k) {
            REQUIRE(!(jobs[k].used && jobs[k].env == env), "this env already has a turn in flight");
            if (slot < 0 && !jobs[k].used) slot = k;
        }
        REQUIRE(slot >= 0, "at most 8 turns in flight");
        Job& j = jobs[slot];
        j = Job();
        j.used = true; j.env = env; j.max_new = max_new < c.max_positions ? max_new : c.max_positions;
        j.eos.assign(eos, eos + n_eos);
        n_generated_b[slot] = 0;
        return slot;
    }
    // one iteration; returns the number of jobs still running afterwards, finished_slots = jobs that completed in this iteration
    int batch_step(int32_t* finished_slots, int32_t* n_finished) override {
        *n_finished = 0;
        // a turn whose env has nothing left to prefill (its rows were reset or already consumed) is dropped alone, with an error that
        // names it; the other turns stay valid for the next call
        for (int k = 0; k < MAXB; ++k) {
            Job& j = jobs[k];
            if (!j.used || j.finished || !j.prefill) continue;
            const Env& e = envs[j.env];
            if (e.n_embeds - e.kv_len < 1) {
                const int env = j.env;
                drop_job(k);
                REQUIRE(false, "batch_step: env " + std::to_string(env) + " has no rows to prefill (reset after submit?); its turn was dropped");
            }
        }
        try {
            return batch_step_run(finished_slots, n_finished);
        } catch (...) {
            // the iteration failed part-way (page pool exhausted, non-finite logits, a HIP error): which jobs advanced is unknown, so every
            // turn in flight is dropped and the scheduler is idle again; the envs keep their rows (reset them or submit again)
            (void)hipStreamSynchronize(st);
            for (int k = 0; k < MAXB; ++k) drop_job(k);
            *n_finished = 0;
            throw;
        }
    }
    int batch_step_run(int32_t* finished_slots, int32_t* n_finished) {
        std::vector<int> dec, pre;                      // job slots decoding / prefilling in this iteration
        for (int k = 0; k < MAXB; ++k)
            if (jobs[k].used && !jobs[k].finished) (jobs[k].prefill ? pre : dec).push_back(k);
        if (dec.empty() && pre.empty()) return 0;
        const int nd = (int)dec.size();
        const bool pen = rep_penalty != 1.0f;
        // prefill jobs that fit the row workspaces next to the decode rows (the rest wait for the next iteration; a turn of up to
        // max_positions rows runs as soon as no decode row shares the pass)
        std::vector<Seg> segs;
        std::vector<int> pre_now;
        int M = nd;
        for (int k : pre) {
            Env& e = envs[jobs[k].env];
            const int Tn = e.n_embeds - e.kv_len;
            REQUIRE(Tn >= 1 && Tn <= c.max_positions, "prefill length out of range");
            if (M + Tn > c.max_positions) continue;
            ensure_pages(e, e.n_embeds);
            segs.push_back(Seg{&e, e.kv_len, Tn, M});
            pre_now.push_back(k);
            M += Tn;
        }
        HIP_CHECK(hipEventRecord(ph_ev[2], st));
        std::vector<int> order;                         // job slot of each output token of this iteration
        if (nd > 0) {
            int B = nd;
            if (segs.empty()) { B = 1; while (B < nd) B <<= 1; }      // the pure decode kernels come in power-of-two batch sizes
            for (int k = 0; k < B; ++k) {
                const int kk = k < nd ? k : 0;                          // padding slots replay slot 0 (same writes, ignored outputs)
                Env& e = envs[jobs[dec[kk]].env];
                REQUIRE(e.kv_len + 1 <= c.max_positions, "sequence exceeds max_positions during decode");
                if (k < nd) ensure_pages(e, e.kv_len + 1);
                h_slots[k].page_table = e.d_pages; h_slots[k].pos = e.kv_len; h_slots[k].pad = 0;
                h_tok_b[k] = jobs[dec[kk]].last_tok;
                if (pen) h_pen_rows[k] = dec[kk];
            }
            HIP_CHECK(hipMemcpyAsync(d_slots, h_slots, B * sizeof(DecodeSlot), hipMemcpyHostToDevice, st));
            HIP_CHECK(hipMemcpyAsync(d_tok_b, h_tok_b, B * sizeof(int), hipMemcpyHostToDevice, st));
            if (segs.empty()) {
                if (pen) HIP_CHECK(hipMemcpyAsync(d_pen_rows, h_pen_rows, B * sizeof(int), hipMemcpyHostToDevice, st));
                // gather + 28 layers + head for B decode rows -> d_tok_b, xn = final-norm rows.  With hipGraph replay on, the step of each
                // (B, fp8, penalty) combination is captured once: every run-time value (page tables, positions, fed tokens) is in d_slots / d_tok_b.
                if (use_graph) {
                    const int key = B | (pen ? 256 : 0) | (fp8_gemm_on ? 512 : 0);
                    auto it = bgraphs.find(key);
                    if (it == bgraphs.end())      // (a failed capture leaves no entry behind)
                        it = bgraphs.emplace(key, capture_graph([&] { decode_ops_batched(B, pen); })).first;
                    HIP_CHECK(hipGraphLaunch(it->second, st));
                } else {
                    decode_ops_batched(B, pen);
                }
            } else {
                launch_gather_rows<T>(st, d_tok_b, embed, feats, x, nd, H);
            }
            for (int k = 0; k < nd; ++k) order.push_back(dec[k]);
        }
        if (!segs.empty()) {
            prefill_rows(segs, M, nd);
            // last row of every job of this iteration -> one lm_head pass
            const int nj = nd + (int)segs.size();
            for (int k = 0; k < nd; ++k)
                HIP_CHECK(hipMemcpyAsync(last_rows + (size_t)k * H, x + (size_t)k * H, (size_t)H * sizeof(T), hipMemcpyDeviceToDevice, st));
            for (size_t q = 0; q < segs.size(); ++q) {
                const T* last = x + (size_t)(segs[q].off + segs[q].Tn - 1) * H;
                HIP_CHECK(hipMemcpyAsync(last_rows + (size_t)(nd + q) * H, last, (size_t)H * sizeof(T), hipMemcpyDeviceToDevice, st));
                order.push_back(pre_now[q]);
            }
            int Bp = 1; while (Bp < nj) Bp <<= 1;
            for (int k = nj; k < Bp; ++k)
                HIP_CHECK(hipMemcpyAsync(last_rows + (size_t)k * H, last_rows, (size_t)H * sizeof(T), hipMemcpyDeviceToDevice, st));
            if (pen) {
                for (int k = 0; k < Bp; ++k) h_pen_rows[k] = order[k < nj ? k : 0];
                HIP_CHECK(hipMemcpyAsync(d_pen_rows, h_pen_rows, Bp * sizeof(int), hipMemcpyHostToDevice, st));
            }
            head_batched(last_rows, Bp, pen);
        }
        const int nj = (int)order.size();
        for (int q = 0; q < nj; ++q) tap_copy(xn + (size_t)q * H, jobs[order[q]].count, order[q]);
        HIP_CHECK(hipEventRecord(ph_ev[3], st));
        HIP_CHECK(hipMemcpyAsync(h_tok_b, d_tok_b, nj * sizeof(int), hipMemcpyDeviceToHost, st));
        HIP_CHECK(hipStreamSynchronize(st));
        LAUNCH_CHECK("batch_step");
        {
            float t = 0.f;
            HIP_CHECK(hipEventElapsedTime(&t, ph_ev[2], ph_ev[3])); ph_ms[segs.empty() ? 2 : 1] += t;
            if (vision_pending) { HIP_CHECK(hipEventElapsedTime(&t, ph_ev[0], ph_ev[1])); ph_ms[0] += t; vision_pending = false; }
        }
        for (int q = 0; q < nj; ++q) {
            Job& j = jobs[order[q]];
            Env& e = envs[j.env];
            const int tok = h_tok_b[q];
            REQUIRE(tok >= 0 && tok < V, "non-finite logits: the arg-max found no finite value (check the weights / fp8 scales)");
            if (j.prefill) { e.kv_len = e.n_embeds; j.prefill = false; }      // this iteration prefilled the turn
            else e.kv_len += 1;                                               // ... or fed the previous token
            j.out.push_back(tok);
            j.count += 1;
            j.last_tok = tok;
            bool stop = j.count >= j.max_new;
            for (int64_t id : j.eos) stop |= id == tok;
            if (pen && !stop) {                                               // the token counts as generated for this job's next arg-max
                h_pen_ids[q] = tok;
                HIP_CHECK(hipMemcpyAsync(d_pen_ids + q, h_pen_ids + q, sizeof(int), hipMemcpyHostToDevice, st));
                launch_set_flags(st, pen_flags_b + (size_t)order[q] * V, d_pen_ids + q, nullptr, 1, 1);
            }
            if (stop) {
                j.finished = true;
                n_generated_b[order[q]] = j.count;
                finished_slots[(*n_finished)++] = order[q];
            }
        }
        if (pen) HIP_CHECK(hipStreamSynchronize(st));                         // h_pen_ids is rewritten by the next iteration
        int running = 0;
        for (int k = 0; k < MAXB; ++k) running += jobs[k].used && !jobs[k].finished;
        return running;
    }
    void batch_result(int slot, int32_t* env, int64_t* out, int cap, int32_t* n_out) override {
        REQUIRE(slot >= 0 && slot < MAXB && jobs[slot].used && jobs[slot].finished, "no finished turn in this slot");
        Job& j = jobs[slot];
        const int n = (int)j.out.size();
        for (int k = 0; k < n && k < cap; ++k) out[k] = j.out[k];
        *n_out = n; *env = j.env;
        drop_job(slot);
    }
    // svln_generate on each listed env, executed together: submit all, iterate until all are done (all envs prefill in the first
    // iteration and decode in lockstep afterwards: the special case of the scheduler in which every turn falls due at once)
    void generate_batch(const int32_t* env_ids, int n_envs, int max_new, const int64_t* eos, int n_eos, int64_t* out, int cap,
                        int32_t* n_out) override {
        REQUIRE(n_envs >= 1 && n_envs <= MAXB, "1..8 envs per batch");
        REQUIRE(max_new >= 1 && cap >= 1, "max_new_tokens must be >= 1");
        for (int k = 0; k < MAXB; ++k) REQUIRE(!jobs[k].used, "svln_generate_batch needs an idle scheduler (turns are in flight)");
        for (int s = 0; s < n_envs; ++s)
            for (int t = 0; t < s; ++t) REQUIRE(env_ids[t] != env_ids[s], "duplicate env in batch");
        std::vector<int> slots(n_envs);
        try {
            for (int s = 0; s < n_envs; ++s) slots[s] = batch_submit(env_ids[s], max_new < cap ? max_new : cap, eos, n_eos);
            int32_t fin[MAXB], nf = 0;
            while (batch_step(fin, &nf) > 0) {}
        } catch (...) {
            for (int k = 0; k < MAXB; ++k) drop_job(k);
            throw;
        }
        for (int s = 0; s < n_envs; ++s) {
            int32_t env = 0;
            batch_result(slots[s], &env, out + (size_t)s * cap, cap, &n_out[s]);
        }
    }
    void get_hidden_batch(int slot, float* out, int max_rows, int32_t* n_rows) override {
        REQUIRE(slot >= 0 && slot < MAXB, "slot");
        int n = n_generated_b[slot] < 8 ? n_generated_b[slot] : 8;
        if (n > max_rows) n = max_rows;
        for (int t = 0; t < n; ++t) read_rows_f32(hid_tap + (size_t)(t * MAXB + slot) * H, (size_t)H, out + (size_t)t * H);
        *n_rows = n;
    }

    int read_token() {
        HIP_CHECK(hipMemcpyAsync(h_token, d_token, sizeof(int), hipMemcpyDeviceToHost, st));
        HIP_CHECK(hipMemcpyAsync(h_top2, d_top2, 2 * sizeof(float), hipMemcpyDeviceToHost, st));
        HIP_CHECK(hipStreamSynchronize(st));
        return h_token[0];
    }

    // StreamVLNForCausalLM.generate -> GenerationMixin._sample (greedy): prefill embeds[kv_len:], then feed each generated id until
    // one is in the EOS set (appended, not fed) or max_new_tokens.  The loop state lives on the device (GenCtl): the prefill, its
    // arg-max and RUN_AHEAD decode steps are enqueued back to back, then ONE synchronisation reads the ids emitted so far; steps
    // enqueued past the end of the generation are no-ops (every kernel checks the done flag).
    void generate(int env, int max_new, const int64_t* eos, int n_eos, int64_t* out, int cap, int32_t* n_out, bool fixed) override {
        Env& e = env_at(env);
        REQUIRE(weights_missing() == 0, g_err);
        const int P = e.kv_len, L = e.n_embeds, Tn = L - P;
        REQUIRE(Tn >= 1, "nothing to prefill: inputs_embeds not longer than the KV cache");
        REQUIRE(max_new >= 1 && cap >= 1, "max_new_tokens must be >= 1");
        if (fixed) n_eos = 0;
        REQUIRE(n_eos >= 0 && n_eos <= (V > 16 ? V : 16), "too many eos ids");
        const int limit = max_new < cap ? max_new : cap;                     // tokens this call may emit
        {   // EOS set -> device (cached across calls: the harness passes the same list every turn)
            std::vector<int> ev(n_eos);
            for (int k = 0; k < n_eos; ++k) ev[k] = eos[k] >= 0 && eos[k] < V ? (int)eos[k] : -2;      // ids outside the vocabulary never match
            if (!eos_valid || ev != eos_cached) {
                HIP_CHECK(hipStreamSynchronize(st));
                eos_cached.swap(ev);
                if (n_eos) HIP_CHECK(hipMemcpy(d_eos, eos_cached.data(), (size_t)n_eos * sizeof(int), hipMemcpyHostToDevice));
                eos_valid = true;
            }
        }
        ensure_pages(e, L);
        // the first decode step feeds token 0 at position L: pos / kv_len advance when the arg-max step appends without stopping
        h_ctl->pos = L - 1; h_ctl->kv_len = L; h_ctl->done = 0; h_ctl->count = 0; h_ctl->max_new = limit; h_ctl->n_eos = n_eos;
        h_ctl->pad0 = h_ctl->pad1 = 0;
        if (rep_penalty != 1.0f)      // tokens of the previous generate (still listed in d_out_ids[0 .. d_ctl.count)) no longer count
            launch_set_flags(st, pen_flags, d_out_ids, &d_ctl->count, 0, 0);
        HIP_CHECK(hipMemcpyAsync(d_ctl, h_ctl, sizeof(GenCtl), hipMemcpyHostToDevice, st));
        HIP_CHECK(hipEventRecord(ph_ev[2], st));
        prefill(e, P, Tn);
        head(x + (size_t)(Tn - 1) * H, 0, true);
        e.kv_len = L;
        HIP_CHECK(hipEventRecord(ph_ev[3], st));
        int enq = 1;                      // tokens whose arg-max step has been enqueued
        int n = 0;
        bool done = false, decoded = false;
        while (true) {
            int steps = limit - enq;
            if (steps > RUN_AHEAD) steps = RUN_AHEAD;
            if (steps > c.max_positions - (L + enq - 1)) steps = c.max_positions - (L + enq - 1);       // step k feeds position L + enq - 1
            if (steps > 0) {
                ensure_pages(e, L + enq + steps - 1);          // pages of every position the batch may write, before it is enqueued
                decode_steps(e, env, enq, steps);
                enq += steps;
                decoded = true;
            }
            if (decoded) HIP_CHECK(hipEventRecord(ph_ev[4], st));
            HIP_CHECK(hipMemcpyAsync(h_ctl, d_ctl, sizeof(GenCtl), hipMemcpyDeviceToHost, st));
            HIP_CHECK(hipMemcpyAsync(h_out_ids, d_out_ids, (size_t)enq * sizeof(int), hipMemcpyDeviceToHost, st));
            HIP_CHECK(hipMemcpyAsync(h_top2, d_top2, 2 * sizeof(float), hipMemcpyDeviceToHost, st));
            if (persistent_on) HIP_CHECK(hipMemcpyAsync(h_giveup, dl_giveup, sizeof(unsigned), hipMemcpyDeviceToHost, st));
            HIP_CHECK(hipStreamSynchronize(st));
            LAUNCH_CHECK("generate");
            if (persistent_on && h_giveup[0]) {
                const unsigned code = h_giveup[0];
                h_giveup[0] = 0;
                HIP_CHECK(hipMemsetAsync(dl_giveup, 0, sizeof(unsigned), st));
                char msg[200];
                snprintf(msg, sizeof(msg), "persistent decode layer: a hand-off wait timed out (code 0x%x); the turn's results are invalid (is another process using the GPU?)", code);
                REQUIRE(false, msg);
            }
            n = h_ctl->count;
            done = h_ctl->done != 0;
            REQUIRE(n >= 1 && n <= enq, "generation state out of range");
            for (int k = 0; k < n; ++k)
                REQUIRE(h_out_ids[k] >= 0 && h_out_ids[k] < V, "non-finite logits: the arg-max found no finite value (check the weights / fp8 scales)");
            if (done) break;
            REQUIRE(steps > 0, "sequence exceeds max_positions during decode");
        }
        for (int k = 0; k < n; ++k) out[k] = h_out_ids[k];
        e.kv_len = L + n - 1;             // EOS (or the last token) is appended to the ids but never fed
        {
            float t = 0.f;
            HIP_CHECK(hipEventElapsedTime(&t, ph_ev[2], ph_ev[3])); ph_ms[1] += t;
            if (vision_pending) { HIP_CHECK(hipEventElapsedTime(&t, ph_ev[0], ph_ev[1])); ph_ms[0] += t; vision_pending = false; }
            if (decoded) { HIP_CHECK(hipEventElapsedTime(&t, ph_ev[3], ph_ev[4])); ph_ms[2] += t; }
        }
        n_generated = n;
        *n_out = n;
    }

    // ------------------------------------------------------------------------------- taps
    // Parity taps of the single-env prefill (test infrastructure, off by default): the LAST row of the residual stream after every
    // decoder layer (error-versus-depth curves), and every row entering / leaving ONE probed layer (a fused layer checked against
    // the oracle at its own scale, on the engine's own inputs: a whole-stack tolerance would hide an O(1)-wrong stage).
    // probe buffers of the probed layer: 0 = x entering the layer, 1 = x leaving it, 2 = attention output (the A operand of o_proj),
    // 3 = x after the attention residual, 4 = post_attention_layernorm(x) (A of gate/up), 5 = silu(gate) * up (A of down_proj)
    // 6 = input_layernorm(x) (A of the q/k/v product), 7 = the q | k | v buffer after bias and RoPE (v columns: the plain product)
    static constexpr int N_PROBE = 8;
    bool layer_taps_on = false; int probe_layer = -1; int probe_rows = 0;
    T* layer_tap = nullptr; T* probe_buf[N_PROBE] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};
    int probe_cols(int which) const { return which == 2 ? nq * 128 : which == 5 ? I : which == 7 ? qkv_dim : H; }
    void set_layer_taps(int enable, int layer) override {
        REQUIRE(layer < c.layers, "probe layer out of range");
        HIP_CHECK(hipStreamSynchronize(st));
        layer_taps_on = enable != 0;
        probe_layer = layer_taps_on ? layer : -1;
        probe_rows = 0;
        if (layer_taps_on && !layer_tap) layer_tap = dalloc<T>((size_t)c.layers * H, true);
        if (probe_layer >= 0 && !probe_buf[0])
            for (int k = 0; k < N_PROBE; ++k) probe_buf[k] = dalloc<T>((size_t)c.max_positions * probe_cols(k));
    }
    void probe_copy(int which, const T* src, int M) {
        HIP_CHECK(hipMemcpyAsync(probe_buf[which], src, (size_t)M * probe_cols(which) * sizeof(T), hipMemcpyDeviceToDevice, st));
    }
    void get_layer_taps(float* out) override {
        REQUIRE(layer_tap != nullptr, "layer taps were never enabled");
        read_rows_f32(layer_tap, (size_t)c.layers * H, out);
    }
    void get_layer_probe(int which, float* out, int64_t max_elems, int32_t* n_rows, int32_t* n_cols) override {
        REQUIRE(probe_buf[0] != nullptr && which >= 0 && which < N_PROBE, "no such layer probe");
        const int cols = probe_cols(which);
        REQUIRE((int64_t)probe_rows * cols <= max_elems, "layer probe: output buffer too small");
        if (probe_rows > 0) read_rows_f32(probe_buf[which], (size_t)probe_rows * cols, out);
        *n_rows = probe_rows; *n_cols = cols;
    }
    void read_rows_f32(const T* src, size_t n, float* out) {
        float* tmp = nullptr;
        HIP_CHECK(hipMalloc((void**)&tmp, n * sizeof(float)));
        launch_to_f32<T>(st, src, tmp, (int64_t)n);
        HIP_CHECK(hipMemcpyAsync(out, tmp, n * sizeof(float), hipMemcpyDeviceToHost, st));
        HIP_CHECK(hipStreamSynchronize(st));
        HIP_CHECK(hipFree(tmp));
    }
    void get_hidden(float* out, int max_rows, int32_t* n_rows) override {
        int n = n_generated < HID_TAP_ROWS ? n_generated : HID_TAP_ROWS;
        if (n > max_rows) n = max_rows;
        read_rows_f32(hid_tap, (size_t)n * H, out);
        *n_rows = n;
    }
    void get_embeds(int env, int start, int n, float* out) override {
        Env& e = env_at(env);
        REQUIRE(start >= 0 && n >= 0 && start + n <= e.n_embeds, "embeds range");
        read_rows_f32(e.embeds + (size_t)start * H, (size_t)n * H, out);
    }
    void get_feats(int start, int n, float* out) override {
        REQUIRE(start >= 0 && n >= 0 && start + n <= n_feat_frames * otok, "feats range");
        read_rows_f32(feats + (size_t)start * H, (size_t)n * H, out);
    }
    void get_top2(float* out) override { out[0] = h_top2[0]; out[1] = h_top2[1]; }
    void sync() override { HIP_CHECK(hipStreamSynchronize(st)); }
    void set_graph(int enable) override { use_graph = enable != 0; if (!use_graph) drop_graphs(); }
    void ensure_prune_scratch() {
        if (prune_partial) return;
        const int nmax = c.max_frames * otok;
        prune_partial = dalloc<float>((size_t)((nmax + 63) / 64) * H);
        prune_mean = dalloc<float>(H);
        prune_score = dalloc<float>(nmax);
        d_sel = dalloc<int>(nmax);
        HIP_CHECK(hipHostMalloc((void**)&h_sel, nmax * sizeof(int)));
    }
    void set_memory_prune(int keep) override {
        REQUIRE(keep >= 0, "keep must be >= 0 (0 = the reference behaviour: all num_history x 196 memory tokens)");
        if (keep > 0) ensure_prune_scratch();
        prune_keep = keep;
    }
    // indices (ascending) of the `keep` rows of m [n_rows][H] (engine dtype, device) that survive; h_sel holds them on return
    void run_memory_prune(const T* m, int n_rows, int keep) {
        ensure_prune_scratch();
        REQUIRE(n_rows >= 1 && n_rows <= c.max_frames * otok && keep >= 1 && keep <= n_rows, "memory prune: bad sizes");
        launch_memory_prune<T>(st, m, n_rows, H, keep, prune_partial, prune_mean, prune_score, d_sel);
        HIP_CHECK(hipMemcpyAsync(h_sel, d_sel, keep * sizeof(int), hipMemcpyDeviceToHost, st));
        HIP_CHECK(hipStreamSynchronize(st));
    }
    void op_memory_prune(const void* m, int n_rows, int keep, int32_t* out_idx, float* out_score) override {
        run_memory_prune((const T*)m, n_rows, keep);
        for (int j = 0; j < keep; ++j) out_idx[j] = h_sel[j];
        if (out_score) HIP_CHECK(hipMemcpy(out_score, prune_score, n_rows * sizeof(float), hipMemcpyDeviceToHost));
    }
    // Opt-in (no reference counterpart): the single-env decode step and the lm_head read e4m3 copies of the LLM
    // weights (per-row scale) instead of the bf16 ones; prefill, vision and the lockstep multi-env path keep bf16.  Both copies stay
    // resident (15.2 + 7.6 GB of 288).  Quantised from the tensors loaded at the time of the first enable.
    void build_fp8_weights() {
        REQUIRE(sizeof(T) == 2, "fp8 weights need the bf16 engine");
        REQUIRE(weights_missing() == 0, g_err);
        REQUIRE(H % 16 == 0 && I % 16 == 0, "fp8 weights need hidden and intermediate sizes that are multiples of 16");
        if (fp8_built) return;
        auto build = [&](Q8& q, const T* w, int64_t rows, int cols) {
            q.q = dalloc<uint8_t>((size_t)rows * cols);
            q.s = dalloc<float>((size_t)rows);
            launch_quant_fp8_rows(st, w, cols, q.q, q.s, rows, cols);
        };
        for (auto& L : ll) {
            build(L.qkv8, L.qkv_w, qkv_dim, H);
            build(L.o8, L.o_w, H, nq * 128);
            build(L.gu8, L.gu_w, 2 * I, H);
            build(L.down8, L.down_w, H, I);
        }
        build(lm_head8, lm_head, V, H);
        HIP_CHECK(hipStreamSynchronize(st));
        fp8_built = true;
    }
    void set_fp8_decode(int enable) override {
        if (!enable) { if (fp8_on) drop_graphs(); fp8_on = false; return; }
        build_fp8_weights();
        if (!fp8_on) drop_graphs();
        fp8_on = true;
    }
    // Opt-in (no reference counterpart): the LLM's dense products with more than one row -- prefill (svln_generate,
    // the scheduler) and the decode steps of >= 4 lockstep envs -- run as e4m3 x e4m3 MFMA products (v_mfma_f32_32x32x16_fp8_fp8, fp32
    // accumulate, bf16 out): per-row weight scales (the copies of svln_set_fp8_decode), per-row activation scales computed on the fly.
    // Half the operand bytes per k through HBM / L2 / LDS.  Vision, attention, norms and the lm_head stay bf16.
    void set_fp8_gemm(int enable) override {
        if (!enable) { fp8_gemm_on = false; return; }
        build_fp8_weights();
        if (!act8) {
            const size_t widest = (size_t)(I > nq * 128 ? I : nq * 128);
            act8 = dalloc<uint8_t>((size_t)c.max_positions * (widest > (size_t)H ? widest : (size_t)H));
            act8_scale = dalloc<float>((size_t)c.max_positions);
        }
        fp8_gemm_on = true;
    }
    // C = A . W^T (+ epilogue) for an LLM linear: bf16 operands, or -- with svln_set_fp8_gemm -- the rows of A quantised on the fly
    // against the e4m3 copy of W
    // act8_src: the rows whose e4m3 copy act8 currently holds because the reduce that produced them also quantised them (the normalised
    // rows handed from o_proj to gate/up and from down_proj to the next layer's qkv: two of the four quantise launches of a layer)
    const void* act8_src = nullptr; int act8_rows = 0;
    bool llm_gemm(GemmArgs a, const Q8& q) {
        if (fp8_gemm_on && q.q) {
            if (!(act8_src == a.A && act8_rows == a.M && a.lda == a.K)) launch_quant_fp8_rows(st, a.A, a.lda, act8, act8_scale, a.M, a.K);
            act8_src = nullptr;
            if (a.norm_out && a.norm_w && !a.norm_b && (size_t)a.N <= (size_t)H) { a.norm_q8 = act8; a.norm_q8_scale = act8_scale; }
            const void* nout = a.norm_out; const int rows = a.M;
            a.A = act8; a.lda = a.K; a.W = q.q; a.ldw = a.K; a.a_scale = act8_scale; a.w_scale = q.s;
            const bool fused = launch_gemm<T>(st, a);
            if (fused && a.norm_q8) { act8_src = nout; act8_rows = rows; }      // the reduce ran after the product had consumed act8
            return fused;
        }
        act8_src = nullptr;
        return launch_gemm<T>(st, a);
    }
    bool op_gemm_fp8(const GemmArgs& a0) override {
        REQUIRE(sizeof(T) == 2, "fp8 products need the bf16 engine");
        GemmArgs a = a0; a.ws = gemm_ws; a.ws_elems = gemm_ws_elems; a.zeros = zero_line;
        REQUIRE(a.a_scale && a.w_scale, "missing scales");
        REQUIRE(a.K % 16 == 0 && a.lda % 16 == 0 && a.ldw % 16 == 0, "K and the row strides must be multiples of 16");
        REQUIRE(a.epi == EPI_NONE || a.epi == EPI_SWIGLU, "fp8 products: plain or SwiGLU epilogue");
        if (a.force_split > 1) {
            REQUIRE((size_t)a.force_split * a.M * a.N <= gemm_ws_elems, "force_split: S * M * N exceeds the split-K workspace");
            REQUIRE(a.N % 4 == 0 && !(a.epi == EPI_SWIGLU && a.N % 64 != 0), "force_split: N must be a multiple of 4 (64 with SwiGLU)");
        }
        const bool fused = launch_gemm<T>(st, a);
        sync();
        LAUNCH_CHECK("op_gemm_fp8");
        return fused;
    }
    void probe_reset() override {
        if (probe_ev.empty()) { probe_ev.resize(4096); for (auto& ev : probe_ev) HIP_CHECK(hipEventCreate(&ev)); }
        if (pprobe_ev.empty()) { pprobe_ev.resize(512); for (auto& ev : pprobe_ev) HIP_CHECK(hipEventCreate(&ev)); }
        probe_used = 0; probe_on = true; pprobe_used = 0; pprobe_rows = 0;
        probe_bytes = (double)2 * I * H * sizeof(T);
        // persistent layer: the four products one launch streams (o, gate/up, down, the next layer's q|k|v)
        if (persistent_active()) probe_bytes = ((double)H * nq * 128 + (double)2 * I * H + (double)H * I + (double)qkv_dim * H) * sizeof(T);
    }
    void probe_read(double* ms, int64_t* launches, double* bytes) override {
        HIP_CHECK(hipStreamSynchronize(st));
        double tot = 0;
        for (size_t k = 0; k + 1 < probe_used; k += 2) { float t = 0; HIP_CHECK(hipEventElapsedTime(&t, probe_ev[k], probe_ev[k + 1])); tot += t; }
        *ms = tot; *launches = (int64_t)(probe_used / 2); *bytes = probe_bytes;
        probe_on = false;
    }
    // steady-prefill gate/up products timed since probe_reset: total ms, count, mean rows, flops of one (2 * rows * 2I * H) and its weight bytes
    void probe_read_prefill(double* ms, int64_t* n, double* rows, double* flops, double* wbytes) override {
        HIP_CHECK(hipStreamSynchronize(st));
        double tot = 0;
        for (size_t k = 0; k + 1 < pprobe_used; k += 2) { float t = 0; HIP_CHECK(hipEventElapsedTime(&t, pprobe_ev[k], pprobe_ev[k + 1])); tot += t; }
        const int64_t cnt = (int64_t)(pprobe_used / 2);
        *ms = tot; *n = cnt; *rows = cnt ? pprobe_rows / cnt : 0;
        *flops = 2.0 * (*rows) * 2.0 * I * H; *wbytes = (double)2 * I * H * sizeof(T);
    }
    void phase_times(double* v, double* p, double* d, int reset) override {
        HIP_CHECK(hipStreamSynchronize(st));
        if (vision_pending) { float t = 0; HIP_CHECK(hipEventElapsedTime(&t, ph_ev[0], ph_ev[1])); ph_ms[0] += t; vision_pending = false; }
        *v = ph_ms[0]; *p = ph_ms[1]; *d = ph_ms[2];
        if (reset) ph_ms[0] = ph_ms[1] = ph_ms[2] = 0;
    }

    // ------------------------------------------------------------------------------- op-level entry points
    bool op_gemm(const GemmArgs& a0) override {
        GemmArgs a = a0; a.ws = gemm_ws; a.ws_elems = gemm_ws_elems; a.zeros = zero_line;
        REQUIRE(a.M >= 0 && a.N >= 0 && a.K >= 0, "negative GEMM extent");
        if (a.force_split > 1) {
            REQUIRE((size_t)a.force_split * a.M * a.N <= gemm_ws_elems, "force_split: S * M * N exceeds the split-K workspace");
            REQUIRE(a.N % 4 == 0 && !(a.epi == EPI_SWIGLU && a.N % 64 != 0), "force_split: N must be a multiple of 4 (64 with SwiGLU)");
        }
        const bool fused = launch_gemm<T>(st, a);
        sync();
        return fused;
    }
    void op_gemv(GemvArgs a, int32_t* host_token) override {
        REQUIRE(a.w8 == nullptr || sizeof(T) == 2, "fp8 weights need the bf16 engine");
        a.part_val = part_val; a.part_idx = part_idx;
        launch_gemv<T>(st, a);
        if (a.epi == EPI_ARGMAX) {
            launch_argmax_final(st, part_val, part_idx, gemv_grid(a.N), d_token, d_top2);
            const int t = read_token();
            if (host_token) *host_token = t;
        }
        sync();
    }
    void op_gemv_batched(GemvBatchArgs a, int32_t* host_tokens) override {
        REQUIRE(a.B == 1 || a.B == 2 || a.B == 4 || a.B == 8, "B must be 1, 2, 4 or 8");
        REQUIRE(a.K % Elt<T>::PER_CHUNK == 0 && a.N >= 1, "bad GEMV extents");
        REQUIRE(a.epi != EPI_SWIGLU || a.N % 64 == 0, "SwiGLU needs N % 64 == 0 (32-row gate / up blocks)");
        a.part_val = part_val_b; a.part_idx = part_idx_b;
        if (a.epi == EPI_ARGMAX && !a.norm_w) argmax_rows(a.W, a.ldw, (const T*)a.x, a.ldx, a.N, a.K, a.B, false);      // the engine's own routing (MFMA tiles from B = 4)
        else launch_gemv_batched<T>(st, a);
        if (a.epi == EPI_ARGMAX) {
            if (a.norm_w) launch_argmax_final_batched(st, part_val_b, part_idx_b, gemv_batched_grid(a.N, EPI_ARGMAX, a.B), a.B, d_tok_b);
            HIP_CHECK(hipMemcpyAsync(h_tok_b, d_tok_b, a.B * sizeof(int), hipMemcpyDeviceToHost, st));
            sync();
            if (host_tokens) for (int b = 0; b < a.B; ++b) host_tokens[b] = h_tok_b[b];
        }
        sync();
        LAUNCH_CHECK("op_gemv_batched");
    }
    void op_quant_fp8(const void* w, int64_t rows, int cols, void* w8, float* scale) override {
        REQUIRE(sizeof(T) == 2, "fp8 quantisation reads bf16 weights");
        REQUIRE(cols % 16 == 0, "cols must be a multiple of 16");
        launch_quant_fp8_rows(st, w, cols, w8, scale, rows, cols);
        sync();
    }
    void op_rmsnorm(const void* xi, const void* g, void* y, int rows, int n, float eps) override { launch_rmsnorm<T>(st, xi, g, y, rows, n, eps); sync(); }
    void op_layernorm(const void* xi, const void* g, const void* b, void* y, int rows, int n, float eps) override {
        launch_layernorm<T>(st, xi, g, b, y, rows, n, eps); sync();
    }
    // LLM attention on layer-0 pools / env 0: context rows (ctx_T positions from 0) are roped + appended first, then the T new rows
    void op_attention_llm(void* qkv_new, int ld, int Tn, int P, const void* ctx, int ctx_T, void* out, int o_stride, int nsplit) override {
        Env& e = env_at(0);
        REQUIRE(ctx_T == P, "context length must equal P");
        reset_env(0);
        ensure_pages(e, P + Tn);
        const LLayer& L = ll[0];
        RopeKvArgs r; r.Kpool = L.kpool; r.Vpool = L.vpool; r.page_table = e.d_pages; r.rope_tab = rope_tab; r.nq = nq; r.nkv = nkv; r.dyn_pos = nullptr;
        r.ld = ld;
        if (ctx_T > 0) { r.qkv = const_cast<void*>(ctx); r.T = ctx_T; r.P = 0; launch_rope_kv<T>(st, r); }
        r.qkv = qkv_new; r.T = Tn; r.P = P; launch_rope_kv<T>(st, r);
        AttnArgs a = llm_attn_args(L, e, qkv_new, ld, out, o_stride, Tn, P, P + Tn, false);
        if (nsplit > 1) {           // decode-style: one wave per kv head and key page
            REQUIRE(Tn * (nq / nkv) <= 32, "split-KV path takes at most 32 rows per kv head");
            a.nsplit = nsplit_max; a.tiles_per_split = tiles_per_split; a.rows_pad = 32;
            launch_attention<T>(st, a, 128, 1);
            launch_attention_combine<T>(st, a, 128);
        } else {                    // prefill: engine heuristic (may split the keys for few row blocks)
            launch_attention<T>(st, a, 128, 4);
            if (a.nsplit > 1) launch_attention_combine<T>(st, a, 128);
        }
        sync();
        reset_env(0);
    }
    void op_attention_vit(const void* qkv_buf, int ld, int F, void* out, int o_stride) override {
        REQUIRE(F >= 1 && F <= c.max_frames, "frames");
        vit_attention(qkv_buf, ld, F, out, o_stride); sync();
    }
    void op_pool(const void* in, void* out, int F) override { launch_pool<T>(st, in, out, tap_idx, tap_w, F, side, oside, H); sync(); }
    void op_patchify(const float* p, void* out, int F) override { launch_patchify<T>(st, p, out, F, c.v_image, c.v_patch, kp); sync(); }
};

}  // namespace svln

// =================================================================================== C ABI
using namespace svln;
struct svln_engine { EngineBase* impl; };

#define API_BEGIN try {
#define API_BEGIN_H try { REQUIRE(h && h->impl, "null engine handle"); DeviceGuard guard_(h->impl->device_id());
#define API_END                                   \
    return 0;                                     \
    }                                             \
    catch (const std::exception& ex) {            \
        g_err = ex.what();                        \
        return -1;                                \
    }

extern "C" {

const char* svln_last_error(void) { return g_err.c_str(); }

int svln_create(const svln_config* cfg, int device, svln_engine** out) {
    API_BEGIN
    REQUIRE(cfg && out, "null argument");
    int n = 0;
    HIP_CHECK(hipGetDeviceCount(&n));
    REQUIRE(n > 0 && device < n, "no such HIP device");
    svln_engine* h = new svln_engine;
    if (cfg->dtype == SVLN_F32) h->impl = new Engine<float>(*cfg, device);
    else h->impl = new Engine<bf16>(*cfg, device);
    *out = h;
    API_END
}
void svln_destroy(svln_engine* h) { if (h) { delete h->impl; delete h; } }       /* ~Engine switches to its device and back */
int svln_sync(svln_engine* h) { API_BEGIN_H h->impl->sync(); API_END }
int svln_synth_tensor(svln_engine* h, const char* name, uint64_t seed_t, float hw, float base) { API_BEGIN_H h->impl->synth_tensor(name, seed_t, hw, base); API_END }
int svln_set_tensor(svln_engine* h, const char* name, const void* data, int dtype, int64_t numel, int on_device) {
    API_BEGIN_H h->impl->set_tensor(name, data, dtype, numel, on_device); API_END
}
int svln_weights_ready(svln_engine* h) { API_BEGIN_H if (h->impl->weights_missing()) return -2; API_END }
int svln_get_tensor_f32(svln_engine* h, const char* name, float* out, int64_t numel) { API_BEGIN_H h->impl->get_tensor_f32(name, out, numel); API_END }
int svln_reset_env(svln_engine* h, int env) { API_BEGIN_H h->impl->reset_env(env); API_END }
int svln_kv_reset(svln_engine* h, int env) { API_BEGIN_H h->impl->kv_reset(env); API_END }
int svln_env_state(svln_engine* h, int env, int32_t* n_embeds, int32_t* kv_len) { API_BEGIN_H h->impl->env_state(env, n_embeds, kv_len); API_END }
int svln_encode_frames(svln_engine* h, const float* pixels, int n_frames, int on_device) { API_BEGIN_H h->impl->encode_frames(pixels, n_frames, on_device); API_END }
int svln_preprocess_frames(svln_engine* h, const uint8_t* rgb, int n_frames, int height, int width, int on_device, float* out_dev) {
    API_BEGIN_H h->impl->preprocess_frames(rgb, n_frames, height, width, on_device, out_dev, true); API_END
}
int svln_preprocess_frames_enqueue(svln_engine* h, const uint8_t* rgb, int n_frames, int height, int width, int on_device, float* out_dev) {
    API_BEGIN_H h->impl->preprocess_frames(rgb, n_frames, height, width, on_device, out_dev, false); API_END
}
int svln_frame_ring(svln_engine* h, int slots, int height, int width, uint8_t** host_base, int64_t* slot_stride) {
    API_BEGIN_H h->impl->frame_ring(slots, height, width, host_base, slot_stride); API_END
}
int svln_frame_ring_wait(svln_engine* h, int slot) { API_BEGIN_H h->impl->frame_ring_wait(slot); API_END }
int svln_turn(svln_engine* h, const svln_turn_args* a, int64_t* out_ids, int out_cap, int32_t* n_out, int32_t* kv_len) {
    API_BEGIN_H REQUIRE(a, "null argument"); h->impl->turn(*a, out_ids, out_cap, n_out, kv_len); API_END
}
int svln_engine_stream(svln_engine* h, void** stream) { API_BEGIN_H REQUIRE(stream, "null output pointer"); *stream = h->impl->stream_handle(); API_END }
int svln_preprocess_time(svln_engine* h, double* gpu_ms, int64_t* frames, int reset) { API_BEGIN_H h->impl->preprocess_time(gpu_ms, frames, reset); API_END }
int svln_append_turn(svln_engine* h, int env, const int64_t* ids, int n_ids, int n_memory) { API_BEGIN_H h->impl->append_turn(env, ids, n_ids, 0, n_memory); API_END }
int svln_append_turn_at(svln_engine* h, int env, const int64_t* ids, int n_ids, int frame_base, int n_memory) {
    API_BEGIN_H h->impl->append_turn(env, ids, n_ids, frame_base, n_memory); API_END
}
int svln_generate_batch(svln_engine* h, const int32_t* envs, int n_envs, int max_new, const int64_t* eos, int n_eos, int64_t* out, int cap,
                        int32_t* n_out) {
    API_BEGIN_H h->impl->generate_batch(envs, n_envs, max_new, eos, n_eos, out, cap, n_out); API_END
}
int svln_batch_submit(svln_engine* h, int env, int max_new_tokens, const int64_t* eos_ids, int n_eos, int32_t* slot) {
    API_BEGIN_H
    REQUIRE(slot, "null slot pointer");
    *slot = h->impl->batch_submit(env, max_new_tokens, eos_ids, n_eos);
    API_END
}
int svln_batch_step(svln_engine* h, int32_t* running, int32_t* finished_slots, int32_t* n_finished) {
    API_BEGIN_H
    REQUIRE(running && finished_slots && n_finished, "null output pointer");
    *running = h->impl->batch_step(finished_slots, n_finished);
    API_END
}
int svln_batch_result(svln_engine* h, int slot, int32_t* env, int64_t* out_ids, int out_cap, int32_t* n_out) {
    API_BEGIN_H h->impl->batch_result(slot, env, out_ids, out_cap, n_out); API_END
}
int svln_batch_cancel(svln_engine* h, int slot) { API_BEGIN_H h->impl->batch_cancel(slot); API_END }
int svln_set_turn_row_limit(svln_engine* h, int rows) { API_BEGIN_H h->impl->set_turn_row_limit(rows); API_END }
int svln_set_repetition_penalty(svln_engine* h, float penalty) { API_BEGIN_H h->impl->set_repetition_penalty(penalty); API_END }
int svln_get_hidden_batch(svln_engine* h, int slot, float* out, int max_rows, int32_t* n_rows) { API_BEGIN_H h->impl->get_hidden_batch(slot, out, max_rows, n_rows); API_END }
int svln_generate(svln_engine* h, int env, int max_new, const int64_t* eos, int n_eos, int64_t* out, int cap, int32_t* n_out) {
    API_BEGIN_H h->impl->generate(env, max_new, eos, n_eos, out, cap, n_out, false); API_END
}
int svln_generate_fixed(svln_engine* h, int env, int n_tokens, int64_t* out) {
    API_BEGIN_H int32_t n = 0; h->impl->generate(env, n_tokens, nullptr, 0, out, n_tokens, &n, true); API_END
}
int svln_get_hidden(svln_engine* h, float* out, int max_rows, int32_t* n_rows) { API_BEGIN_H h->impl->get_hidden(out, max_rows, n_rows); API_END }
int svln_set_layer_taps(svln_engine* h, int enable, int probe_layer) { API_BEGIN_H h->impl->set_layer_taps(enable, probe_layer); API_END }
int svln_get_layer_taps(svln_engine* h, float* out) { API_BEGIN_H REQUIRE(out, "null output pointer"); h->impl->get_layer_taps(out); API_END }
int svln_get_layer_probe(svln_engine* h, int which, float* out, int64_t max_elems, int32_t* n_rows, int32_t* n_cols) {
    API_BEGIN_H REQUIRE(out && n_rows && n_cols, "null output pointer"); h->impl->get_layer_probe(which, out, max_elems, n_rows, n_cols); API_END
}
int svln_get_embeds(svln_engine* h, int env, int start, int n, float* out) { API_BEGIN_H h->impl->get_embeds(env, start, n, out); API_END }
int svln_get_frame_feats(svln_engine* h, int start, int n, float* out) { API_BEGIN_H h->impl->get_feats(start, n, out); API_END }
int svln_get_top2(svln_engine* h, float* out) { API_BEGIN_H h->impl->get_top2(out); API_END }
int svln_set_decode_graph(svln_engine* h, int enable) { API_BEGIN_H h->impl->set_graph(enable); API_END }
int svln_set_decode_persistent(svln_engine* h, int enable) { API_BEGIN_H h->impl->set_decode_persistent(enable); API_END }
int svln_probe_decode_layer(svln_engine* h, int layer, unsigned long long* out, int max_wgs, int32_t* n_wgs) {
    API_BEGIN_H REQUIRE(out && n_wgs, "null output pointer"); h->impl->probe_decode_layer(layer, out, max_wgs, n_wgs); API_END
}
int svln_set_fp8_decode(svln_engine* h, int enable) { API_BEGIN_H h->impl->set_fp8_decode(enable); API_END }
int svln_set_fp8_gemm(svln_engine* h, int enable) { API_BEGIN_H h->impl->set_fp8_gemm(enable); API_END }
int svln_set_memory_prune(svln_engine* h, int keep_tokens) { API_BEGIN_H h->impl->set_memory_prune(keep_tokens); API_END }
int svln_op_memory_prune(svln_engine* h, const void* mem, int n_rows, int keep, int32_t* out_idx, float* out_score) {
    API_BEGIN_H h->impl->op_memory_prune(mem, n_rows, keep, out_idx, out_score); API_END
}
int svln_probe_reset(svln_engine* h) { API_BEGIN_H h->impl->probe_reset(); API_END }
int svln_probe_read(svln_engine* h, double* ms, int64_t* launches, double* bytes) { API_BEGIN_H h->impl->probe_read(ms, launches, bytes); API_END }
int svln_probe_read_prefill(svln_engine* h, double* ms, int64_t* n, double* rows, double* flops, double* wbytes) {
    API_BEGIN_H h->impl->probe_read_prefill(ms, n, rows, flops, wbytes); API_END
}
int svln_phase_times(svln_engine* h, double* v, double* p, double* d, int reset) { API_BEGIN_H h->impl->phase_times(v, p, d, reset); API_END }
int svln_set_feature_cache(svln_engine* h, int capacity_frames) { API_BEGIN_H h->impl->set_feature_cache(capacity_frames); API_END }
int svln_feature_cache_stats(svln_engine* h, int64_t* hits, int64_t* misses) { API_BEGIN_H h->impl->feature_cache_stats(hits, misses); API_END }

int svln_op_gemm(svln_engine* h, const void* A, int lda, const void* W, int ldw, void* C, int ldc, const void* bias, const void* res, int ldr,
                 int res_mod, int M, int N, int K, int epi, int force_cfg, int force_split) {
    API_BEGIN_H
    GemmArgs a; std::memset(&a, 0, sizeof(a));
    a.A = A; a.lda = lda; a.W = W; a.ldw = ldw; a.C = C; a.ldc = ldc; a.bias = bias; a.res = res; a.ldr = ldr; a.res_mod = res_mod;
    a.M = M; a.N = N; a.K = K; a.epi = epi; a.nsplit = 1; a.force_cfg = force_cfg; a.force_split = force_split;
    h->impl->op_gemm(a);
    API_END
}
int svln_op_gemm_norm(svln_engine* h, const void* A, int lda, const void* W, int ldw, void* C, int ldc, const void* bias, const void* res, int ldr,
                      const void* norm_w, const void* norm_b, void* norm_out, float eps, int M, int N, int K, int force_split, int* fused) {
    API_BEGIN_H
    GemmArgs a; std::memset(&a, 0, sizeof(a));
    a.A = A; a.lda = lda; a.W = W; a.ldw = ldw; a.C = C; a.ldc = ldc; a.bias = bias; a.res = res; a.ldr = ldr; a.norm_b = norm_b;
    a.M = M; a.N = N; a.K = K; a.epi = EPI_NONE; a.nsplit = 1; a.force_split = force_split; a.norm_w = norm_w; a.norm_out = norm_out; a.norm_eps = eps;
    const bool f = h->impl->op_gemm(a);
    if (fused) *fused = f ? 1 : 0;
    API_END
}
int svln_op_gemm_norm_q8(svln_engine* h, const void* A, int lda, const void* W, int ldw, void* C, int ldc, const void* res, int ldr,
                         const void* norm_w, void* norm_out, float eps, int M, int N, int K, int force_split, void* q8, float* q8_scale, int* fused) {
    API_BEGIN_H
    GemmArgs a; std::memset(&a, 0, sizeof(a));
    a.A = A; a.lda = lda; a.W = W; a.ldw = ldw; a.C = C; a.ldc = ldc; a.res = res; a.ldr = ldr;
    a.M = M; a.N = N; a.K = K; a.epi = EPI_NONE; a.nsplit = 1; a.force_split = force_split; a.norm_w = norm_w; a.norm_out = norm_out; a.norm_eps = eps;
    a.norm_q8 = q8; a.norm_q8_scale = q8_scale; a.pen = 1.0f;
    const bool f = h->impl->op_gemm(a);
    if (fused) *fused = f ? 1 : 0;
    API_END
}
int svln_op_gemv(svln_engine* h, const void* W, int ldw, const void* x, const void* norm_w, float eps, const void* bias, const void* res, void* y,
                 int N, int K, int epi, int32_t* host_token) {
    API_BEGIN_H
    GemvArgs a; a.W = W; a.ldw = ldw; a.x = x; a.norm_w = norm_w; a.eps = eps; a.bias = bias; a.res = res; a.y = y; a.N = N; a.K = K; a.epi = epi;
    a.part_val = nullptr; a.part_idx = nullptr; a.w8 = nullptr; a.scale = nullptr; a.skip = nullptr;
    h->impl->op_gemv(a, host_token);
    API_END
}
int svln_op_gemm_fp8(svln_engine* h, const void* A8, const float* a_scale, int lda, const void* W8, const float* w_scale, int ldw, void* C, int ldc,
                     const void* bias, const void* res, int ldr, int M, int N, int K, int epi, int force_cfg, int force_split) {
    API_BEGIN_H
    GemmArgs a; std::memset(&a, 0, sizeof(a));
    a.A = A8; a.lda = lda; a.W = W8; a.ldw = ldw; a.C = C; a.ldc = ldc; a.bias = bias; a.res = res; a.ldr = ldr; a.a_scale = a_scale; a.w_scale = w_scale;
    a.M = M; a.N = N; a.K = K; a.epi = epi; a.nsplit = 1; a.force_cfg = force_cfg; a.force_split = force_split;
    h->impl->op_gemm_fp8(a);
    API_END
}
int svln_op_gemv_batched(svln_engine* h, const void* W, int ldw, const void* x, int ldx, const void* norm_w, float eps, const void* bias,
                         const void* res, int ldr, void* y, int ldy, int N, int K, int epi, int B, int32_t* host_tokens) {
    API_BEGIN_H
    GemvBatchArgs a; a.W = W; a.ldw = ldw; a.x = x; a.ldx = ldx; a.norm_w = norm_w; a.eps = eps; a.bias = bias; a.res = res; a.ldr = ldr;
    a.y = y; a.ldy = ldy; a.N = N; a.K = K; a.epi = epi; a.B = B; a.part_val = nullptr; a.part_idx = nullptr;
    h->impl->op_gemv_batched(a, host_tokens);
    API_END
}
int svln_op_quant_fp8(svln_engine* h, const void* w_bf16, int64_t rows, int cols, void* w8, float* scale) {
    API_BEGIN_H h->impl->op_quant_fp8(w_bf16, rows, cols, w8, scale); API_END
}
int svln_op_gemv_fp8(svln_engine* h, const void* w8, const float* scale, int ldw, const void* x, const void* norm_w, float eps, const void* bias,
                     const void* res, void* y, int N, int K, int epi, int32_t* host_token) {
    API_BEGIN_H
    if (K % 16 != 0) throw std::runtime_error("K must be a multiple of 16");
    GemvArgs a; a.W = nullptr; a.ldw = ldw; a.x = x; a.norm_w = norm_w; a.eps = eps; a.bias = bias; a.res = res; a.y = y; a.N = N; a.K = K; a.epi = epi;
    a.part_val = nullptr; a.part_idx = nullptr; a.w8 = w8; a.scale = scale; a.skip = nullptr;
    h->impl->op_gemv(a, host_token);
    API_END
}
int svln_op_rmsnorm(svln_engine* h, const void* x, const void* g, void* y, int rows, int n, float eps) { API_BEGIN_H h->impl->op_rmsnorm(x, g, y, rows, n, eps); API_END }
int svln_op_layernorm(svln_engine* h, const void* x, const void* g, const void* b, void* y, int rows, int n, float eps) {
    API_BEGIN_H h->impl->op_layernorm(x, g, b, y, rows, n, eps); API_END
}
int svln_op_attention_llm(svln_engine* h, void* qkv, int ld, int T, int P, const void* ctx, int ctx_T, void* out, int o_stride, int nsplit) {
    API_BEGIN_H h->impl->op_attention_llm(qkv, ld, T, P, ctx, ctx_T, out, o_stride, nsplit); API_END
}
int svln_op_attention_vit(svln_engine* h, const void* qkv, int ld, int F, void* out, int o_stride) { API_BEGIN_H h->impl->op_attention_vit(qkv, ld, F, out, o_stride); API_END }
int svln_op_pool(svln_engine* h, const void* in, void* out, int F) { API_BEGIN_H h->impl->op_pool(in, out, F); API_END }
int svln_op_patchify(svln_engine* h, const float* pix, void* out, int F) { API_BEGIN_H h->impl->op_patchify(pix, out, F); API_END }

}  // extern "C"
