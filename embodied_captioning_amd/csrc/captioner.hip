// Host side of libcaptioner_hip.so: weight registry, arena, the encoder / decoder launch sequences and the C ABI
// declared in include/captioner_hip.h.  Everything on the data path is a kernel launch on the caller's stream; the
// launch sequences never allocate or synchronise (so they can be captured into a hipGraph).
#include <stdarg.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <algorithm>
#include <atomic>
#include <map>
#include <string>
#include <set>
#include <mutex>
#include <vector>

#include "../../include/captioner_hip.h"
#include "gemm.h"
#include "ops.h"
#include "decode_small.h"

// ------------------------------------------------------------------------------------------------ errors
static thread_local char g_err[2048] = "";
void cap_set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

int cap_kernel_setup(const void* kernel, int lds_bytes, int* n_cu) {
    static std::mutex mu;
    static std::set<std::pair<const void*, int>> done;
    static std::map<int, int> cus;
    int dev = 0;
    CAP_HIP_CHECK(hipGetDevice(&dev));
    std::lock_guard<std::mutex> lock(mu);
    if (!done.count({kernel, dev})) {
        if (lds_bytes > 64 * 1024)
            CAP_HIP_CHECK(hipFuncSetAttribute(kernel, hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes));
        done.insert({kernel, dev});
    }
    if (n_cu) {
        auto it = cus.find(dev);
        if (it == cus.end()) {
            int n = 0;
            CAP_HIP_CHECK(hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev));
            it = cus.emplace(dev, n).first;
        }
        *n_cu = it->second;
    }
    return 0;
}

namespace {

struct Slot {            // where one checkpoint tensor (or a row range of a fused tensor) lives
    void* dst = nullptr;
    int dtype = CAP_DT_F32;   // storage type at dst: compute dtype or fp32
    int64_t rows = 0, cols = 0;
    int dst_ld = 0;
    void* aux = nullptr;      // CAP_DT_I8W: the rows' fp32 scales
    bool loaded = false;
};

struct ProfTag { std::string tag; double flops, bytes; hipEvent_t e0, e1; };

// The weights of one model on one GPU: device buffers in allocation order + the name -> buffer registry.  Read-only once
// loaded, so every handle of an EnginePool points at the same store (cap_create_shared): a pool of n engines costs one copy
// of the weights plus n arenas.  Freed when the last handle that references it is destroyed.
struct WeightStore {
    std::vector<void*> ptrs;
    std::vector<size_t> sizes;
    std::multimap<std::string, Slot> slots;
    size_t bytes = 0;
    std::atomic<int> refs{1};
    int device = 0;
};

struct VLayer {
    void *w_qkv, *w_proj, *w_fc1, *w_fc2;
    float *b_qkv, *b_proj, *b_fc1, *b_fc2, *ln1_g, *ln1_b, *ln2_g, *ln2_b;
};
struct QLayer {            // BLIP-2 Q-Former layer (queries only): self-attention, optional cross-attention, query FFN; post-LN
    bool cross = false;
    void *w_qkv = nullptr, *w_so = nullptr, *w_cq = nullptr, *w_ckv = nullptr, *w_co = nullptr, *w_f1 = nullptr, *w_f2 = nullptr;
    float *b_qkv = nullptr, *b_so = nullptr, *so_g = nullptr, *so_b = nullptr, *b_cq = nullptr, *b_ckv = nullptr, *b_co = nullptr,
          *co_g = nullptr, *co_b = nullptr, *b_f1 = nullptr, *b_f2 = nullptr, *f_g = nullptr, *f_b = nullptr;
};
// int8 weights: up to this many crops per call the prompt pass (crops x 33 rows) runs on the weight-streaming kernels like a decode
// step; beyond, it is a GEMM proper and goes to the tiled kernels (run_opt).  Within each range a crop's bits do not depend on the
// batch it is in; across the two the prompt's sums are formed in a different order (fp32-rounding-level differences before the bf16
// roundings).
constexpr int kI8SkinnyPromptCrops = 4;

struct OLayer {            // OPT decoder layer (pre-LN): fused q|k|v, out_proj, fc1 (ReLU), fc2; K/V caches [B][Lmax][T]
    void *w_qkv, *w_o, *w_f1, *w_f2, *kc, *vc;
    float *s_qkv = nullptr, *s_o = nullptr, *s_f1 = nullptr, *s_f2 = nullptr;   // CapConfig.weight_int8: row scales of the int8 weights
    float *b_qkv, *b_o, *b_f1, *b_f2, *ln1_g, *ln1_b, *ln2_g, *ln2_b;
};
struct TLayer {
    void *w_qkv, *w_so, *w_cq, *w_co, *w_f1, *w_f2;
    float *b_qkv, *b_so, *so_g, *so_b, *b_cq, *b_co, *co_g, *co_b, *b_f1, *b_f2, *f_g, *f_b;
    void* self_cache;     // [2][R][H][max_len][64] (T)
};

struct CBlock {            // one CoCa text block: causal self-attention (unimodal / multimodal) or cross-attention
    bool cross = false;
    int cache = -1;        // self blocks: index of their K/V cache
    int cross_idx = -1;    // cross blocks: multimodal layer index (cross K/V cache slot)
    void *w_in = nullptr, *w_o = nullptr, *w_fc = nullptr, *w_pr = nullptr;   // w_in: [3E,E] self, [E,E] cross query
    float *b_in = nullptr, *b_o = nullptr, *b_fc = nullptr, *b_pr = nullptr;
    float *ln1_g = nullptr, *ln1_b = nullptr, *ln2_g = nullptr, *ln2_b = nullptr;
};

constexpr bool kDeltaInT = false;
// The ViT branch GEMMs (proj, fc2) of the MFMA-staged types add their output to the residual stream X IN PLACE (gemm_pp.hip's
// residual epilogue: C = acc + bias + C) and the next LayerNorm reads X once.  The older scheme - branch output to `delta`, the
// add+LayerNorm kernel reads delta and X and writes X back - moves 5 x M x D x 4 bytes per (GEMM, LayerNorm) pair against 4 here,
// and stays for the exact fp32 mode, whose stream kernels have no residual operand.  Same fp32 add of the same two operands: the
// residual stream has the same bits either way.
inline bool vit_adds_in_place(int gdt) { return !kDeltaInT && gdt != CAP_DT_F32; }

struct Captioner {
    CapConfig c;
    int dt; size_t esz;          // storage type of activations / K-V caches that kernels other than the GEMMs read
    int gdt;                     // type of every GEMM operand (A and W): == dt, except CAP_F32_SPLIT: dt = fp32, gdt = G8
                                 // (split fp16, common.h) - there every kernel whose output feeds a GEMM writes G8
    int NT, P, Kpatch, Kpad;
    bool kv16 = false;           // split mode: the cross-attention K/V cache is KV16 (int16 + one scale per head row, common.h) instead of fp32
    size_t kvrow = 0;            // bytes of one 64-wide head row of that cache (KV16: 132, amortised)
    // bytes of one (layer, k | v) block of the cross cache holding `rows` head rows
    size_t cross_block(size_t rows) const { return kv16 ? kv16_block_bytes(rows) : rows * kvrow; }
    size_t dev_bytes = 0;        // arena (+ the weights when this handle created the store)
    std::vector<void*> allocs;   // arena: owned by this handle
    WeightStore* ws = nullptr;   // weights: shared
    bool replay = false;         // building a handle on an existing store: walloc hands out the store's buffers in order
    size_t wcur = 0;
    float* stage = nullptr; size_t stage_elems = 0;
    unsigned int* absmax_dev = nullptr;   // cap_load_weight: max |w| of a tensor bound for a G8 slot (range check)
    // early exit of the decode loop (cap_set_early_exit): poll every `poll` steps through a host-mapped word
    int poll = 0; int* host_flag = nullptr; int* host_flag_dev = nullptr;
    int last_steps = 0;          // decode steps the last cap_generate ran (cap_last_decode_steps)
    int decode_path = 0;         // cap_set_decode_path: 0 = by row count (<= SMALL_MAX_ROWS rows: the fused small-batch kernels),
                                 // 1 = always the batch kernels, 2 = always the small-batch kernels (an error beyond their row limit)
    int last_path = 0;           // what the last cap_generate's decode steps ran on (cap_last_decode_path): 1 batch, 2 small-batch
    int compaction = 1;          // cap_set_row_compaction: 1 = the greedy batch path works on the open captions' rows only (RowMap)
    int last_compacted = 0;      // did the last cap_generate's decode loop run compacted (cap_last_row_compaction)
    int *live = nullptr, *n_live = nullptr;       // RowMap storage: int32 [max rows] + the count
    // vision weights
    float *cls, *vpos, *b_patch, *post_g, *post_b;
    void* w_patch;
    std::vector<VLayer> vl;
    // text weights
    float *word_f32, *tpos, *emb_g, *emb_b, *b_ckv, *b_tr, *tr_g, *tr_b, *b_vocab;
    void *word_t, *w_ckv, *w_tr;
    std::vector<TLayer> tl;
    // arena
    void *patches, *ln, *qkv, *ctx, *mlp, *emb_t, *cross;
    float *X, *emb_f;
    void* delta;                 // ViT branch output (proj / fc2), folded into X by the next add+LayerNorm; fp32, or the
                                 // compute type when kDeltaInT (measured: -0.6 ms per 256 frames, but 77 % instead of 80 %
                                 // of bf16 captions token-identical to the fp32 mode's - not worth it, so off)
    int *seq, *finished, *lens, *anc;
    float *dx, *dy, *logits, *dpart;
    float* dx2 = nullptr;        // fused decode paths: second fp32 LayerNorm row buffer (ping-pong with dx), as many rows as dx
    void *dx_t, *dq, *dctx, *dh;
    void* beam = nullptr;
    size_t cache_layer_bytes = 0;
    // ---- CoCa (CAP_ARCH_COCA)
    int Q = 0, E = 0;
    float *ln_pre_g = nullptr, *ln_pre_b = nullptr, *lnk_g = nullptr, *lnk_b = nullptr, *lnpost_g = nullptr,
          *lnpost_b = nullptr, *pool_q = nullptr, *b_pool_kv = nullptr, *b_pool_o = nullptr, *ones = nullptr,
          *zeros = nullptr, *tok_emb = nullptr, *lnf_g = nullptr, *lnf_b = nullptr, *pool_o = nullptr,
          *img_tokens = nullptr;
    void *w_pool_kv = nullptr, *w_pool_o = nullptr, *w_cvocab = nullptr, *pool_kvbuf = nullptr, *pool_ctx = nullptr,
         *xhat = nullptr;
    std::vector<CBlock> cb;
    std::vector<void*> ccache;
    int ldl;
    // ---- BLIP-2 (CAP_ARCH_BLIP2)
    std::vector<QLayer> ql;
    std::vector<OLayer> ol;
    bool wq8 = false;            // CapConfig.weight_int8: the OPT decoder's Linear weights are row-quantised int8 (gemm_skinny.hip)
    void* w8_scratch = nullptr;  // one weight matrix as row-major bf16 integers: the prompt pass of more than kI8SkinnyPromptCrops crops
    float *q_x0 = nullptr, *b_lproj = nullptr, *o_tok = nullptr, *o_pos = nullptr, *o_lnf_g = nullptr, *o_lnf_b = nullptr;
    void *w_lproj = nullptr, *o_tok_t = nullptr;
    float *qx = nullptr, *qy = nullptr, *lm_proj = nullptr, *ox = nullptr;      // activations
    void *qx_t = nullptr, *qqkv = nullptr, *qctx = nullptr, *qh = nullptr, *qkvimg = nullptr, *oh_t = nullptr, *oqkv = nullptr,
         *octx = nullptr, *off = nullptr;
    // ---- sentence encoder (CAP_ARCH_MINILM): token-type row 0, activations [max_batch * max_len, .]
    float *tok_type = nullptr, *te_x = nullptr, *te_y = nullptr;
    void *te_xt = nullptr, *te_qkv = nullptr, *te_ctx = nullptr, *te_h = nullptr;
    // profiling
    bool prof = false;
    std::vector<ProfTag> prof_recs;
};

int dev_alloc(Captioner* m, void** p, size_t bytes) {
    bytes = (bytes + 255) & ~(size_t)255;
    if (bytes == 0) bytes = 256;
    CAP_HIP_CHECK(hipMalloc(p, bytes));
    m->allocs.push_back(*p);
    m->dev_bytes += bytes;
    return 0;
}

#define TRY(x) do { if ((x) != 0) return -1; } while (0)

// weight buffer: a new allocation recorded in the store, or (replay) the store's next buffer - the build_* functions run
// the same sequence of calls for the same architecture, which the size check enforces
int walloc(Captioner* m, void** p, size_t bytes) {
    bytes = (bytes + 255) & ~(size_t)255;
    if (bytes == 0) bytes = 256;
    WeightStore* ws = m->ws;
    if (m->replay) {
        if (m->wcur >= ws->ptrs.size() || ws->sizes[m->wcur] != bytes) {
            cap_set_error("cap_create_shared: the configuration does not describe the model whose weights are shared "
                          "(buffer %zu: %zu bytes wanted)", m->wcur, bytes);
            return -1;
        }
        *p = ws->ptrs[m->wcur++];
        return 0;
    }
    CAP_HIP_CHECK(hipMalloc(p, bytes));
    ws->ptrs.push_back(*p);
    ws->sizes.push_back(bytes);
    ws->bytes += bytes;
    m->dev_bytes += bytes;
    return 0;
}

int add_slot(Captioner* m, const std::string& name, void* dst, int dtype, int64_t rows, int64_t cols, int dst_ld = 0, void* aux = nullptr) {
    if (m->replay) return 0;
    Slot s; s.dst = dst; s.dtype = dtype; s.rows = rows; s.cols = cols; s.dst_ld = dst_ld ? dst_ld : (int)cols; s.aux = aux;
    m->ws->slots.insert({name, s});
    return 0;
}

// allocate a fp32 vector and register it
int reg_f32(Captioner* m, const std::string& name, float** p, int64_t n) {
    TRY(walloc(m, (void**)p, n * 4));
    return add_slot(m, name, *p, CAP_DT_F32, 1, n);
}
// allocate a compute-dtype matrix [rows, ld] and register it
int reg_mat(Captioner* m, const std::string& name, void** p, int64_t rows, int64_t cols, int ld = 0) {
    if (!ld) ld = (int)cols;
    TRY(walloc(m, p, (size_t)rows * ld * m->esz));
    if (ld != cols && !m->replay) CAP_HIP_CHECK(hipMemset(*p, 0, (size_t)rows * ld * m->esz));
    return add_slot(m, name, *p, m->gdt, rows, cols, ld);
}

// allocate an int8 weight [rows, cols] in fragment order + its row scales and register it (CapConfig.weight_int8)
int reg_mat_i8(Captioner* m, const std::string& name, void** p, float** scale, int64_t rows, int64_t cols) {
    TRY(walloc(m, p, (size_t)rows * cols));
    TRY(walloc(m, (void**)scale, (size_t)rows * 4));
    return add_slot(m, name, *p, CAP_DT_I8W, rows, cols, 0, *scale);
}

int build_blip(Captioner* m) {
    const CapConfig& c = m->c;
    const int D = c.v_hidden, Mv = c.v_mlp, T = c.t_hidden, F = c.t_ffn, V = c.vocab;
    const std::string vm = "vision_model.";
    TRY(reg_f32(m, vm + "embeddings.class_embedding", &m->cls, D));
    TRY(reg_f32(m, vm + "embeddings.position_embedding", &m->vpos, (int64_t)m->NT * D));
    TRY(reg_mat(m, vm + "embeddings.patch_embedding.weight", &m->w_patch, D, m->Kpatch, m->Kpad));
    TRY(reg_f32(m, vm + "embeddings.patch_embedding.bias", &m->b_patch, D));
    m->vl.resize(c.v_layers);
    for (int i = 0; i < c.v_layers; ++i) {
        VLayer& L = m->vl[i];
        const std::string p = vm + "encoder.layers." + std::to_string(i) + ".";
        TRY(reg_mat(m, p + "self_attn.qkv.weight", &L.w_qkv, 3 * D, D));
        TRY(reg_f32(m, p + "self_attn.qkv.bias", &L.b_qkv, 3 * D));
        TRY(reg_mat(m, p + "self_attn.projection.weight", &L.w_proj, D, D));
        TRY(reg_f32(m, p + "self_attn.projection.bias", &L.b_proj, D));
        TRY(reg_f32(m, p + "layer_norm1.weight", &L.ln1_g, D));
        TRY(reg_f32(m, p + "layer_norm1.bias", &L.ln1_b, D));
        TRY(reg_mat(m, p + "mlp.fc1.weight", &L.w_fc1, Mv, D));
        TRY(reg_f32(m, p + "mlp.fc1.bias", &L.b_fc1, Mv));
        TRY(reg_mat(m, p + "mlp.fc2.weight", &L.w_fc2, D, Mv));
        TRY(reg_f32(m, p + "mlp.fc2.bias", &L.b_fc2, D));
        TRY(reg_f32(m, p + "layer_norm2.weight", &L.ln2_g, D));
        TRY(reg_f32(m, p + "layer_norm2.bias", &L.ln2_b, D));
    }
    TRY(reg_f32(m, vm + "post_layernorm.weight", &m->post_g, D));
    TRY(reg_f32(m, vm + "post_layernorm.bias", &m->post_b, D));

    const std::string tb = "text_decoder.bert.";
    // the embedding table is read twice: fp32 rows for the lookup, compute-dtype [V,T] as the (tied) LM-head weight
    TRY(walloc(m, (void**)&m->word_f32, (size_t)V * T * 4));
    add_slot(m, tb + "embeddings.word_embeddings.weight", m->word_f32, CAP_DT_F32, V, T);
    if (m->gdt == CAP_DT_F32) {
        m->word_t = m->word_f32;
    } else {
        TRY(walloc(m, &m->word_t, (size_t)V * T * m->esz));
        add_slot(m, tb + "embeddings.word_embeddings.weight", m->word_t, m->gdt, V, T);
    }
    TRY(reg_f32(m, tb + "embeddings.position_embeddings.weight", &m->tpos, (int64_t)c.max_pos * T));
    TRY(reg_f32(m, tb + "embeddings.LayerNorm.weight", &m->emb_g, T));
    TRY(reg_f32(m, tb + "embeddings.LayerNorm.bias", &m->emb_b, T));
    // cross-attention K/V projections of all layers fused into one [L*2*T, D] weight (one GEMM per image batch)
    TRY(walloc(m, &m->w_ckv, (size_t)c.t_layers * 2 * T * D * m->esz));
    TRY(walloc(m, (void**)&m->b_ckv, (size_t)c.t_layers * 2 * T * 4));
    m->tl.resize(c.t_layers);
    for (int i = 0; i < c.t_layers; ++i) {
        TLayer& L = m->tl[i];
        const std::string p = tb + "encoder.layer." + std::to_string(i) + ".";
        TRY(walloc(m, &L.w_qkv, (size_t)3 * T * T * m->esz));
        TRY(walloc(m, (void**)&L.b_qkv, (size_t)3 * T * 4));
        const char* nm[3] = {"query", "key", "value"};
        for (int j = 0; j < 3; ++j) {
            add_slot(m, p + "attention.self." + nm[j] + ".weight", (char*)L.w_qkv + (size_t)j * T * T * m->esz, m->gdt, T, T);
            add_slot(m, p + "attention.self." + nm[j] + ".bias", L.b_qkv + (size_t)j * T, CAP_DT_F32, 1, T);
        }
        TRY(reg_mat(m, p + "attention.output.dense.weight", &L.w_so, T, T));
        TRY(reg_f32(m, p + "attention.output.dense.bias", &L.b_so, T));
        TRY(reg_f32(m, p + "attention.output.LayerNorm.weight", &L.so_g, T));
        TRY(reg_f32(m, p + "attention.output.LayerNorm.bias", &L.so_b, T));
        TRY(reg_mat(m, p + "crossattention.self.query.weight", &L.w_cq, T, T));
        TRY(reg_f32(m, p + "crossattention.self.query.bias", &L.b_cq, T));
        for (int j = 0; j < 2; ++j) {
            add_slot(m, p + "crossattention.self." + nm[j + 1] + ".weight",
                     (char*)m->w_ckv + ((size_t)i * 2 + j) * T * D * m->esz, m->gdt, T, D);
            add_slot(m, p + "crossattention.self." + nm[j + 1] + ".bias", m->b_ckv + ((size_t)i * 2 + j) * T, CAP_DT_F32, 1, T);
        }
        TRY(reg_mat(m, p + "crossattention.output.dense.weight", &L.w_co, T, T));
        TRY(reg_f32(m, p + "crossattention.output.dense.bias", &L.b_co, T));
        TRY(reg_f32(m, p + "crossattention.output.LayerNorm.weight", &L.co_g, T));
        TRY(reg_f32(m, p + "crossattention.output.LayerNorm.bias", &L.co_b, T));
        TRY(reg_mat(m, p + "intermediate.dense.weight", &L.w_f1, F, T));
        TRY(reg_f32(m, p + "intermediate.dense.bias", &L.b_f1, F));
        TRY(reg_mat(m, p + "output.dense.weight", &L.w_f2, T, F));
        TRY(reg_f32(m, p + "output.dense.bias", &L.b_f2, T));
        TRY(reg_f32(m, p + "output.LayerNorm.weight", &L.f_g, T));
        TRY(reg_f32(m, p + "output.LayerNorm.bias", &L.f_b, T));
    }
    const std::string cp = "text_decoder.cls.predictions.";
    TRY(reg_mat(m, cp + "transform.dense.weight", &m->w_tr, T, T));
    TRY(reg_f32(m, cp + "transform.dense.bias", &m->b_tr, T));
    TRY(reg_f32(m, cp + "transform.LayerNorm.weight", &m->tr_g, T));
    TRY(reg_f32(m, cp + "transform.LayerNorm.bias", &m->tr_b, T));
    TRY(reg_f32(m, cp + "bias", &m->b_vocab, V));
    return 0;
}

int reg_block(Captioner* m, const std::string& p, CBlock& b, int E, int F, bool cross, int idx) {
    b.cross = cross;
    TRY(reg_f32(m, p + ".ln_1.weight", &b.ln1_g, E));
    TRY(reg_f32(m, p + ".ln_1.bias", &b.ln1_b, E));
    if (!cross) {
        TRY(reg_mat(m, p + ".attn.in_proj_weight", &b.w_in, 3 * E, E));
        TRY(reg_f32(m, p + ".attn.in_proj_bias", &b.b_in, 3 * E));
    } else {
        const std::string d = "derived.cross_q." + std::to_string(idx);
        TRY(reg_mat(m, d + ".weight", &b.w_in, E, E));
        TRY(reg_f32(m, d + ".bias", &b.b_in, E));
    }
    TRY(reg_mat(m, p + ".attn.out_proj.weight", &b.w_o, E, E));
    TRY(reg_f32(m, p + ".attn.out_proj.bias", &b.b_o, E));
    TRY(reg_f32(m, p + ".ln_2.weight", &b.ln2_g, E));
    TRY(reg_f32(m, p + ".ln_2.bias", &b.ln2_b, E));
    TRY(reg_mat(m, p + ".mlp.c_fc.weight", &b.w_fc, F, E));
    TRY(reg_f32(m, p + ".mlp.c_fc.bias", &b.b_fc, F));
    TRY(reg_mat(m, p + ".mlp.c_proj.weight", &b.w_pr, E, F));
    TRY(reg_f32(m, p + ".mlp.c_proj.bias", &b.b_pr, E));
    return 0;
}

// open_clip CoCa state-dict names (SURVEY.md section 5 "Checkpoint / resume"); `derived.*` tensors are computed once by
// the host loader (embodied_captioning_amd/coca_weights.py) from the checkpoint.
int build_coca(Captioner* m) {
    const CapConfig& c = m->c;
    const int D = c.v_hidden, Mv = c.v_mlp, E = c.embed_dim, F = c.t_ffn, V = c.vocab, Q = c.pool_queries;
    m->Q = Q; m->E = E;
    TRY(reg_f32(m, "visual.class_embedding", &m->cls, D));
    TRY(reg_f32(m, "visual.positional_embedding", &m->vpos, (int64_t)m->NT * D));
    TRY(reg_mat(m, "visual.conv1.weight", &m->w_patch, D, m->Kpatch, m->Kpad));
    m->b_patch = nullptr;                                  // conv1 has no bias in open_clip's ViT
    TRY(reg_f32(m, "visual.ln_pre.weight", &m->ln_pre_g, D));
    TRY(reg_f32(m, "visual.ln_pre.bias", &m->ln_pre_b, D));
    m->vl.resize(c.v_layers);
    for (int i = 0; i < c.v_layers; ++i) {
        VLayer& L = m->vl[i];
        const std::string p = "visual.transformer.resblocks." + std::to_string(i) + ".";
        TRY(reg_mat(m, p + "attn.in_proj_weight", &L.w_qkv, 3 * D, D));
        TRY(reg_f32(m, p + "attn.in_proj_bias", &L.b_qkv, 3 * D));
        TRY(reg_mat(m, p + "attn.out_proj.weight", &L.w_proj, D, D));
        TRY(reg_f32(m, p + "attn.out_proj.bias", &L.b_proj, D));
        TRY(reg_f32(m, p + "ln_1.weight", &L.ln1_g, D));
        TRY(reg_f32(m, p + "ln_1.bias", &L.ln1_b, D));
        TRY(reg_mat(m, p + "mlp.c_fc.weight", &L.w_fc1, Mv, D));
        TRY(reg_f32(m, p + "mlp.c_fc.bias", &L.b_fc1, Mv));
        TRY(reg_mat(m, p + "mlp.c_proj.weight", &L.w_fc2, D, Mv));
        TRY(reg_f32(m, p + "mlp.c_proj.bias", &L.b_fc2, D));
        TRY(reg_f32(m, p + "ln_2.weight", &L.ln2_g, D));
        TRY(reg_f32(m, p + "ln_2.bias", &L.ln2_b, D));
    }
    TRY(reg_f32(m, "visual.attn_pool.ln_k.weight", &m->lnk_g, D));
    TRY(reg_f32(m, "visual.attn_pool.ln_k.bias", &m->lnk_b, D));
    TRY(reg_f32(m, "derived.pool_q", &m->pool_q, (int64_t)Q * E));
    TRY(reg_mat(m, "derived.pool_kv.weight", &m->w_pool_kv, 2 * E, D));
    TRY(reg_f32(m, "derived.pool_kv.bias", &m->b_pool_kv, 2 * E));
    TRY(reg_mat(m, "visual.attn_pool.attn.out_proj.weight", &m->w_pool_o, E, E));
    TRY(reg_f32(m, "visual.attn_pool.attn.out_proj.bias", &m->b_pool_o, E));
    TRY(reg_f32(m, "visual.ln_post.weight", &m->lnpost_g, E));
    TRY(reg_f32(m, "visual.ln_post.bias", &m->lnpost_b, E));
    TRY(reg_f32(m, "text.token_embedding.weight", &m->tok_emb, (int64_t)V * E));
    TRY(reg_f32(m, "text.positional_embedding", &m->tpos, (int64_t)c.max_pos * E));
    m->cb.resize(c.t_layers + 2 * c.mm_layers);
    int nb = 0, ncache = 0;
    for (int i = 0; i < c.t_layers; ++i) {
        m->cb[nb].cache = ncache++;
        TRY(reg_block(m, "text.transformer.resblocks." + std::to_string(i), m->cb[nb], E, F, false, i));
        ++nb;
    }
    for (int i = 0; i < c.mm_layers; ++i) {
        m->cb[nb].cache = ncache++;
        TRY(reg_block(m, "text_decoder.resblocks." + std::to_string(i), m->cb[nb], E, F, false, i));
        ++nb;
        m->cb[nb].cross_idx = i;
        TRY(reg_block(m, "text_decoder.cross_attn." + std::to_string(i), m->cb[nb], E, F, true, i));
        ++nb;
    }
    TRY(reg_mat(m, "derived.cross_kv.weight", &m->w_ckv, (int64_t)c.mm_layers * 2 * E, E));
    TRY(reg_f32(m, "derived.cross_kv.bias", &m->b_ckv, (int64_t)c.mm_layers * 2 * E));
    TRY(reg_f32(m, "text_decoder.ln_final.weight", &m->lnf_g, E));
    TRY(reg_f32(m, "text_decoder.ln_final.bias", &m->lnf_b, E));
    TRY(reg_mat(m, "derived.vocab.weight", &m->w_cvocab, V, E));
    // constant vectors for the affine-free LayerNorm that feeds the folded cross-K/V projection
    TRY(walloc(m, (void**)&m->ones, (size_t)E * 4));
    TRY(walloc(m, (void**)&m->zeros, (size_t)E * 4));
    if (!m->replay) {
        TRY(launch_fill_f32(m->ones, 1.0f, E, nullptr));
        TRY(launch_fill_f32(m->zeros, 0.0f, E, nullptr));
        CAP_HIP_CHECK(hipDeviceSynchronize());
    }
    return 0;
}

int build_arena_coca(Captioner* m) {
    const CapConfig& c = m->c;
    const size_t Bm = c.max_batch, NT = m->NT, D = c.v_hidden, E = c.embed_dim, e = m->esz, Q = c.pool_queries;
    const size_t M = Bm * NT, R = Bm * c.max_beams, Lm = c.max_len, H = c.t_heads;     // R: decode rows (image x beam)
    TRY(dev_alloc(m, &m->patches, Bm * m->P * m->Kpad * e));
    CAP_HIP_CHECK(hipMemset(m->patches, 0, Bm * m->P * m->Kpad * e));
    TRY(dev_alloc(m, (void**)&m->X, M * D * 4));
    if (!vit_adds_in_place(m->gdt)) TRY(dev_alloc(m, (void**)&m->delta, M * D * (kDeltaInT ? m->esz : 4)));
    TRY(dev_alloc(m, &m->ln, M * D * e));
    TRY(dev_alloc(m, &m->qkv, M * 3 * D * e));
    TRY(dev_alloc(m, &m->ctx, M * D * e));
    TRY(dev_alloc(m, &m->mlp, M * c.v_mlp * e));
    TRY(dev_alloc(m, (void**)&m->emb_f, 256));
    TRY(dev_alloc(m, &m->emb_t, M * D * e));
    TRY(dev_alloc(m, &m->pool_kvbuf, M * 2 * E * e));
    TRY(dev_alloc(m, &m->pool_ctx, Bm * Q * E * e));
    TRY(dev_alloc(m, (void**)&m->pool_o, Bm * Q * E * 4));
    TRY(dev_alloc(m, (void**)&m->img_tokens, Bm * Q * E * 4));
    TRY(dev_alloc(m, &m->xhat, Bm * Q * E * e));
    TRY(dev_alloc(m, &m->cross, (size_t)c.mm_layers * 2 * m->cross_block((size_t)Bm * H * Q)));
    TRY(dev_alloc(m, (void**)&m->seq, R * Lm * 4));
    TRY(dev_alloc(m, (void**)&m->finished, R * 4));
    TRY(dev_alloc(m, (void**)&m->lens, R * 4));
    TRY(dev_alloc(m, (void**)&m->anc, 2 * R * Lm * 4 + 256));    // beam ancestry of the self-attention caches (beam.hip)
    TRY(dev_alloc(m, (void**)&m->dx, R * E * 4));
    TRY(dev_alloc(m, (void**)&m->dy, R * E * 4));
    TRY(dev_alloc(m, (void**)&m->dx2, (size_t)(R > SMALL_MAX_ROWS ? R : SMALL_MAX_ROWS) * E * 4));
    TRY(dev_alloc(m, (void**)&m->dpart, 12 * R * E * 4));
    TRY(dev_alloc(m, &m->dx_t, R * E * e));
    TRY(dev_alloc(m, &m->dq, R * E * e));
    TRY(dev_alloc(m, &m->dctx, R * E * e));
    TRY(dev_alloc(m, &m->dh, R * c.t_ffn * e));
    m->ldl = (c.vocab + 3) & ~3;
    TRY(dev_alloc(m, (void**)&m->logits, R * (size_t)m->ldl * 4));
    m->ccache.resize(c.t_layers + c.mm_layers);
    for (auto& p : m->ccache) TRY(dev_alloc(m, &p, 2 * R * H * Lm * 64 * e));
    TRY(dev_alloc(m, &m->beam, beam_state_bytes((int)Bm, c.max_beams, (int)Lm)));      // (a 1-beam search exists: beam groups)
    return 0;
}

int build_arena(Captioner* m) {
    const CapConfig& c = m->c;
    const size_t Bm = c.max_batch, NT = m->NT, D = c.v_hidden, T = c.t_hidden, e = m->esz;
    const size_t M = Bm * NT, R = Bm * c.max_beams, Lm = c.max_len, H = c.t_heads;
    TRY(dev_alloc(m, &m->patches, Bm * m->P * m->Kpad * e));
    CAP_HIP_CHECK(hipMemset(m->patches, 0, Bm * m->P * m->Kpad * e));
    TRY(dev_alloc(m, (void**)&m->X, M * D * 4));
    if (!vit_adds_in_place(m->gdt)) TRY(dev_alloc(m, (void**)&m->delta, M * D * (kDeltaInT ? m->esz : 4)));
    TRY(dev_alloc(m, &m->ln, M * D * e));
    TRY(dev_alloc(m, &m->qkv, M * 3 * D * e));
    TRY(dev_alloc(m, &m->ctx, M * D * e));
    TRY(dev_alloc(m, &m->mlp, M * c.v_mlp * e));
    TRY(dev_alloc(m, (void**)&m->emb_f, M * D * 4));
    TRY(dev_alloc(m, &m->emb_t, M * D * e));
    TRY(dev_alloc(m, &m->cross, (size_t)c.t_layers * 2 * m->cross_block((size_t)Bm * H * NT)));
    TRY(dev_alloc(m, (void**)&m->seq, R * Lm * 4));
    TRY(dev_alloc(m, (void**)&m->finished, R * 4));
    TRY(dev_alloc(m, (void**)&m->lens, R * 4));
    TRY(dev_alloc(m, (void**)&m->live, R * 4));
    TRY(dev_alloc(m, (void**)&m->n_live, 256));
    TRY(dev_alloc(m, (void**)&m->anc, 2 * R * Lm * 4));
    TRY(dev_alloc(m, (void**)&m->dx, R * T * 4));
    TRY(dev_alloc(m, (void**)&m->dy, R * T * 4));
    TRY(dev_alloc(m, (void**)&m->dx2, (size_t)(R > SMALL_MAX_ROWS ? R : SMALL_MAX_ROWS) * T * 4));
    TRY(dev_alloc(m, (void**)&m->dpart, 12 * R * T * 4));      // split-K slabs: 8 x [R,T] (ffn) or 4 x [R,3T] (qkv)
    TRY(dev_alloc(m, &m->dx_t, R * T * e));
    TRY(dev_alloc(m, &m->dq, R * T * e));
    TRY(dev_alloc(m, &m->dctx, R * T * e));
    TRY(dev_alloc(m, &m->dh, R * c.t_ffn * e));
    m->ldl = (c.vocab + 3) & ~3;
    TRY(dev_alloc(m, (void**)&m->logits, R * (size_t)m->ldl * 4));
    for (int i = 0; i < c.t_layers; ++i) TRY(dev_alloc(m, &m->tl[i].self_cache, 2 * R * H * Lm * 64 * e));
    TRY(dev_alloc(m, &m->beam, beam_state_bytes((int)Bm, c.max_beams, (int)Lm)));
    return 0;
}

// ---------------------------------------------------------------------------------------------- profiling
struct ProfScope {
    Captioner* m; hipStream_t s; int idx = -1;
    ProfScope(Captioner* m_, hipStream_t s_, const char* tag, double flops, double bytes) : m(m_), s(s_) {
        if (!m->prof) return;
        ProfTag t; t.tag = tag; t.flops = flops; t.bytes = bytes;
        if (hipEventCreate(&t.e0) != hipSuccess || hipEventCreate(&t.e1) != hipSuccess) return;
        (void)hipEventRecord(t.e0, s);
        m->prof_recs.push_back(t);
        idx = (int)m->prof_recs.size() - 1;
    }
    ~ProfScope() { if (idx >= 0) (void)hipEventRecord(m->prof_recs[idx].e1, s); }
};

int gemm(Captioner* m, hipStream_t s, const char* tag, const void* A, int lda, const void* W, int ldw, void* C, int ldc,
         const float* bias, const float* resid, int M, int N, int K, int gelu, int out_f32, int epi = EPI_STORE,
         int p0 = 0, int p1 = 0, int p2 = 0, int p3 = 0, const float* aux = nullptr, void* C2 = nullptr) {
    GemmParams p;
    memset(&p, 0, sizeof(p));
    p.A = A; p.lda = lda; p.W = W; p.ldw = ldw; p.C = C; p.ldc = ldc; p.bias = bias; p.resid = resid; p.ldr = ldc;
    p.M = M; p.N = N; p.K = K; p.gelu = gelu; p.out_f32 = out_f32; p.epi = epi;
    p.p0 = p0; p.p1 = p1; p.p2 = p2; p.p3 = p3; p.aux = aux; p.C2 = C2; p.splitk = 1;
    p.kv16 = epi == EPI_CROSSKV && m->kv16 ? 1 : 0;
    const double osz = out_f32 ? 4.0 : (double)m->esz;
    ProfScope ps(m, s, tag, 2.0 * M * N * K, ((double)M * K + (double)N * K) * m->esz + (double)M * N * (osz + (resid ? 4.0 : 0.0)));
    return launch_gemm(m->gdt, p, 0, s);   // tile 0 = auto (stream kernel for encoder-sized problems without residual)
}

// Decode loops stop when every caption is finished, as HF generate does (`unfinished_sequences.max() == 0` /
// `is_done.all()`); the remaining steps would only write pad.  The device keeps the state, the host looks at it every
// `poll` steps: one tiny kernel writes the number of open rows (greedy) or the beam loop's active flag to a host-mapped
// word, then the stream is synchronised.  Off by default (poll = 0): the loop is then free of host synchronisation and
// can be captured in a graph.
__global__ void poll_open_kernel(const int* __restrict__ finished, int R, const int* __restrict__ beam_active, int* host_out) {
    if (beam_active) { if (threadIdx.x == 0) *host_out = *beam_active; return; }
    __shared__ int cnt;
    if (threadIdx.x == 0) cnt = 0;
    __syncthreads();
    int c = 0;
    for (int r = threadIdx.x; r < R; r += blockDim.x) c += finished[r] == 0;
    if (c) atomicAdd(&cnt, c);
    __syncthreads();
    if (threadIdx.x == 0) *host_out = cnt;
}
// true = every row is finished (the caller leaves its loop).  step = index of the step just completed, steps = loop length.
int poll_all_finished(Captioner* m, int step, int steps, const int* finished, int R, const int* beam_active, hipStream_t s, bool* done) {
    *done = false;
    if (m->poll <= 0 || !m->host_flag || (step + 1) % m->poll != 0 || step + 2 >= steps) return 0;
    hipLaunchKernelGGL(poll_open_kernel, dim3(1), dim3(256), 0, s, finished, R, beam_active, m->host_flag_dev);
    CAP_HIP_CHECK(hipGetLastError());
    CAP_HIP_CHECK(hipStreamSynchronize(s));
    *done = *(volatile int*)m->host_flag == 0;
    return 0;
}

// ---------------------------------------------------------------------------------------------- BLIP-2 OPT
int run_encoder(Captioner* m, const void* pixels, int fmt, int B, float* out_embeds, hipStream_t s);
int gemm_partial(Captioner* m, hipStream_t s, const char* tag, const void* A, const void* W, float* part, int R, int N,
                 int K, int max_S, int* S_out, const int* m_live = nullptr);
__global__ void init_seq_kernel(int* seq, int* fin, int* len, int R, int L, int bos, int pad);
__global__ void copy_logits_kernel(const float* src, int ld, float* dst, int R, int V);
__global__ void copy_new_tokens_kernel(const int* seq, int seq_ld, int P, const int* lens, int* out_ids, int* out_len, int B, int n) {
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < B * n; i += gridDim.x * blockDim.x) out_ids[i] = seq[(size_t)(i / n) * seq_ld + P + i % n];
    if (out_len)
        for (int b = blockIdx.x * blockDim.x + threadIdx.x; b < B; b += gridDim.x * blockDim.x) out_len[b] = min(lens[b] - P, n);
}
// HF `Blip2ForConditionalGeneration` state-dict names (transformers 5.x).  `derived.qformer_x0` = qformer.layernorm(
// query_tokens), computed once by the host loader (a constant of the checkpoint).
int build_blip2(Captioner* m) {
    const CapConfig& c = m->c;
    const int D = c.v_hidden, Mv = c.v_mlp, Q = c.q_hidden, F = c.q_ffn, T = c.t_hidden, G = c.t_ffn, V = c.vocab, nq = c.num_query_tokens;
    const std::string vm = "vision_model.";
    TRY(reg_f32(m, vm + "embeddings.class_embedding", &m->cls, D));
    TRY(reg_f32(m, vm + "embeddings.position_embedding", &m->vpos, (int64_t)m->NT * D));
    TRY(reg_mat(m, vm + "embeddings.patch_embedding.weight", &m->w_patch, D, m->Kpatch, m->Kpad));
    TRY(reg_f32(m, vm + "embeddings.patch_embedding.bias", &m->b_patch, D));
    m->vl.resize(c.v_layers);
    for (int i = 0; i < c.v_layers; ++i) {
        VLayer& L = m->vl[i];
        const std::string p = vm + "encoder.layers." + std::to_string(i) + ".";
        TRY(reg_mat(m, p + "self_attn.qkv.weight", &L.w_qkv, 3 * D, D));
        TRY(reg_f32(m, p + "self_attn.qkv.bias", &L.b_qkv, 3 * D));
        TRY(reg_mat(m, p + "self_attn.projection.weight", &L.w_proj, D, D));
        TRY(reg_f32(m, p + "self_attn.projection.bias", &L.b_proj, D));
        TRY(reg_f32(m, p + "layer_norm1.weight", &L.ln1_g, D));
        TRY(reg_f32(m, p + "layer_norm1.bias", &L.ln1_b, D));
        TRY(reg_mat(m, p + "mlp.fc1.weight", &L.w_fc1, Mv, D));
        TRY(reg_f32(m, p + "mlp.fc1.bias", &L.b_fc1, Mv));
        TRY(reg_mat(m, p + "mlp.fc2.weight", &L.w_fc2, D, Mv));
        TRY(reg_f32(m, p + "mlp.fc2.bias", &L.b_fc2, D));
        TRY(reg_f32(m, p + "layer_norm2.weight", &L.ln2_g, D));
        TRY(reg_f32(m, p + "layer_norm2.bias", &L.ln2_b, D));
    }
    TRY(reg_f32(m, vm + "post_layernorm.weight", &m->post_g, D));
    TRY(reg_f32(m, vm + "post_layernorm.bias", &m->post_b, D));

    TRY(reg_f32(m, "derived.qformer_x0", &m->q_x0, (int64_t)nq * Q));
    m->ql.resize(c.q_layers);
    const char* nm[3] = {"query", "key", "value"};
    for (int i = 0; i < c.q_layers; ++i) {
        QLayer& L = m->ql[i];
        L.cross = i % c.q_cross_freq == 0;
        const std::string p = "qformer.encoder.layer." + std::to_string(i) + ".";
        TRY(walloc(m, &L.w_qkv, (size_t)3 * Q * Q * m->esz));
        TRY(walloc(m, (void**)&L.b_qkv, (size_t)3 * Q * 4));
        for (int j = 0; j < 3; ++j) {
            add_slot(m, p + "attention.attention." + nm[j] + ".weight", (char*)L.w_qkv + (size_t)j * Q * Q * m->esz, m->gdt, Q, Q);
            add_slot(m, p + "attention.attention." + nm[j] + ".bias", L.b_qkv + (size_t)j * Q, CAP_DT_F32, 1, Q);
        }
        TRY(reg_mat(m, p + "attention.output.dense.weight", &L.w_so, Q, Q));
        TRY(reg_f32(m, p + "attention.output.dense.bias", &L.b_so, Q));
        TRY(reg_f32(m, p + "attention.output.LayerNorm.weight", &L.so_g, Q));
        TRY(reg_f32(m, p + "attention.output.LayerNorm.bias", &L.so_b, Q));
        if (L.cross) {
            TRY(reg_mat(m, p + "crossattention.attention.query.weight", &L.w_cq, Q, Q));
            TRY(reg_f32(m, p + "crossattention.attention.query.bias", &L.b_cq, Q));
            TRY(walloc(m, &L.w_ckv, (size_t)2 * Q * D * m->esz));
            TRY(walloc(m, (void**)&L.b_ckv, (size_t)2 * Q * 4));
            for (int j = 0; j < 2; ++j) {
                add_slot(m, p + "crossattention.attention." + nm[j + 1] + ".weight", (char*)L.w_ckv + (size_t)j * Q * D * m->esz, m->gdt, Q, D);
                add_slot(m, p + "crossattention.attention." + nm[j + 1] + ".bias", L.b_ckv + (size_t)j * Q, CAP_DT_F32, 1, Q);
            }
            TRY(reg_mat(m, p + "crossattention.output.dense.weight", &L.w_co, Q, Q));
            TRY(reg_f32(m, p + "crossattention.output.dense.bias", &L.b_co, Q));
            TRY(reg_f32(m, p + "crossattention.output.LayerNorm.weight", &L.co_g, Q));
            TRY(reg_f32(m, p + "crossattention.output.LayerNorm.bias", &L.co_b, Q));
        }
        TRY(reg_mat(m, p + "intermediate_query.dense.weight", &L.w_f1, F, Q));
        TRY(reg_f32(m, p + "intermediate_query.dense.bias", &L.b_f1, F));
        TRY(reg_mat(m, p + "output_query.dense.weight", &L.w_f2, Q, F));
        TRY(reg_f32(m, p + "output_query.dense.bias", &L.b_f2, Q));
        TRY(reg_f32(m, p + "output_query.LayerNorm.weight", &L.f_g, Q));
        TRY(reg_f32(m, p + "output_query.LayerNorm.bias", &L.f_b, Q));
    }
    TRY(reg_mat(m, "language_projection.weight", &m->w_lproj, T, Q));
    TRY(reg_f32(m, "language_projection.bias", &m->b_lproj, T));

    const std::string lm = "language_model.model.decoder.";
    // token table twice: fp32 rows for the lookup, compute dtype as the tied LM head
    TRY(walloc(m, (void**)&m->o_tok, (size_t)V * T * 4));
    add_slot(m, lm + "embed_tokens.weight", m->o_tok, CAP_DT_F32, V, T);
    if (m->gdt == CAP_DT_F32) m->o_tok_t = m->o_tok;
    else { TRY(walloc(m, &m->o_tok_t, (size_t)V * T * m->esz)); add_slot(m, lm + "embed_tokens.weight", m->o_tok_t, m->gdt, V, T); }
    TRY(reg_f32(m, lm + "embed_positions.weight", &m->o_pos, (int64_t)(c.max_pos + 2) * T));
    const size_t Bm = c.max_batch, Lmax = nq + 1 + c.max_len;
    m->ol.resize(c.t_layers);
    const char* pn[3] = {"q_proj", "k_proj", "v_proj"};
    for (int i = 0; i < c.t_layers; ++i) {
        OLayer& L = m->ol[i];
        const std::string p = lm + "layers." + std::to_string(i) + ".";
        TRY(walloc(m, &L.w_qkv, (size_t)3 * T * T * (m->wq8 ? 1 : m->esz)));
        if (m->wq8) TRY(walloc(m, (void**)&L.s_qkv, (size_t)3 * T * 4));
        TRY(walloc(m, (void**)&L.b_qkv, (size_t)3 * T * 4));
        for (int j = 0; j < 3; ++j) {
            // (int8: a 16-row tile's blocks are contiguous, so rows j T .. of the fused matrix start at byte j T T)
            if (m->wq8) add_slot(m, p + "self_attn." + pn[j] + ".weight", (char*)L.w_qkv + (size_t)j * T * T, CAP_DT_I8W, T, T, 0, L.s_qkv + (size_t)j * T);
            else add_slot(m, p + "self_attn." + pn[j] + ".weight", (char*)L.w_qkv + (size_t)j * T * T * m->esz, m->gdt, T, T);
            add_slot(m, p + "self_attn." + pn[j] + ".bias", L.b_qkv + (size_t)j * T, CAP_DT_F32, 1, T);
        }
        if (m->wq8) {
            TRY(reg_mat_i8(m, p + "self_attn.out_proj.weight", &L.w_o, &L.s_o, T, T));
            TRY(reg_mat_i8(m, p + "fc1.weight", &L.w_f1, &L.s_f1, G, T));
            TRY(reg_mat_i8(m, p + "fc2.weight", &L.w_f2, &L.s_f2, T, G));
        } else {
            TRY(reg_mat(m, p + "self_attn.out_proj.weight", &L.w_o, T, T));
            TRY(reg_mat(m, p + "fc1.weight", &L.w_f1, G, T));
            TRY(reg_mat(m, p + "fc2.weight", &L.w_f2, T, G));
        }
        TRY(reg_f32(m, p + "self_attn.out_proj.bias", &L.b_o, T));
        TRY(reg_f32(m, p + "self_attn_layer_norm.weight", &L.ln1_g, T));
        TRY(reg_f32(m, p + "self_attn_layer_norm.bias", &L.ln1_b, T));
        TRY(reg_f32(m, p + "fc1.bias", &L.b_f1, G));
        TRY(reg_f32(m, p + "fc2.bias", &L.b_f2, T));
        TRY(reg_f32(m, p + "final_layer_norm.weight", &L.ln2_g, T));
        TRY(reg_f32(m, p + "final_layer_norm.bias", &L.ln2_b, T));
        TRY(dev_alloc(m, &L.kc, Bm * Lmax * T * m->esz));
        TRY(dev_alloc(m, &L.vc, Bm * Lmax * T * m->esz));
    }
    TRY(reg_f32(m, lm + "final_layer_norm.weight", &m->o_lnf_g, T));
    TRY(reg_f32(m, lm + "final_layer_norm.bias", &m->o_lnf_b, T));

    // arena
    const size_t NT = m->NT, M = Bm * NT, e = m->esz, P = nq + 1;
    TRY(dev_alloc(m, &m->patches, Bm * m->P * m->Kpad * e));
    CAP_HIP_CHECK(hipMemset(m->patches, 0, Bm * m->P * m->Kpad * e));
    TRY(dev_alloc(m, (void**)&m->X, M * D * 4));
    if (!vit_adds_in_place(m->gdt)) TRY(dev_alloc(m, (void**)&m->delta, M * D * (kDeltaInT ? e : 4)));
    TRY(dev_alloc(m, &m->ln, M * D * e));
    TRY(dev_alloc(m, &m->qkv, M * 3 * D * e));
    TRY(dev_alloc(m, &m->ctx, M * D * e));
    TRY(dev_alloc(m, &m->mlp, M * Mv * e));
    TRY(dev_alloc(m, (void**)&m->emb_f, M * D * 4));
    TRY(dev_alloc(m, &m->emb_t, M * D * e));
    TRY(dev_alloc(m, (void**)&m->qx, Bm * nq * Q * 4));
    TRY(dev_alloc(m, (void**)&m->qy, Bm * nq * Q * 4));
    TRY(dev_alloc(m, &m->qx_t, Bm * nq * Q * e));
    TRY(dev_alloc(m, &m->qqkv, Bm * nq * 3 * Q * e));
    TRY(dev_alloc(m, &m->qctx, Bm * nq * Q * e));
    TRY(dev_alloc(m, &m->qh, Bm * nq * F * e));
    TRY(dev_alloc(m, &m->qkvimg, M * 2 * Q * e));
    TRY(dev_alloc(m, (void**)&m->lm_proj, Bm * nq * T * 4));
    TRY(dev_alloc(m, (void**)&m->ox, Bm * P * T * 4));
    TRY(dev_alloc(m, &m->oh_t, Bm * P * T * e));
    TRY(dev_alloc(m, &m->oqkv, Bm * P * 3 * T * e));
    TRY(dev_alloc(m, &m->octx, Bm * P * T * e));
    TRY(dev_alloc(m, &m->off, Bm * P * G * e));
    {
        size_t dpart_bytes = (size_t)8 * Bm * std::max(3 * T, G) * 4;                     // split-K slabs of the decode-step GEMMs
        if (m->wq8) {                                   // int8 weights: the prompt pass runs the same chain over Bm * P rows
            const size_t Sm = (size_t)std::max(skinny_i8_plan(T, T, false), skinny_i8_plan(T, G, false));
            dpart_bytes = std::max(dpart_bytes, Sm * std::min<size_t>(Bm, kI8SkinnyPromptCrops) * P * T * 4);
            if (Bm > (size_t)kI8SkinnyPromptCrops) {    // larger batches: the tiled GEMM's fp32 output [Bm * P, N] + the bf16 scratch
                dpart_bytes = std::max(dpart_bytes, Bm * P * (size_t)std::max(3 * T, G) * 4);
                TRY(dev_alloc(m, &m->w8_scratch, (size_t)std::max(3 * T, G) * T * 2));
            }
        }
        TRY(dev_alloc(m, (void**)&m->dpart, dpart_bytes));
    }
    TRY(dev_alloc(m, (void**)&m->seq, Bm * Lmax * 4));
    TRY(dev_alloc(m, (void**)&m->finished, Bm * 4));
    TRY(dev_alloc(m, (void**)&m->lens, Bm * 4));
    m->ldl = (V + 3) & ~3;
    TRY(dev_alloc(m, (void**)&m->logits, Bm * (size_t)m->ldl * 4));
    return 0;
}

// Q-Former over the image tokens (emb_t [B * NT, D]) -> qx_t [B * nq, Q]
int run_qformer(Captioner* m, int B, hipStream_t s) {
    const CapConfig& c = m->c;
    const int D = c.v_hidden, Q = c.q_hidden, F = c.q_ffn, H = c.q_heads, nq = c.num_query_tokens, NT = m->NT, R = B * nq, hd = Q / H;
    const size_t e = m->esz;
    const int af = m->gdt == CAP_DT_G8 ? 1 : 0;     // split mode: what the attention kernels read (q, k, v) is fp32, only GEMM operands are G8
    TRY(launch_rows_broadcast(m->gdt, m->q_x0, m->qx, m->qx_t, B, nq, Q, s));
    for (int i = 0; i < c.q_layers; ++i) {
        const QLayer& L = m->ql[i];
        TRY(gemm(m, s, "qf_gemm_qkv", m->qx_t, Q, L.w_qkv, Q, m->qqkv, 3 * Q, L.b_qkv, nullptr, R, 3 * Q, Q, 0, af));
        {
            ProfScope ps(m, s, "qf_self_attn", 4.0 * B * H * (double)nq * nq * hd, (double)R * 4 * Q * e);
            const char* base = (const char*)m->qqkv;
            TRY(launch_generic_attention(m->dt, base, 3 * Q, (long)nq * 3 * Q, base + Q * e, 3 * Q, (long)nq * 3 * Q, base + 2 * Q * e, 3 * Q,
                                         (long)nq * 3 * Q, m->qctx, Q, (long)nq * Q, B, nq, nq, H, hd, -1, s, m->gdt));
        }
        TRY(gemm(m, s, "qf_gemm_so", m->qctx, Q, L.w_so, Q, m->qy, Q, L.b_so, m->qx, R, Q, Q, 0, 1));
        TRY(launch_layernorm(m->gdt, m->qy, Q, L.so_g, L.so_b, c.q_eps, m->qx_t, m->qx, R, Q, s));
        if (L.cross) {
            TRY(gemm(m, s, "qf_gemm_cq", m->qx_t, Q, L.w_cq, Q, m->qqkv, Q, L.b_cq, nullptr, R, Q, Q, 0, af));
            TRY(gemm(m, s, "qf_gemm_ckv", m->emb_t, D, L.w_ckv, D, m->qkvimg, 2 * Q, L.b_ckv, nullptr, B * NT, 2 * Q, D, 0, af));
            {
                ProfScope ps(m, s, "qf_cross_attn", 4.0 * B * H * (double)nq * NT * hd, (double)B * NT * 2 * Q * e);
                const char* kv = (const char*)m->qkvimg;
                TRY(launch_generic_attention(m->dt, m->qqkv, Q, (long)nq * Q, kv, 2 * Q, (long)NT * 2 * Q, kv + Q * e, 2 * Q, (long)NT * 2 * Q,
                                             m->qctx, Q, (long)nq * Q, B, nq, NT, H, hd, -1, s, m->gdt));
            }
            TRY(gemm(m, s, "qf_gemm_co", m->qctx, Q, L.w_co, Q, m->qy, Q, L.b_co, m->qx, R, Q, Q, 0, 1));
            TRY(launch_layernorm(m->gdt, m->qy, Q, L.co_g, L.co_b, c.q_eps, m->qx_t, m->qx, R, Q, s));
        }
        TRY(gemm(m, s, "qf_gemm_f1", m->qx_t, Q, L.w_f1, Q, m->qh, F, L.b_f1, nullptr, R, F, Q, 1, 0));
        TRY(gemm(m, s, "qf_gemm_f2", m->qh, F, L.w_f2, F, m->qy, Q, L.b_f2, m->qx, R, Q, F, 0, 1));
        TRY(launch_layernorm(m->gdt, m->qy, Q, L.f_g, L.f_b, c.q_eps, m->qx_t, m->qx, R, Q, s));
    }
    return 0;
}

// OPT decoder over L new positions per row (x = ox [B * L, T] fp32 with positions already added), cached prefix of `past`
// positions; leaves the logits of each row's last new position in m->logits.
int opt_step_gemm(Captioner* m, hipStream_t s, const char* tag, const void* A, const void* W, const float* bias, int act, void* out_t, int B,
                  int N, int K, bool ln, int* S_out, int out_dt, const float* wscale);

int run_opt(Captioner* m, int B, int L, int past, hipStream_t s) {
    const CapConfig& c = m->c;
    const int T = c.t_hidden, G = c.t_ffn, H = c.t_heads, hd = T / H, R = B * L;
    const int Lmax = c.num_query_tokens + 1 + c.max_len;
    const size_t e = m->esz;
    const int af = m->gdt == CAP_DT_G8 ? 1 : 0;     // split mode: q | k | v for the attention kernels and the caches are fp32
    if (m->wq8 && B > kI8SkinnyPromptCrops) {
        // int8 weights, a batch's prompt pass: B x 33 rows is a GEMM proper.  Each weight matrix is unpacked into a row-major bf16
        // scratch of its INTEGERS (exact), multiplied by the tiled kernel into fp32, the row scales applied to that output, and the
        // decode step's consumers (S = 1) finish it: (A . q^T) * scale + bias as in the weight-streaming kernels, the sums in the
        // tiled kernel's order.  Measured at 32 crops (us per fc1 GEMM): 365 on the rows-walking weight-stream kernel, ~135 here.
        auto tiled = [&](const char* tag, const void* A, const void* Wp, const float* wscale, int N, int K) -> int {
            TRY(launch_dequant_i8_rowmajor(Wp, m->w8_scratch, N, K, s));
            TRY(gemm(m, s, tag, A, K, m->w8_scratch, K, m->dpart, N, nullptr, nullptr, R, N, K, 0, 1));
            return launch_scale_cols(m->dpart, wscale, R, N, s);
        };
        TRY(launch_layernorm(m->gdt, m->ox, T, m->ol[0].ln1_g, m->ol[0].ln1_b, c.t_eps, m->oh_t, nullptr, R, T, s));
        for (int i = 0; i < c.t_layers; ++i) {
            const OLayer& Ly = m->ol[i];
            TRY(tiled("opt_gemm_qkv", m->oh_t, Ly.w_qkv, Ly.s_qkv, 3 * T, T));
            TRY(launch_reduce_bias_act(m->dt, m->dpart, 1, Ly.b_qkv, m->oqkv, R, 3 * T, 0, s));
            TRY(launch_kv_append(m->dt, m->oqkv, Ly.kc, Ly.vc, B, L, T, Lmax, past, s));
            {
                ProfScope ps(m, s, "opt_attn", 4.0 * B * H * (double)L * (past + L) * hd, 2.0 * B * (past + L) * T * e);
                if (past == 0) TRY(launch_vit_attention(m->dt, m->oqkv, m->octx, B, L, H, 0, s, hd, 1, m->gdt));
                else TRY(launch_generic_attention(m->dt, m->oqkv, 3 * T, (long)L * 3 * T, Ly.kc, T, (long)Lmax * T, Ly.vc, T, (long)Lmax * T, m->octx, T,
                                                  (long)L * T, B, L, past + L, H, hd, past, s, m->gdt));
            }
            TRY(tiled("opt_gemm_o", m->octx, Ly.w_o, Ly.s_o, T, T));
            TRY(launch_reduce_layernorm(m->gdt, m->dpart, 1, Ly.b_o, m->ox, Ly.ln2_g, Ly.ln2_b, c.t_eps, m->oh_t, nullptr, m->ox, R, T, s, true));
            TRY(tiled("opt_gemm_f1", m->oh_t, Ly.w_f1, Ly.s_f1, G, T));
            TRY(launch_reduce_bias_act(m->gdt, m->dpart, 1, Ly.b_f1, m->off, R, G, 2, s));
            const bool last = i + 1 == c.t_layers;
            TRY(tiled("opt_gemm_f2", m->off, Ly.w_f2, Ly.s_f2, T, G));
            TRY(launch_reduce_layernorm(m->gdt, m->dpart, 1, Ly.b_f2, m->ox, last ? m->o_lnf_g : m->ol[i + 1].ln1_g,
                                        last ? m->o_lnf_b : m->ol[i + 1].ln1_b, c.t_eps, m->oh_t, nullptr, m->ox, R, T, s, true));
        }
        return gemm(m, s, "opt_gemm_vocab", (const char*)m->oh_t + (size_t)(L - 1) * T * e, L * T, m->o_tok_t, T, m->logits, m->ldl, nullptr, nullptr, B,
                    c.vocab, T, 0, 1);
    }
    if (m->wq8) {
        // int8 weights: the prompt's rows go through the decode step's chain (the weight-streaming GEMMs take any row count: further
        // row groups re-read a unit's bytes from the XCD's L2; out_proj / fc2 as slice sums finished by the reduce + LayerNorm
        // consumer) with the prompt's causal attention in place of the cached one.  A one-crop prompt is 33 rows: a weight stream too.
        int S = 1;
        TRY(launch_layernorm(m->gdt, m->ox, T, m->ol[0].ln1_g, m->ol[0].ln1_b, c.t_eps, m->oh_t, nullptr, R, T, s));
        for (int i = 0; i < c.t_layers; ++i) {
            const OLayer& Ly = m->ol[i];
            TRY(opt_step_gemm(m, s, "opt_gemm_qkv", m->oh_t, Ly.w_qkv, Ly.b_qkv, 0, m->oqkv, R, 3 * T, T, false, &S, m->dt, Ly.s_qkv));
            TRY(launch_kv_append(m->dt, m->oqkv, Ly.kc, Ly.vc, B, L, T, Lmax, past, s));
            {
                ProfScope ps(m, s, "opt_attn", 4.0 * B * H * (double)L * (past + L) * hd, 2.0 * B * (past + L) * T * e);
                if (past == 0) TRY(launch_vit_attention(m->dt, m->oqkv, m->octx, B, L, H, 0, s, hd, 1, m->gdt));
                else TRY(launch_generic_attention(m->dt, m->oqkv, 3 * T, (long)L * 3 * T, Ly.kc, T, (long)Lmax * T, Ly.vc, T, (long)Lmax * T, m->octx, T,
                                                  (long)L * T, B, L, past + L, H, hd, past, s, m->gdt));
            }
            TRY(opt_step_gemm(m, s, "opt_gemm_o", m->octx, Ly.w_o, nullptr, 0, nullptr, R, T, T, true, &S, m->gdt, Ly.s_o));
            TRY(launch_reduce_layernorm(m->gdt, m->dpart, S, Ly.b_o, m->ox, Ly.ln2_g, Ly.ln2_b, c.t_eps, m->oh_t, nullptr, m->ox, R, T, s, true));
            TRY(opt_step_gemm(m, s, "opt_gemm_f1", m->oh_t, Ly.w_f1, Ly.b_f1, 2, m->off, R, G, T, false, &S, m->gdt, Ly.s_f1));
            const bool last = i + 1 == c.t_layers;
            TRY(opt_step_gemm(m, s, "opt_gemm_f2", m->off, Ly.w_f2, nullptr, 0, nullptr, R, T, G, true, &S, m->gdt, Ly.s_f2));
            TRY(launch_reduce_layernorm(m->gdt, m->dpart, S, Ly.b_f2, m->ox, last ? m->o_lnf_g : m->ol[i + 1].ln1_g,
                                        last ? m->o_lnf_b : m->ol[i + 1].ln1_b, c.t_eps, m->oh_t, nullptr, m->ox, R, T, s, true));
        }
        // oh_t = final LayerNorm of every row: the tied LM head (bf16, no bias) reads each image's last position
        return gemm(m, s, "opt_gemm_vocab", (const char*)m->oh_t + (size_t)(L - 1) * T * e, L * T, m->o_tok_t, T, m->logits, m->ldl, nullptr, nullptr, B,
                    c.vocab, T, 0, 1);
    }
    for (int i = 0; i < c.t_layers; ++i) {
        const OLayer& Ly = m->ol[i];
        TRY(launch_layernorm(m->gdt, m->ox, T, Ly.ln1_g, Ly.ln1_b, c.t_eps, m->oh_t, nullptr, R, T, s));
        TRY(gemm(m, s, "opt_gemm_qkv", m->oh_t, T, Ly.w_qkv, T, m->oqkv, 3 * T, Ly.b_qkv, nullptr, R, 3 * T, T, 0, af));
        TRY(launch_kv_append(m->dt, m->oqkv, Ly.kc, Ly.vc, B, L, T, Lmax, past, s));
        {
            ProfScope ps(m, s, "opt_attn", 4.0 * B * H * (double)L * (past + L) * hd, 2.0 * B * (past + L) * T * e);
            if (past == 0)      // the prompt: causal self-attention over the fused q|k|v rows (MFMA kernel for bf16 heads wider than 64)
                TRY(launch_vit_attention(m->dt, m->oqkv, m->octx, B, L, H, 0, s, hd, 1, m->gdt));
            else
                TRY(launch_generic_attention(m->dt, m->oqkv, 3 * T, (long)L * 3 * T, Ly.kc, T, (long)Lmax * T, Ly.vc, T, (long)Lmax * T, m->octx, T,
                                             (long)L * T, B, L, past + L, H, hd, past, s, m->gdt));
        }
        TRY(gemm(m, s, "opt_gemm_o", m->octx, T, Ly.w_o, T, m->ox, T, Ly.b_o, m->ox, R, T, T, 0, 1));
        TRY(launch_layernorm(m->gdt, m->ox, T, Ly.ln2_g, Ly.ln2_b, c.t_eps, m->oh_t, nullptr, R, T, s));
        TRY(gemm(m, s, "opt_gemm_f1", m->oh_t, T, Ly.w_f1, T, m->off, G, Ly.b_f1, nullptr, R, G, T, 2, 0));
        TRY(gemm(m, s, "opt_gemm_f2", m->off, G, Ly.w_f2, G, m->ox, T, Ly.b_f2, m->ox, R, T, G, 0, 1));
    }
    // final LayerNorm of each row's last new position, then the tied LM head (no bias)
    TRY(launch_layernorm(m->gdt, m->ox + (size_t)(L - 1) * T, L * T, m->o_lnf_g, m->o_lnf_b, c.t_eps, m->oh_t, nullptr, B, T, s));
    return gemm(m, s, "opt_gemm_vocab", m->oh_t, T, m->o_tok_t, T, m->logits, m->ldl, nullptr, nullptr, B, c.vocab, T, 0, 1);
}

// One decode step (one new position per row, x = ox [B, T] with its position added, oh_t = LayerNorm_1 of layer 0 already
// applied): at a few dozen rows every GEMM is a weight stream, so K is split over blocks (all slabs of a block in flight at
// once) and the consumers finish the sums - bias (+ ReLU) -> T for q|k|v and fc1; bias + residual + the NEXT LayerNorm for
// out_proj and fc2 (pre-LN blocks: what follows a residual add is always a LayerNorm).
// One projection of the decode step.  bf16 at a shape the weight-streaming kernel takes: that kernel, finished in place
// (ln == false: bias + act -> out_t) or as slice sums for the reduce+LayerNorm consumer.  Otherwise the tiled split-K GEMM
// with a reduce kernel.  The choice depends on dtype and (N, K) only - never on the row count.
int opt_step_gemm(Captioner* m, hipStream_t s, const char* tag, const void* A, const void* W, const float* bias, int act,
                  void* out_t, int B, int N, int K, bool ln, int* S_out, int out_dt,        // out_dt: type of out_t (split mode:
                  const float* wscale = nullptr) {                                         // fp32 for q|k|v, G8 for a GEMM operand)
    if (m->wq8) {                                       // int8 weights (load_in_8bit): W = fragment-ordered bytes, wscale = row scales
        const int S8 = skinny_i8_plan(N, K, !ln);
        ProfScope ps(m, s, tag, 2.0 * B * N * K, (double)B * K * 2 + (double)N * K + (ln ? (double)S8 * B * N * 4 : (double)B * N * 2));
        *S_out = S8;
        return launch_gemm_skinny_i8(A, K, W, wscale, bias, act, out_t, N, ln ? m->dpart : nullptr, B, N, K, s) == S8 ? 0 : -1;
    }
    const int S = m->dt == CAP_DT_BF16 ? skinny_plan(N, K, !ln) : 0;
    if (S >= 1) {
        ProfScope ps(m, s, tag, 2.0 * B * N * K, ((double)B * K + (double)N * K) * 2 + (ln ? (double)S * B * N * 4 : (double)B * N * 2));
        *S_out = S;
        return launch_gemm_skinny(A, K, W, K, bias, act, out_t, N, ln ? m->dpart : nullptr, B, N, K, s) == S ? 0 : -1;
    }
    TRY(gemm_partial(m, s, tag, A, W, m->dpart, B, N, K, 8, S_out));
    if (!ln) TRY(launch_reduce_bias_act(out_dt, m->dpart, *S_out, bias, out_t, B, N, act, s));
    return 0;
}

int run_opt_step(Captioner* m, int B, int past, hipStream_t s) {
    const CapConfig& c = m->c;
    const int T = c.t_hidden, G = c.t_ffn, H = c.t_heads, hd = T / H;
    const int Lmax = c.num_query_tokens + 1 + c.max_len;
    const size_t e = m->esz;
    int S = 1;
    for (int i = 0; i < c.t_layers; ++i) {
        const OLayer& Ly = m->ol[i];
        TRY(opt_step_gemm(m, s, "opt_gemm_qkv", m->oh_t, Ly.w_qkv, Ly.b_qkv, 0, m->oqkv, B, 3 * T, T, false, &S, m->dt, Ly.s_qkv));
        {
            ProfScope ps(m, s, "opt_attn", 4.0 * B * H * (double)(past + 1) * hd, 2.0 * B * (past + 1) * T * e);
            if (hd % 8 == 0 && hd <= 128)
                TRY(launch_opt_decode_attention(m->dt, m->oqkv, Ly.kc, Ly.vc, m->octx, B, T, H, Lmax, past, s, m->gdt));
            else {
                TRY(launch_kv_append(m->dt, m->oqkv, Ly.kc, Ly.vc, B, 1, T, Lmax, past, s));
                TRY(launch_generic_attention(m->dt, m->oqkv, 3 * T, 3 * T, Ly.kc, T, (long)Lmax * T, Ly.vc, T, (long)Lmax * T, m->octx, T, T, B, 1,
                                             past + 1, H, hd, past, s, m->gdt));
            }
        }
        TRY(opt_step_gemm(m, s, "opt_gemm_o", m->octx, Ly.w_o, nullptr, 0, nullptr, B, T, T, true, &S, m->gdt, Ly.s_o));
        TRY(launch_reduce_layernorm(m->gdt, m->dpart, S, Ly.b_o, m->ox, Ly.ln2_g, Ly.ln2_b, c.t_eps, m->oh_t, nullptr, m->ox, B, T, s, true));
        TRY(opt_step_gemm(m, s, "opt_gemm_f1", m->oh_t, Ly.w_f1, Ly.b_f1, 2, m->off, B, G, T, false, &S, m->gdt, Ly.s_f1));
        const bool last = i + 1 == c.t_layers;
        TRY(opt_step_gemm(m, s, "opt_gemm_f2", m->off, Ly.w_f2, nullptr, 0, nullptr, B, T, G, true, &S, m->gdt, Ly.s_f2));
        TRY(launch_reduce_layernorm(m->gdt, m->dpart, S, Ly.b_f2, m->ox, last ? m->o_lnf_g : m->ol[i + 1].ln1_g,
                                    last ? m->o_lnf_b : m->ol[i + 1].ln1_b, c.t_eps, m->oh_t, nullptr, m->ox, B, T, s, true));
    }
    return gemm(m, s, "opt_gemm_vocab", m->oh_t, T, m->o_tok_t, T, m->logits, m->ldl, nullptr, nullptr, B, c.vocab, T, 0, 1);
}

// HF Blip2ForConditionalGeneration.generate, greedy: out_ids [B, max_len] = the new tokens (pad after EOS), out_len [B] =
// their count incl. EOS, out_step_logits [max_len, B, vocab].
int run_generate_blip2(Captioner* m, const void* pixels, int fmt, int B, int max_len, int32_t* out_ids, int32_t* out_len,
                       float* out_step_logits, hipStream_t s) {
    const CapConfig& c = m->c;
    const int T = c.t_hidden, nq = c.num_query_tokens, P = nq + 1, Lmax = P + c.max_len;
    TRY(run_encoder(m, pixels, fmt, B, nullptr, s));
    TRY(run_qformer(m, B, s));
    TRY(gemm(m, s, "b2_gemm_lproj", m->qx_t, c.q_hidden, m->w_lproj, c.q_hidden, m->lm_proj, T, m->b_lproj, nullptr, B * nq, T, c.q_hidden, 0, 1));
    TRY(launch_opt_prefill_inputs(m->lm_proj, m->o_tok, m->o_pos, m->ox, B, nq, T, c.bos, s));
    hipLaunchKernelGGL(init_seq_kernel, dim3(64), dim3(256), 0, s, m->seq, m->finished, m->lens, B, Lmax, c.bos, c.pad);
    CAP_HIP_CHECK(hipGetLastError());
    TRY(run_opt(m, B, P, 0, s));
    for (int t = 0; t < max_len; ++t) {
        m->last_steps = t + 1;
        if (out_step_logits) {
            hipLaunchKernelGGL(copy_logits_kernel, dim3(1024), dim3(256), 0, s, m->logits, m->ldl,
                               out_step_logits + (size_t)t * B * c.vocab, B, c.vocab);
            CAP_HIP_CHECK(hipGetLastError());
        }
        // token of position P + t; a row finishes on EOS or at P + max_len tokens (greedy_select's `t` is the last filled index)
        TRY(launch_greedy_select(m->logits, m->ldl, c.vocab, m->seq, Lmax, P - 1 + t, P + max_len, c.eos, c.pad, m->finished, m->lens, B, s, 0, 0));
        if (t + 1 == max_len) break;
        {
            bool done;
            TRY(poll_all_finished(m, t, max_len, m->finished, B, nullptr, s, &done));
            if (done) break;
        }
        TRY(launch_opt_token_inputs(m->seq, Lmax, P + t, m->o_tok, m->o_pos, m->ox, B, T, s));
        TRY(launch_layernorm(m->gdt, m->ox, T, m->ol[0].ln1_g, m->ol[0].ln1_b, c.t_eps, m->oh_t, nullptr, B, T, s));
        TRY(run_opt_step(m, B, P + t, s));
    }
    hipLaunchKernelGGL(copy_new_tokens_kernel, dim3(64), dim3(256), 0, s, m->seq, Lmax, P, m->lens, out_ids, out_len, B, max_len);
    CAP_HIP_CHECK(hipGetLastError());
    return 0;
}

// ---------------------------------------------------------------------------------------------- sentence encoder
// HF BertModel state-dict names, as sentence-transformers stores all-MiniLM-L6-v2 (6 layers, 384 wide, 12 heads of 32,
// FFN 1536, post-LN, eps 1e-12).  Replaces `SentenceTransformer("all-MiniLM-L6-v2").encode(caption)` - reference
// agents/goal_exploration/goal_exploration.py:57,102 and detector/pseudolabeler.py:568,677.
int build_minilm(Captioner* m) {
    const CapConfig& c = m->c;
    const int T = c.t_hidden, F = c.t_ffn, V = c.vocab;
    TRY(reg_f32(m, "embeddings.word_embeddings.weight", &m->word_f32, (int64_t)V * T));
    TRY(reg_f32(m, "embeddings.position_embeddings.weight", &m->tpos, (int64_t)c.max_pos * T));
    TRY(walloc(m, (void**)&m->tok_type, (size_t)2 * T * 4));
    add_slot(m, "embeddings.token_type_embeddings.weight", m->tok_type, CAP_DT_F32, 2, T);
    TRY(reg_f32(m, "embeddings.LayerNorm.weight", &m->emb_g, T));
    TRY(reg_f32(m, "embeddings.LayerNorm.bias", &m->emb_b, T));
    m->tl.resize(c.t_layers);
    for (int i = 0; i < c.t_layers; ++i) {
        TLayer& L = m->tl[i];
        const std::string p = "encoder.layer." + std::to_string(i) + ".";
        TRY(walloc(m, &L.w_qkv, (size_t)3 * T * T * m->esz));
        TRY(walloc(m, (void**)&L.b_qkv, (size_t)3 * T * 4));
        const char* nm[3] = {"query", "key", "value"};
        for (int j = 0; j < 3; ++j) {
            add_slot(m, p + "attention.self." + nm[j] + ".weight", (char*)L.w_qkv + (size_t)j * T * T * m->esz, m->dt, T, T);
            add_slot(m, p + "attention.self." + nm[j] + ".bias", L.b_qkv + (size_t)j * T, CAP_DT_F32, 1, T);
        }
        TRY(reg_mat(m, p + "attention.output.dense.weight", &L.w_so, T, T));
        TRY(reg_f32(m, p + "attention.output.dense.bias", &L.b_so, T));
        TRY(reg_f32(m, p + "attention.output.LayerNorm.weight", &L.so_g, T));
        TRY(reg_f32(m, p + "attention.output.LayerNorm.bias", &L.so_b, T));
        TRY(reg_mat(m, p + "intermediate.dense.weight", &L.w_f1, F, T));
        TRY(reg_f32(m, p + "intermediate.dense.bias", &L.b_f1, F));
        TRY(reg_mat(m, p + "output.dense.weight", &L.w_f2, T, F));
        TRY(reg_f32(m, p + "output.dense.bias", &L.b_f2, T));
        TRY(reg_f32(m, p + "output.LayerNorm.weight", &L.f_g, T));
        TRY(reg_f32(m, p + "output.LayerNorm.bias", &L.f_b, T));
    }
    const size_t M = (size_t)c.max_batch * c.max_len, e = m->esz;
    TRY(dev_alloc(m, (void**)&m->te_x, M * T * 4));
    TRY(dev_alloc(m, (void**)&m->te_y, M * T * 4));
    TRY(dev_alloc(m, &m->te_xt, M * T * e));
    TRY(dev_alloc(m, &m->te_qkv, M * 3 * T * e));
    TRY(dev_alloc(m, &m->te_ctx, M * T * e));
    TRY(dev_alloc(m, &m->te_h, M * F * e));
    return 0;
}

// ids int32 [B, L] (padded rows: any valid id), lens int32 [B] (tokens incl. [CLS]/[SEP]) -> out fp32 [B, T]: mean of the
// last hidden states over the valid tokens, L2-normalised.
int run_text_encoder(Captioner* m, const int* ids, const int* lens, int B, int L, float* out, hipStream_t s) {
    const CapConfig& c = m->c;
    const int T = c.t_hidden, F = c.t_ffn, H = c.t_heads, M = B * L;
    {
        ProfScope ps(m, s, "te_embed", 0, (double)M * T * (12 + m->esz));
        TRY(launch_embed_tokens(m->dt, ids, L, m->word_f32, m->tpos, m->tok_type, m->emb_g, m->emb_b, c.t_eps, m->te_xt,
                                m->te_x, M, T, s, c.vocab));
    }
    for (int i = 0; i < c.t_layers; ++i) {
        const TLayer& Ly = m->tl[i];
        TRY(gemm(m, s, "te_gemm_qkv", m->te_xt, T, Ly.w_qkv, T, m->te_qkv, 3 * T, Ly.b_qkv, nullptr, M, 3 * T, T, 0, 0,
                 EPI_STORE, 0, 0, 0, 0, nullptr, nullptr));
        {
            ProfScope ps(m, s, "te_attention", 4.0 * B * H * (double)L * L * (T / H), (double)M * 4 * T * m->esz);
            TRY(launch_text_attention(m->dt, m->te_qkv, lens, m->te_ctx, B, L, H, T / H, s));
        }
        TRY(gemm(m, s, "te_gemm_o", m->te_ctx, T, Ly.w_so, T, m->te_y, T, Ly.b_so, m->te_x, M, T, T, 0, 1, EPI_STORE, 0, 0, 0,
                 0, nullptr, nullptr));
        {
            ProfScope ps(m, s, "te_layernorm", 0, (double)M * T * (8 + m->esz));
            TRY(launch_layernorm(m->dt, m->te_y, T, Ly.so_g, Ly.so_b, c.t_eps, m->te_xt, m->te_x, M, T, s));
        }
        TRY(gemm(m, s, "te_gemm_f1", m->te_xt, T, Ly.w_f1, T, m->te_h, F, Ly.b_f1, nullptr, M, F, T, 1, 0, EPI_STORE, 0, 0, 0,
                 0, nullptr, nullptr));
        TRY(gemm(m, s, "te_gemm_f2", m->te_h, F, Ly.w_f2, F, m->te_y, T, Ly.b_f2, m->te_x, M, T, F, 0, 1, EPI_STORE, 0, 0, 0,
                 0, nullptr, nullptr));
        {
            ProfScope ps(m, s, "te_layernorm", 0, (double)M * T * (8 + m->esz));
            TRY(launch_layernorm(m->dt, m->te_y, T, Ly.f_g, Ly.f_b, c.t_eps, m->te_xt, m->te_x, M, T, s));
        }
    }
    ProfScope ps(m, s, "te_pool", 0, (double)M * T * 4);
    return launch_mean_pool_normalize(m->te_x, lens, B, L, T, out, s);
}

// ---------------------------------------------------------------------------------------------- encoder
int run_encoder(Captioner* m, const void* pixels, int fmt, int B, float* out_embeds, hipStream_t s) {
    // BLIP: final LayerNorm = post_layernorm -> image_embeds (fp32 to the caller + T for the cross-K/V GEMM).
    // CoCa: ln_pre after the embeddings; final LayerNorm = the pooler's ln_k -> T only (run_coca_pool continues).
    const bool coca = m->c.arch == CAP_ARCH_COCA;
    const CapConfig& c = m->c;
    const int D = c.v_hidden, NT = m->NT, M = B * NT, H = c.v_heads;
    {
        ProfScope ps(m, s, "patchify", 0, (double)B * 3 * c.image_size * c.image_size * (fmt ? 1 : 4) + (double)B * m->P * m->Kpad * m->esz);
        TRY(launch_patchify(m->gdt, pixels, fmt, B, c.image_size, c.patch_size, m->Kpad, m->patches, c.pix_mean, c.pix_std, s));
    }
    TRY(gemm(m, s, "gemm_patch", m->patches, m->Kpad, m->w_patch, m->Kpad, m->X, D, m->b_patch, nullptr, B * m->P, D,
             m->Kpad, 0, 1, EPI_PATCH, m->P, 0, 0, 0, m->vpos));
    TRY(launch_cls_rows(m->cls, m->vpos, m->X, B, NT, D, s));
    if (coca) TRY(launch_layernorm(m->dt, m->X, D, m->ln_pre_g, m->ln_pre_b, c.v_eps, nullptr, m->X, M, D, s));
    // Pre-LN blocks.  Exact fp32 mode: the two branch GEMMs (proj, fc2) write their output to `delta`; the next LayerNorm kernel
    // folds it into the residual stream X (fp32) in the same pass that normalises it, so the GEMM epilogues are store-only.
    // Other modes: the branch GEMMs add into X themselves (vit_adds_in_place).
    const bool in_place = vit_adds_in_place(m->gdt);
    bool pending = false;                                  // delta holds a branch output not yet added to X
    auto add_ln = [&](const float* g, const float* b, void* out_t, float* out_f) -> int {
        ProfScope ps(m, s, "layernorm", 0, (double)M * D * ((pending ? 8 + (kDeltaInT ? m->esz : 4) : 4) + m->esz + (out_f ? 4 : 0)));
        if (pending)
            return launch_reduce_layernorm(m->gdt, m->delta, 1, nullptr, m->X, g, b, c.v_eps, out_t, out_f, m->X, M, D, s, false, kDeltaInT);
        return launch_layernorm(m->gdt, m->X, D, g, b, c.v_eps, out_t, out_f, M, D, s);
    };
    for (int i = 0; i < c.v_layers; ++i) {
        const VLayer& L = m->vl[i];
        TRY(add_ln(L.ln1_g, L.ln1_b, m->ln, nullptr));
        // split mode: q|k|v stay G8 when the split-fp16 MFMA attention kernel covers this token count, else the GEMM writes
        // fp32 for the fp32 attention kernels; either way the context comes out as G8 (the proj GEMM's operand)
        const bool g8_attn = m->gdt == CAP_DT_G8 && D / H == 64 && vit_attention_takes_g8(NT);
        TRY(gemm(m, s, "gemm_qkv", m->ln, D, L.w_qkv, D, m->qkv, 3 * D, L.b_qkv, nullptr, M, 3 * D, D, 0,
                 (m->gdt == CAP_DT_G8 && !g8_attn) ? 1 : 0));
        {
            ProfScope ps(m, s, "vit_attention", 4.0 * B * H * (double)NT * NT * 64, (double)M * 4 * D * m->esz);
            TRY(launch_vit_attention(g8_attn ? CAP_DT_G8 : m->dt, m->qkv, m->ctx, B, NT, H, 0, s, D / H, 0, m->gdt));
        }
        if (in_place) TRY(gemm(m, s, "gemm_proj", m->ctx, D, L.w_proj, D, m->X, D, L.b_proj, m->X, M, D, D, 0, 1));
        else TRY(gemm(m, s, "gemm_proj", m->ctx, D, L.w_proj, D, m->delta, D, L.b_proj, nullptr, M, D, D, 0, kDeltaInT ? 0 : 1));
        pending = !in_place;
        TRY(add_ln(L.ln2_g, L.ln2_b, m->ln, nullptr));
        TRY(gemm(m, s, "gemm_fc1", m->ln, D, L.w_fc1, D, m->mlp, c.v_mlp, L.b_fc1, nullptr, M, c.v_mlp, D, 1, 0));
        if (in_place) TRY(gemm(m, s, "gemm_fc2", m->mlp, c.v_mlp, L.w_fc2, c.v_mlp, m->X, D, L.b_fc2, m->X, M, D, c.v_mlp, 0, 1));
        else TRY(gemm(m, s, "gemm_fc2", m->mlp, c.v_mlp, L.w_fc2, c.v_mlp, m->delta, D, L.b_fc2, nullptr, M, D, c.v_mlp, 0, kDeltaInT ? 0 : 1));
    }
    if (coca) TRY(add_ln(m->lnk_g, m->lnk_b, m->emb_t, nullptr));
    else TRY(add_ln(m->post_g, m->post_b, m->emb_t, out_embeds ? out_embeds : m->emb_f));
    return 0;
}

// CoCa attentional pooler + ln_post, then the affine-free normalisation that feeds the folded cross-K/V projection.
// tokens_out (optional): fp32 [B, Q, E] = ln_post(pooler output); row 0 of each image is the pooled token, rows 1..Q-1
// are the image_embs the decoder cross-attends (reference coca_model.py:152-155).
int run_coca_pool(Captioner* m, int B, float* tokens_out, hipStream_t s) {
    const CapConfig& c = m->c;
    const int D = c.v_hidden, E = m->E, Q = m->Q, NT = m->NT;
    // split mode: k | v of the pooler are read by the attention kernel as fp32; its context is the out_proj GEMM's G8 operand
    TRY(gemm(m, s, "gemm_pool_kv", m->emb_t, D, m->w_pool_kv, D, m->pool_kvbuf, 2 * E, m->b_pool_kv, nullptr, B * NT, 2 * E, D, 0,
             m->gdt == CAP_DT_G8 ? 1 : 0));
    {
        ProfScope ps(m, s, "pool_attention", 4.0 * B * Q * (double)NT * E, (double)B * NT * 2 * E * m->esz);
        TRY(launch_pool_attention(m->dt, m->pool_q, m->pool_kvbuf, m->pool_ctx, B, NT, Q, E, c.pool_heads, s, m->gdt));
    }
    TRY(gemm(m, s, "gemm_pool_o", m->pool_ctx, E, m->w_pool_o, E, m->pool_o, E, m->b_pool_o, nullptr, B * Q, E, E, 0, 1));
    float* tok = tokens_out ? tokens_out : m->img_tokens;
    TRY(launch_layernorm(m->dt, m->pool_o, E, m->lnpost_g, m->lnpost_b, c.v_eps, nullptr, tok, B * Q, E, s));
    TRY(launch_layernorm(m->gdt, tok, E, m->ones, m->zeros, c.v_eps, m->xhat, nullptr, B * Q, E, s));
    return 0;
}

// ---------------------------------------------------------------------------------------------- decoder
// The decode state of a contiguous range of images [b0, b0+B) with R = B*K rows; all pointers are pre-offset, row indices
// inside the kernels are range-local.  (cap_generate decodes the whole batch as one range: row slices of ONE batch on their own
// streams measured level at 2 and slower at 3-4 - DESIGN.md section 4 - and were removed; whole batches overlap through
// engine.EnginePool instead.)
struct Dec {
    int b0, B, R, Btot;
    float *dx, *dy, *logits, *dpart;
    float* dx2;           // the fused paths' second LayerNorm row buffer
    char *dx_t, *dq, *dctx, *dh;
    int *seq, *finished, *lens, *anc;
    void* beam;
    size_t cache_off;     // byte offset of this slice's [k|v][R][H][Lm][64] block inside every layer's self cache
    RowMap map;           // compacted greedy loop: the open rows (live / count on the device); null pointers = every row
};

Dec make_slice(Captioner* m, int b0, int B, int Btot, int K, int Lm) {
    const CapConfig& c = m->c;
    const size_t T = c.t_hidden, F = c.t_ffn, H = c.t_heads, e = m->esz;
    const size_t r0 = (size_t)b0 * K;
    Dec d;
    d.b0 = b0; d.B = B; d.R = B * K; d.Btot = Btot;
    d.dx2 = m->dx2 + r0 * T;
    d.dx = m->dx + r0 * T; d.dy = m->dy + r0 * T; d.logits = m->logits + r0 * m->ldl; d.dpart = m->dpart + 12 * r0 * T;
    d.dx_t = (char*)m->dx_t + r0 * T * e; d.dq = (char*)m->dq + r0 * T * e; d.dctx = (char*)m->dctx + r0 * T * e;
    d.dh = (char*)m->dh + r0 * F * e;
    d.seq = m->seq + r0 * Lm; d.finished = m->finished + r0; d.lens = m->lens + r0; d.anc = m->anc + 2 * r0 * Lm;
    d.beam = m->beam;
    d.cache_off = 2 * r0 * H * Lm * 64 * e;
    return d;
}

// K slices of a decode GEMM: a function of (N, K, operand type) ONLY - never of the row count, so a caption's sums are ordered
// the same way alone and in a batch of 256.  bf16 / split mode run the "rows" kernel (gemm_rows_kernel: the block's K range
// over its four waves): as many slices as still give every wave a slab, keep the grid within one round of the CUs at the
// nominal 256 rows, and at most 4 (more slices write and re-read more partial sums than they save).  fp32 mode: the
// register-staged 64x64 tile with >= 3 slabs per slice.
int decode_splitk(const Captioner* m, int N, int K, int max_S) {
    if (m->gdt == CAP_DT_F32) {
        const int nk = K / 32;
        for (int cand : {4, 2})
            if (cand <= max_S && nk % cand == 0 && nk / cand >= 3) return cand;
        return 1;
    }
    // nominal row count the grid is sized for: a property of the ARCHITECTURE's deployment (BLIP / CoCa decode a few hundred
    // rows = images x beams at once; the OPT decoder of BLIP-2 a few dozen), never of the call
    const int plan_rows = m->c.arch == CAP_ARCH_BLIP2 ? 64 : 256;
    const int nk = K / (m->gdt == CAP_DT_BF16 ? 64 : 32), tiles = ((plan_rows + 63) / 64) * ((N + 63) / 64);
    for (int cand : {8, 6, 5, 4, 3, 2})      // (> 4 only where the caller allows it: the OPT decoder's weight streams at a few dozen rows)
        if (cand <= max_S && nk % cand == 0 && nk / cand >= 4 && tiles * cand <= 256) return cand;
    return 1;
}
inline int decode_tile(const Captioner* m) { return m->gdt == CAP_DT_F32 ? 2 : 6; }

int gemm_partial(Captioner* m, hipStream_t s, const char* tag, const void* A, const void* W, float* part, int R, int N,
                 int K, int max_S, int* S_out, const int* m_live) {
    const int S = decode_splitk(m, N, K, max_S);
    GemmParams p;
    memset(&p, 0, sizeof(p));
    p.A = A; p.lda = K; p.W = W; p.ldw = K; p.C = part; p.ldc = N; p.M = R; p.N = N; p.K = K;
    p.out_f32 = 1; p.epi = EPI_PARTIAL; p.splitk = S; p.m_live = m_live;
    *S_out = S;
    ProfScope ps(m, s, tag, 2.0 * R * N * K, ((double)R * K + (double)N * K) * m->esz + (double)S * R * N * 4);
    return launch_gemm(m->gdt, p, decode_tile(m), s);
}

// A finished decode projection (bias + activation -> operand type): fc1 of the text layers.  Same kernel family as the
// split-K ones at every row count.
int gemm_rows(Captioner* m, hipStream_t s, const char* tag, const void* A, const void* W, void* C, const float* bias, int R, int N,
              int K, int act, const int* m_live = nullptr) {
    GemmParams p;
    memset(&p, 0, sizeof(p));
    p.A = A; p.lda = K; p.W = W; p.ldw = K; p.C = C; p.ldc = N; p.bias = bias; p.ldr = N; p.M = R; p.N = N; p.K = K;
    p.gelu = act; p.out_f32 = 0; p.epi = EPI_STORE; p.splitk = 1; p.m_live = m_live;
    ProfScope ps(m, s, tag, 2.0 * R * N * K, ((double)R * K + (double)N * K + (double)R * N) * m->esz);
    return launch_gemm(m->gdt, p, m->gdt == CAP_DT_F32 ? 0 : 6, s);
}

// Decode-sized GEMM whose consumer is a LayerNorm: split K over S blocks per tile (every block's slabs are all in flight
// at once -> one memory round trip), partial sums to dpart, then the block-per-row kernel: y = sum + bias + resid (-> y_out)
// and LayerNorm(y) -> out_t / out_f.  (Running the consumer inside the GEMM kernel behind arrival counters cost more than the
// launch it saves - cross-XCD hand-over, DESIGN.md section 4 - and was removed.)
int gemm_splitk_reduce_ln(Captioner* m, hipStream_t s, const Dec& d, const char* tag, const void* A, const void* W,
                          const float* bias, const float* g, const float* b, float eps, int N, int K, void* out_t,
                          float* out_f, float* y_out, const float* resid = nullptr) {
    const int S = decode_splitk(m, N, K, 4);
    GemmParams p;
    memset(&p, 0, sizeof(p));
    p.A = A; p.lda = K; p.W = W; p.ldw = K; p.C = d.dpart; p.ldc = N; p.M = d.R; p.N = N; p.K = K;
    p.out_f32 = 1; p.epi = EPI_PARTIAL; p.splitk = S; p.m_live = d.map.n;
    {
        ProfScope ps(m, s, tag, 2.0 * d.R * N * K, ((double)d.R * K + (double)N * K) * m->esz + (double)S * d.R * N * 4);
        TRY(launch_gemm(m->gdt, p, decode_tile(m), s));
    }
    const bool per_row_block = d.R < (d.map.n ? 512 : 832) || N > 1024;
    ProfScope ps(m, s, per_row_block ? "dec_reduce_ln" : "dec_reduce_ln_wave", 0, (double)(S + 2) * d.R * N * 4 + (double)d.R * N * m->esz);
    // one 256-thread block per row up to several hundred rows (one memory round trip, latency-bound); at the pool's 1 024-row
    // passes the wave-per-row kernel.  Launches replayed from a captured graph (tools/bench_reduce_ln.py, us per launch, block / wave,
    // split mode | bf16 in place, the CoCa form): 512 rows 5.6 / 6.3 | 5.3 / 6.2, 640 (config 5: 128 images x 5 beams) 6.5 / 6.5 | 6.3 /
    // 6.6, 768 6.6 / 6.7 | 6.4 / 6.8, 896 7.6 / 7.0 | 7.3 / 7.0, 1 024 7.8 / 7.4 | 7.4 / 7.2: the crossing is between 768 and 896 (round 5
    // had put it at 512 from the 1 024-row figure alone, which cost config 5 ~4 % - dec_reduce_ln 17.6 -> 22.3 ms per step).  Same sums in
    // the same order - the two kernels give the same bits (tests/test_kernels_gpu.py::test_reduce_layernorm_kernels_agree_bit_for_bit),
    // so the row count may choose.  In the COMPACTED greedy loop (d.map: rows beyond the live count return at once) the wave kernel
    // keeps its round-5 threshold of 512 rows: a dead row costs it one wave's dispatch instead of four, and most steps of a 768-row
    // pass have far fewer live rows than that - same box, `bench.py --lite`, three interleaved pairs, 512 / 832: 6 476 / 6 460, 6 511 /
    // 6 476, 6 525 / 6 491 captions/s (+0.4 %).
    return launch_reduce_layernorm(m->gdt, d.dpart, S, bias, resid ? resid : d.dx, g, b, eps, out_t, out_f, y_out, d.R, N, s, per_row_block, false, d.map.n);
}

// x: the fp32 LayerNorm row buffer the consumer adds as its residual and replaces (d.dx, or d.dx2 on the fused paths)
int gemm_splitk_ln(Captioner* m, hipStream_t s, const Dec& d, const char* tag, const void* A, const void* W,
                   const float* bias, const float* g, const float* b, int N, int K, float* x = nullptr) {
    return gemm_splitk_reduce_ln(m, s, d, tag, A, W, bias, g, b, m->c.t_eps, N, K, d.dx_t, x ? x : d.dx, nullptr, x);
}

int run_decoder_step(Captioner* m, const Dec& d, const int* tokens, int tok_ld, int t, int K, const int* anc, int Lm,
                     hipStream_t s) {
    const CapConfig& c = m->c;
    const int T = c.t_hidden, F = c.t_ffn, H = c.t_heads, R = d.R, NT = m->NT;
    const size_t e = m->esz;
    // greedy: a caption that has ended (d.finished, set by greedy_select one step before) is not computed any more.  Compacted
    // (d.map, the merged passes of ~1 000 rows: at that size the projections are no longer a weight stream - 47 % of the decode
    // kernels' time): every kernel of the step works on the open captions' rows, packed to the front - activations, split-K
    // slabs and logits are indexed by the compact row, tokens / self-attention cache rows / the image's cross K/V through
    // map.live[c]; row tiles, rows and (row, head) units from *map.n on return at once.  Without a map (beams, per-step logits
    // wanted, forced off): the attention kernels skip ended rows in place and the GEMMs cover every row.
    const bool cm = d.map.n != nullptr;
    const int* skip = (K == 1 && !cm) ? d.finished : nullptr;
    TRY(launch_embed(m->gdt, tokens, tok_ld, t, m->word_f32, m->tpos, m->emb_g, m->emb_b, c.t_eps, d.dx_t, d.dx, R, T, s, nullptr, d.map));
    for (int i = 0; i < c.t_layers; ++i) {
        const TLayer& L = m->tl[i];
        char* kc = (char*)L.self_cache + d.cache_off;
        char* vc = kc + (size_t)R * H * Lm * 64 * e;
        if (t + 1 <= 32) {
            // q/k/v projection as split-K partial sums; the attention kernel finishes the reduction, appends k/v to the
            // cache and attends (one memory round trip per kernel instead of two in the GEMM)
            int S = 1;
            TRY(gemm_partial(m, s, "dec_gemm_qkv", d.dx_t, L.w_qkv, d.dpart, R, 3 * T, T, 4, &S, d.map.n));
            ProfScope ps(m, s, "dec_self_attn", 4.0 * R * H * (t + 1) * 64, 2.0 * R * H * (t + 1) * 64 * e + (double)S * R * 3 * T * 4);
            TRY(launch_decode_attention(m->dt, nullptr, kc, vc, anc, Lm, 1, Lm, t + 1, d.dctx, R, H, 0, s, d.dpart, S,
                                        L.b_qkv, 3 * T, 0, 1, m->gdt, skip, 0, 0, d.map));
        } else {
            if (cm) { cap_set_error("run_decoder_step: the compacted loop takes up to 32 positions"); return -1; }   // (run_generate never asks)
            TRY(gemm(m, s, "dec_gemm_qkv", d.dx_t, T, L.w_qkv, T, d.dq, T, L.b_qkv, nullptr, R, 3 * T, T, 0, 0, EPI_QKVCACHE,
                     R, H, Lm, t, nullptr, kc));
            ProfScope ps(m, s, "dec_self_attn", 4.0 * R * H * (t + 1) * 64, 2.0 * R * H * (t + 1) * 64 * e);
            TRY(launch_decode_attention(m->dt, d.dq, kc, vc, anc, Lm, 1, Lm, t + 1, d.dctx, R, H, 0, s, nullptr, 0, nullptr, 0, 0, 0, m->gdt, skip));
        }
        TRY(gemm_splitk_ln(m, s, d, "dec_gemm_so", d.dctx, L.w_so, L.b_so, L.so_g, L.so_b, T, T));
        {
            int S = 1;
            TRY(gemm_partial(m, s, "dec_gemm_cq", d.dx_t, L.w_cq, d.dpart, R, T, T, 4, &S, d.map.n));
            // beam-shared cross K/V of layer i: [k|v][image (whole batch)][head][token][64]; this slice starts at image b0.  A KV16
            // cache is addressed by row index inside the layer's k / v block: the kernel gets the block bases and the first row
            const size_t blk = m->cross_block((size_t)d.Btot * H * NT), row0 = (size_t)d.b0 * H * NT;
            const char* ck = (char*)m->cross + ((size_t)i * 2 + 0) * blk + (m->kv16 ? 0 : row0 * m->kvrow);
            const char* cv = (char*)m->cross + ((size_t)i * 2 + 1) * blk + (m->kv16 ? 0 : row0 * m->kvrow);
            ProfScope ps(m, s, "dec_cross_attn", 4.0 * R * H * NT * 64, 2.0 * d.B * H * NT * m->kvrow);
            TRY(launch_decode_attention(m->dt, nullptr, ck, cv, nullptr, 0, K, NT, NT, d.dctx, R, H, 0, s, d.dpart, S, L.b_cq,
                                        T, 0, 0, m->gdt, skip, m->kv16 ? 1 : 0, m->kv16 ? row0 : 0, d.map));
        }
        TRY(gemm_splitk_ln(m, s, d, "dec_gemm_co", d.dctx, L.w_co, L.b_co, L.co_g, L.co_b, T, T));
        TRY(gemm_rows(m, s, "dec_gemm_f1", d.dx_t, L.w_f1, d.dh, L.b_f1, R, F, T, 1, d.map.n));
        TRY(gemm_splitk_ln(m, s, d, "dec_gemm_f2", d.dh, L.w_f2, L.b_f2, L.f_g, L.f_b, T, F));
    }
    TRY(gemm(m, s, "dec_gemm_tr", d.dx_t, T, m->w_tr, T, d.dy, T, m->b_tr, nullptr, R, T, T, 1, 1));
    {
        ProfScope ps(m, s, "dec_layernorm", 0, (double)R * T * (8 + e));
        TRY(launch_layernorm(m->gdt, d.dy, T, m->tr_g, m->tr_b, c.t_eps, d.dx_t, d.dx, R, T, s));
    }
    TRY(gemm(m, s, "dec_gemm_vocab", d.dx_t, T, m->word_t, T, d.logits, m->ldl, m->b_vocab, nullptr, R, c.vocab, T, 0, 1));
    return 0;
}


// ---------------------------------------------------------------------------------------------- small-batch decoder step
// Up to SMALL_MAX_ROWS rows (the reference calls the captioner with ONE crop, BASELINE config 1 with 8): the same step as
// run_decoder_step in 6 launches per layer instead of 11 - every split-K consumer / LayerNorm and the self-attention run in the
// prologue of the kernel that needs their result, the cross-attention block (LayerNorm, query projection, attention) is one
// kernel per (row, head) - decode_small.hip.  The sums are those of the batch kernels (same K-slice plan, same chains, same
// LayerNorm / attention arithmetic): logits and tokens have the same bits on either path (tests/test_small_decode_gpu.py).
// The fp32 LayerNorm rows (the batch path's dx) alternate between d.dx and d.dx2: the one workgroup that writes a row never
// writes the buffer the others are still reading.
bool small_path_takes(const Captioner* m, const Dec& d, int t) {
    const CapConfig& c = m->c;
    if (m->gdt == CAP_DT_F32 || (c.arch != CAP_ARCH_BLIP && c.arch != CAP_ARCH_COCA) || !d.dx2) return false;
    if (d.R > SMALL_MAX_ROWS || t + 1 > 32) return false;
    const int T = c.arch == CAP_ARCH_COCA ? m->E : c.t_hidden, F = c.t_ffn, slab = m->gdt == CAP_DT_BF16 ? 64 : 32;
    if (T > 1024 || T != c.t_heads * 64 || T % 16 != 0 || F % 16 != 0 || T % slab != 0 || F % slab != 0) return false;
    // q|k|v partial sums sit beside the [<= 4][R][T] slabs of the other GEMMs in dpart (12 R T floats): at most 2 K slices
    return decode_splitk(m, 3 * T, T, 4) <= 2;
}

int run_decoder_step_small(Captioner* m, const Dec& d, const int* tokens, int tok_ld, int t, int K, const int* anc, int Lm,
                           hipStream_t s) {
    const CapConfig& c = m->c;
    const int T = c.t_hidden, F = c.t_ffn, H = c.t_heads, R = d.R, NT = m->NT;
    const size_t e = m->esz;
    const int* skip = K == 1 ? d.finished : nullptr;
    float* xb[2] = {d.dx, d.dx2};
    int cur = 0;                                   // xb[cur]: the LayerNorm row the next consumer adds as its residual
    float* qkvp = d.dpart + (size_t)4 * R * T;     // q|k|v partial sums, beside the [<= 4][R][T] slabs of the other GEMMs
    const int S_qkv = decode_splitk(m, 3 * T, T, 4), S_tt = decode_splitk(m, T, T, 4), S_f2 = decode_splitk(m, T, F, 4);
    const int kv_kind = m->kv16 ? SMALL_KV_KV16 : (m->dt == CAP_DT_BF16 ? SMALL_KV_BF16 : SMALL_KV_F32);
    TRY(launch_embed(m->gdt, tokens, tok_ld, t, m->word_f32, m->tpos, m->emb_g, m->emb_b, c.t_eps, d.dx_t, xb[0], R, T, s));
    SmallLN pend;                                  // the consumer the next kernel's prologue runs
    memset(&pend, 0, sizeof(pend));
    auto base = [&](const void* W, int N, int Kk, int S, int pro, int epi) {
        SmallGemm g;
        memset(&g, 0, sizeof(g));
        g.W = W; g.R = R; g.N = N; g.K = Kk; g.S = S; g.pro = pro; g.epi = epi; g.nchain = 4;
        return g;
    };
    auto consume = [&](SmallLN ln, bool keep) {    // bind the pending consumer to the current residual row / the other buffer
        ln.resid = xb[cur];
        ln.x_out = keep ? xb[cur ^ 1] : nullptr;
        if (keep) cur ^= 1;
        return ln;
    };
    for (int i = 0; i < c.t_layers; ++i) {
        const TLayer& L = m->tl[i];
        char* kc = (char*)L.self_cache + d.cache_off;
        char* vc = kc + (size_t)R * H * Lm * 64 * e;
        {
            SmallGemm g = base(L.w_qkv, 3 * T, T, S_qkv, i == 0 ? SMALL_PRO_GLOBAL : SMALL_PRO_LN, SMALL_EPI_PARTIAL);
            if (i == 0) g.A = d.dx_t; else g.ln = consume(pend, true);
            g.out_part = qkvp;
            ProfScope ps(m, s, "dec_small_qkv", 2.0 * R * 3 * T * T, ((double)R * T + 3.0 * T * T) * e + (double)S_qkv * R * 3 * T * 4);
            TRY(launch_small_gemm(m->gdt, g, s));
        }
        {
            SmallGemm g = base(L.w_so, T, T, S_tt, SMALL_PRO_SELFATTN, SMALL_EPI_PARTIAL);
            g.sa.qkv_part = qkvp; g.sa.qkv_bias = L.b_qkv; g.sa.qkv_S = S_qkv; g.sa.kc = kc; g.sa.vc = vc; g.sa.anc = anc; g.sa.anc_ld = Lm;
            g.sa.kv_ld = Lm; g.sa.n_keys = t + 1; g.sa.H = H; g.sa.skip = skip;
            g.out_part = d.dpart;
            ProfScope ps(m, s, "dec_small_so", 2.0 * R * T * T + 4.0 * R * H * (t + 1) * 64, ((double)R * T + (double)T * T) * e + (double)S_tt * R * T * 4);
            TRY(launch_small_gemm(m->gdt, g, s));
        }
        {
            SmallCross x;
            memset(&x, 0, sizeof(x));
            x.W = L.w_cq; x.bias = L.b_cq; x.R = R; x.D = T; x.H = H; x.S = S_tt;
            SmallLN ln;
            memset(&ln, 0, sizeof(ln));
            ln.part = d.dpart; ln.S = S_tt; ln.bias = L.b_so; ln.gamma = L.so_g; ln.beta = L.so_b; ln.eps = c.t_eps;
            x.ln = consume(ln, true);
            const size_t blk = m->cross_block((size_t)d.Btot * H * NT);
            x.kbase = (char*)m->cross + ((size_t)i * 2 + 0) * blk;
            x.vbase = (char*)m->cross + ((size_t)i * 2 + 1) * blk;
            x.kv_row0 = (size_t)d.b0 * H * NT;
            x.rows_per_kv = K; x.kv_ld = NT; x.n_keys = NT; x.kv_kind = kv_kind; x.skip = skip; x.out = d.dctx;
            ProfScope ps(m, s, "dec_small_cross", 2.0 * R * T * T + 4.0 * R * H * NT * 64, (double)T * T * e + 2.0 * R * H * NT * m->kvrow);
            TRY(launch_small_cross(m->gdt, x, s));
        }
        {
            SmallGemm g = base(L.w_co, T, T, S_tt, SMALL_PRO_GLOBAL, SMALL_EPI_PARTIAL);
            g.A = d.dctx; g.out_part = d.dpart;
            ProfScope ps(m, s, "dec_small_co", 2.0 * R * T * T, ((double)R * T + (double)T * T) * e + (double)S_tt * R * T * 4);
            TRY(launch_small_gemm(m->gdt, g, s));
        }
        {
            SmallGemm g = base(L.w_f1, F, T, 1, SMALL_PRO_LN, SMALL_EPI_ACT_T);
            SmallLN ln;
            memset(&ln, 0, sizeof(ln));
            ln.part = d.dpart; ln.S = S_tt; ln.bias = L.b_co; ln.gamma = L.co_g; ln.beta = L.co_b; ln.eps = c.t_eps;
            g.ln = consume(ln, true);
            g.bias = L.b_f1; g.act = 1; g.out = d.dh; g.ldc = F;
            ProfScope ps(m, s, "dec_small_f1", 2.0 * R * F * T, ((double)R * T + (double)F * T + (double)R * F) * e);
            TRY(launch_small_gemm(m->gdt, g, s));
        }
        {
            SmallGemm g = base(L.w_f2, T, F, S_f2, SMALL_PRO_GLOBAL, SMALL_EPI_PARTIAL);
            g.A = d.dh; g.out_part = d.dpart;
            ProfScope ps(m, s, "dec_small_f2", 2.0 * R * T * F, ((double)R * F + (double)T * F) * e + (double)S_f2 * R * T * 4);
            TRY(launch_small_gemm(m->gdt, g, s));
        }
        memset(&pend, 0, sizeof(pend));
        pend.part = d.dpart; pend.S = S_f2; pend.bias = L.b_f2; pend.gamma = L.f_g; pend.beta = L.f_b; pend.eps = c.t_eps;
    }
    {   // prediction head transform: LayerNorm of the last layer's FFN in the prologue, bias + GELU -> fp32
        SmallGemm g = base(m->w_tr, T, T, 1, SMALL_PRO_LN, SMALL_EPI_ACT_F32);
        g.nchain = 1;
        g.ln = consume(pend, false);
        g.bias = m->b_tr; g.act = 1; g.out = d.dy; g.ldc = T;
        ProfScope ps(m, s, "dec_small_tr", 2.0 * R * T * T, ((double)R * T + (double)T * T) * e + (double)R * T * 4);
        TRY(launch_small_gemm(m->gdt, g, s));
    }
    {   // vocabulary GEMM with the transform's LayerNorm in the prologue
        SmallGemm g = base(m->word_t, c.vocab, T, 1, SMALL_PRO_LN, SMALL_EPI_ACT_F32);
        g.nchain = 1;
        memset(&g.ln, 0, sizeof(g.ln));
        g.ln.part = d.dy; g.ln.S = 1; g.ln.gamma = m->tr_g; g.ln.beta = m->tr_b; g.ln.eps = c.t_eps;
        g.bias = m->b_vocab; g.act = 0; g.out = d.logits; g.ldc = m->ldl;
        ProfScope ps(m, s, "dec_small_vocab", 2.0 * R * c.vocab * T, ((double)R * T + (double)c.vocab * T) * e + (double)R * c.vocab * 4);
        TRY(launch_small_gemm(m->gdt, g, s));
    }
    return 0;
}

// ---------------------------------------------------------------------------------------------- CoCa decoder
// One KV-cached step through the unimodal text tower (t_layers causal blocks) and the multimodal decoder (mm_layers x
// [causal self-attention block, cross-attention block]).  Every block is pre-LN: the residual stream x stays fp32 in
// d.dx and each split-K consumer kernel both adds the branch to x and emits LayerNorm_next(x) as the next GEMM operand.
// The reference recomputes the whole prefix through both towers every step (coca_model.py:294-303).
// tokens [R, tok_ld]: newest token of every row at column t; K rows per image share the image's cross K/V; anc (beams): the
// ancestry table of the self-attention caches (row r's history position j was written by physical row anc[r][j]).
int run_coca_step(Captioner* m, const Dec& d, const int* tokens, int tok_ld, int t, int K, const int* anc, int Lm, hipStream_t s) {
    const CapConfig& c = m->c;
    const int E = m->E, F = c.t_ffn, H = c.t_heads, R = d.R, Q = m->Q;
    const size_t e = m->esz;
    const int nb = (int)m->cb.size();
    // x = tok_emb[token] + pos[t]  (raw sum to d.dx), ln = LayerNorm_{block0.ln_1}(x)
    TRY(launch_embed(m->gdt, tokens, tok_ld, t, m->tok_emb, m->tpos, m->cb[0].ln1_g, m->cb[0].ln1_b, c.t_eps, d.dx_t, nullptr, R, E, s,
                     d.dx));
    for (int bi = 0; bi < nb; ++bi) {
        const CBlock& b = m->cb[bi];
        const float* next_g = bi + 1 < nb ? m->cb[bi + 1].ln1_g : m->lnf_g;
        const float* next_b = bi + 1 < nb ? m->cb[bi + 1].ln1_b : m->lnf_b;
        int S = 1;
        if (!b.cross) {
            char* kc = (char*)m->ccache[b.cache] + d.cache_off;
            char* vc = kc + (size_t)R * H * Lm * 64 * e;
            TRY(gemm_partial(m, s, "coca_gemm_qkv", d.dx_t, b.w_in, d.dpart, R, 3 * E, E, 4, &S));
            ProfScope ps(m, s, "coca_self_attn", 4.0 * R * H * (t + 1) * 64, 2.0 * R * H * (t + 1) * 64 * e);
            TRY(launch_decode_attention(m->dt, nullptr, kc, vc, anc, Lm, 1, Lm, t + 1, d.dctx, R, H, 0, s, d.dpart, S, b.b_in,
                                        3 * E, 0, 1, m->gdt));
        } else {
            TRY(gemm_partial(m, s, "coca_gemm_cq", d.dx_t, b.w_in, d.dpart, R, E, E, 4, &S));
            // cross K/V of multimodal layer i: [k|v][image][head][Q tokens][64]; token 0 (the pooled token) is skipped
            const size_t blk = m->cross_block((size_t)d.Btot * H * Q), row0 = (size_t)d.b0 * H * Q + 1;
            const char* ck = (char*)m->cross + ((size_t)b.cross_idx * 2 + 0) * blk + (m->kv16 ? 0 : row0 * m->kvrow);
            const char* cv = (char*)m->cross + ((size_t)b.cross_idx * 2 + 1) * blk + (m->kv16 ? 0 : row0 * m->kvrow);
            ProfScope ps(m, s, "coca_cross_attn", 4.0 * R * H * (Q - 1) * 64, 2.0 * d.B * H * (Q - 1) * m->kvrow);
            TRY(launch_decode_attention(m->dt, nullptr, ck, cv, nullptr, 0, K, Q, Q - 1, d.dctx, R, H, 0, s, d.dpart, S, b.b_in, E,
                                        0, 0, m->gdt, nullptr, m->kv16 ? 1 : 0, m->kv16 ? row0 : 0));
        }
        // x += out_proj(ctx) ; ln = LayerNorm_2(x)
        TRY(gemm_splitk_reduce_ln(m, s, d, "coca_gemm_o", d.dctx, b.w_o, b.b_o, b.ln2_g, b.ln2_b, c.t_eps, E, E, d.dx_t, nullptr, d.dx));
        // x += c_proj(gelu(c_fc(ln))) ; ln = LayerNorm of the next block (or ln_final)
        TRY(gemm_rows(m, s, "coca_gemm_fc", d.dx_t, b.w_fc, d.dh, b.b_fc, R, F, E, 1));
        TRY(gemm_splitk_reduce_ln(m, s, d, "coca_gemm_pr", d.dh, b.w_pr, b.b_pr, next_g, next_b, c.t_eps, E, F, d.dx_t, nullptr, d.dx));
    }
    TRY(gemm(m, s, "coca_gemm_vocab", d.dx_t, E, m->w_cvocab, E, d.logits, m->ldl, nullptr, nullptr, R, c.vocab, E, 0, 1));
    return 0;
}


// The same step for up to SMALL_MAX_ROWS rows (the reference calls CoCa with ONE crop, coca.py:27-33): 4 launches per block instead
// of 7 - [consumer + LayerNorm] q|k|v GEMM, [self-attention] output projection, [consumer + LayerNorm] c_fc, c_proj; a
// cross-attention block starts with the (row, head) kernel (LayerNorm, query columns, attention) instead of the first two.  CoCa
// is pre-LN: the consumer's SUM is the new residual row (SmallLN::x_is_sum), its LayerNorm only feeds the next GEMM.  Same sums
// as run_coca_step (tests/test_small_decode_gpu.py).
int run_coca_step_small(Captioner* m, const Dec& d, const int* tokens, int tok_ld, int t, int K, const int* anc, int Lm, hipStream_t s) {
    const CapConfig& c = m->c;
    const int E = m->E, F = c.t_ffn, H = c.t_heads, R = d.R, Q = m->Q;
    const size_t e = m->esz;
    const int nb = (int)m->cb.size();
    float* xb[2] = {d.dx, d.dx2};
    int cur = 0;
    float* qkvp = d.dpart + (size_t)4 * R * E;
    const int S_qkv = decode_splitk(m, 3 * E, E, 4), S_ee = decode_splitk(m, E, E, 4), S_pr = decode_splitk(m, E, F, 4);
    const int kv_kind = m->kv16 ? SMALL_KV_KV16 : (m->dt == CAP_DT_BF16 ? SMALL_KV_BF16 : SMALL_KV_F32);
    TRY(launch_embed(m->gdt, tokens, tok_ld, t, m->tok_emb, m->tpos, m->cb[0].ln1_g, m->cb[0].ln1_b, c.t_eps, d.dx_t, nullptr, R, E, s,
                     xb[0]));
    SmallLN pend;
    memset(&pend, 0, sizeof(pend));
    auto base = [&](const void* W, int N, int Kk, int S, int pro, int epi) {
        SmallGemm g;
        memset(&g, 0, sizeof(g));
        g.W = W; g.R = R; g.N = N; g.K = Kk; g.S = S; g.pro = pro; g.epi = epi; g.nchain = 4;
        return g;
    };
    auto consume = [&](SmallLN ln, const float* gam, const float* bet, bool keep) {
        ln.gamma = gam; ln.beta = bet; ln.eps = c.t_eps; ln.x_is_sum = 1;
        ln.resid = xb[cur];
        ln.x_out = keep ? xb[cur ^ 1] : nullptr;
        if (keep) cur ^= 1;
        return ln;
    };
    for (int bi = 0; bi < nb; ++bi) {
        const CBlock& b = m->cb[bi];
        if (!b.cross) {
            char* kc = (char*)m->ccache[b.cache] + d.cache_off;
            char* vc = kc + (size_t)R * H * Lm * 64 * e;
            {
                SmallGemm g = base(b.w_in, 3 * E, E, S_qkv, bi == 0 ? SMALL_PRO_GLOBAL : SMALL_PRO_LN, SMALL_EPI_PARTIAL);
                if (bi == 0) g.A = d.dx_t; else g.ln = consume(pend, b.ln1_g, b.ln1_b, true);
                g.out_part = qkvp;
                ProfScope ps(m, s, "coca_small_qkv", 2.0 * R * 3 * E * E, ((double)R * E + 3.0 * E * E) * e + (double)S_qkv * R * 3 * E * 4);
                TRY(launch_small_gemm(m->gdt, g, s));
            }
            {
                SmallGemm g = base(b.w_o, E, E, S_ee, SMALL_PRO_SELFATTN, SMALL_EPI_PARTIAL);
                g.sa.qkv_part = qkvp; g.sa.qkv_bias = b.b_in; g.sa.qkv_S = S_qkv; g.sa.kc = kc; g.sa.vc = vc; g.sa.anc = anc; g.sa.anc_ld = Lm;
                g.sa.kv_ld = Lm; g.sa.n_keys = t + 1; g.sa.H = H; g.sa.skip = nullptr;
                g.out_part = d.dpart;
                ProfScope ps(m, s, "coca_small_o", 2.0 * R * E * E + 4.0 * R * H * (t + 1) * 64, ((double)R * E + (double)E * E) * e + (double)S_ee * R * E * 4);
                TRY(launch_small_gemm(m->gdt, g, s));
            }
        } else {
            {
                SmallCross x;
                memset(&x, 0, sizeof(x));
                x.W = b.w_in; x.bias = b.b_in; x.R = R; x.D = E; x.H = H; x.S = S_ee;
                x.ln = consume(pend, b.ln1_g, b.ln1_b, true);
                const size_t blk = m->cross_block((size_t)d.Btot * H * Q);
                x.kbase = (char*)m->cross + ((size_t)b.cross_idx * 2 + 0) * blk;
                x.vbase = (char*)m->cross + ((size_t)b.cross_idx * 2 + 1) * blk;
                x.kv_row0 = (size_t)d.b0 * H * Q + 1;                 // token 0 (the pooled token) is skipped
                x.rows_per_kv = K; x.kv_ld = Q; x.n_keys = Q - 1; x.kv_kind = kv_kind; x.skip = nullptr; x.out = d.dctx;
                ProfScope ps(m, s, "coca_small_cross", 2.0 * R * E * E + 4.0 * R * H * (Q - 1) * 64, (double)E * E * e + 2.0 * R * H * (Q - 1) * m->kvrow);
                TRY(launch_small_cross(m->gdt, x, s));
            }
            {
                SmallGemm g = base(b.w_o, E, E, S_ee, SMALL_PRO_GLOBAL, SMALL_EPI_PARTIAL);
                g.A = d.dctx; g.out_part = d.dpart;
                ProfScope ps(m, s, "coca_small_o", 2.0 * R * E * E, ((double)R * E + (double)E * E) * e + (double)S_ee * R * E * 4);
                TRY(launch_small_gemm(m->gdt, g, s));
            }
        }
        {
            SmallGemm g = base(b.w_fc, F, E, 1, SMALL_PRO_LN, SMALL_EPI_ACT_T);
            SmallLN ln;
            memset(&ln, 0, sizeof(ln));
            ln.part = d.dpart; ln.S = S_ee; ln.bias = b.b_o;
            g.ln = consume(ln, b.ln2_g, b.ln2_b, true);
            g.bias = b.b_fc; g.act = 1; g.out = d.dh; g.ldc = F;
            ProfScope ps(m, s, "coca_small_fc", 2.0 * R * F * E, ((double)R * E + (double)F * E + (double)R * F) * e);
            TRY(launch_small_gemm(m->gdt, g, s));
        }
        {
            SmallGemm g = base(b.w_pr, E, F, S_pr, SMALL_PRO_GLOBAL, SMALL_EPI_PARTIAL);
            g.A = d.dh; g.out_part = d.dpart;
            ProfScope ps(m, s, "coca_small_pr", 2.0 * R * E * F, ((double)R * F + (double)E * F) * e + (double)S_pr * R * E * 4);
            TRY(launch_small_gemm(m->gdt, g, s));
        }
        memset(&pend, 0, sizeof(pend));
        pend.part = d.dpart; pend.S = S_pr; pend.bias = b.b_pr;
    }
    {
        SmallGemm g = base(m->w_cvocab, c.vocab, E, 1, SMALL_PRO_LN, SMALL_EPI_ACT_F32);
        g.nchain = 1;
        g.ln = consume(pend, m->lnf_g, m->lnf_b, false);
        g.bias = nullptr; g.act = 0; g.out = d.logits; g.ldc = m->ldl;
        ProfScope ps(m, s, "coca_small_vocab", 2.0 * R * c.vocab * E, ((double)R * E + (double)c.vocab * E) * e + (double)R * c.vocab * 4);
        TRY(launch_small_gemm(m->gdt, g, s));
    }
    return 0;
}

__global__ void iota_rows_kernel(int* anc, int R, int L) {
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < 2 * R * L; i += gridDim.x * blockDim.x) anc[i] = (i / L) % R;
}
__global__ void init_seq_kernel(int* seq, int* fin, int* len, int R, int L, int bos, int pad) {
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < R * L; i += gridDim.x * blockDim.x) seq[i] = (i % L == 0) ? bos : pad;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < R; i += gridDim.x * blockDim.x) { fin[i] = 0; len[i] = L; }
}
__global__ void copy_logits_kernel(const float* src, int ld, float* dst, int R, int V) {
    const size_t n = (size_t)R * V;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        const size_t r = i / V, c = i - r * V;
        dst[i] = src[r * ld + c];
    }
}
__global__ void copy_i32_kernel(const int* s, int* d, size_t n) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) d[i] = s[i];
}

static int run_image_side(Captioner* m, const void* pixels, int fmt, int B, hipStream_t s) {
    const CapConfig& c = m->c;
    const int NT = m->NT, D = c.v_hidden, T = c.t_hidden, H = c.t_heads;
    TRY(run_encoder(m, pixels, fmt, B, nullptr, s));
    if (c.arch == CAP_ARCH_COCA) {
        TRY(run_coca_pool(m, B, nullptr, s));
        TRY(gemm(m, s, "gemm_crosskv", m->xhat, m->E, m->w_ckv, m->E, m->cross, 0, m->b_ckv, nullptr, B * m->Q,
                 c.mm_layers * 2 * m->E, m->E, 0, 0, EPI_CROSSKV, m->Q, H, B));
    } else {
        TRY(gemm(m, s, "gemm_crosskv", m->emb_t, D, m->w_ckv, D, m->cross, 0, m->b_ckv, nullptr, B * NT, c.t_layers * 2 * T, D,
                 0, 0, EPI_CROSSKV, NT, H, B));
    }
    return 0;
}

int run_generate(Captioner* m, const void* pixels, int fmt, int B, int K, int Lm, float lp, int32_t* out_ids,
                 int32_t* out_len, float* out_scores, float* out_step_logits, hipStream_t s, bool force_beam = false) {
    // force_beam: K == 1 runs as a 1-beam BEAM search (the scorer's bookkeeping, no forced EOS) instead of the greedy loop -
    // what a beam group of size one is (cap_generate_groups)
    const CapConfig& c = m->c;
    const int R = B * K;
    const bool coca = c.arch == CAP_ARCH_COCA;
    const bool greedy = K == 1 && !force_beam;
    TRY(run_image_side(m, pixels, fmt, B, s));
    Dec d = make_slice(m, 0, B, B, K, Lm);
    // Row compaction (ops.h, RowMap): the greedy BLIP loop on the batch kernels, when nobody asked for per-step logits (their rows
    // are the batch's rows) and every position takes the fused self-attention (<= 32: the k / v append goes through map.live)
    const bool compact = greedy && c.arch == CAP_ARCH_BLIP && m->compaction && m->live && !out_step_logits && Lm - 1 <= 32 &&
                         R > SMALL_MAX_ROWS && m->decode_path != 2;
    m->last_compacted = compact ? 1 : 0;
    if (greedy) {
        hipLaunchKernelGGL(init_seq_kernel, dim3(64), dim3(256), 0, s, d.seq, d.finished, d.lens, R, Lm, c.bos, c.pad);
        if (compact) {
            d.map.live = m->live; d.map.n = m->n_live;
            TRY(launch_compact_rows(d.finished, R, m->live, m->n_live, s));
        }
    } else {
        TRY(launch_beam_init(d.beam, B, K, Lm, c.bos, c.pad, c.eos, s, coca ? BEAM_LEGACY_RAW : BEAM_HF_V5));
        hipLaunchKernelGGL(iota_rows_kernel, dim3(64), dim3(256), 0, s, d.anc, R, Lm);
    }
    CAP_HIP_CHECK(hipGetLastError());
    if (m->decode_path == 2 && Lm - 1 > 32) {      // fail at entry, not after 32 steps have run
        cap_set_error("cap_generate: the small-batch decode path was forced (cap_set_decode_path 2) but max_len %d needs %d positions: "
                      "it takes at most 32 (automatic selection continues on the batch kernels from position 33)", Lm, Lm - 1);
        return -1;
    }
    for (int t = 0; t + 1 < Lm; ++t) {
        const int cur_len = t + 1;
        m->last_steps = t + 1;
        const int* tokens = greedy ? d.seq : beam_running_tokens_p(d.beam, B, K, Lm, cur_len & 1);
        const int* anc = greedy ? nullptr : d.anc + (size_t)(cur_len & 1) * R * Lm;
        {
            const bool can = small_path_takes(m, d, t);
            if (m->decode_path == 2 && !can) {
                cap_set_error("cap_generate: the small-batch decode path was forced (cap_set_decode_path 2) but does not take this call "
                              "(%d rows, step %d, compute type %d): at most %d rows, 32 positions, split or bf16 mode, BLIP / CoCa", R, t, m->gdt, SMALL_MAX_ROWS);
                return -1;
            }
            const bool small = can && (m->decode_path == 0 || m->decode_path == 2);
            m->last_path = small ? 2 : 1;
            if (coca) {
                if (small) TRY(run_coca_step_small(m, d, tokens, Lm, t, K, anc, Lm, s));
                else TRY(run_coca_step(m, d, tokens, Lm, t, K, anc, Lm, s));
            } else {
                if (small) TRY(run_decoder_step_small(m, d, tokens, Lm, t, K, anc, Lm, s));
                else TRY(run_decoder_step(m, d, tokens, Lm, t, K, anc, Lm, s));
            }
        }
        if (out_step_logits) {
            hipLaunchKernelGGL(copy_logits_kernel, dim3(1024), dim3(256), 0, s, d.logits, m->ldl,
                               out_step_logits + (size_t)t * R * c.vocab, R, c.vocab);
            CAP_HIP_CHECK(hipGetLastError());
        }
        ProfScope ps(m, s, greedy ? "greedy_select" : "beam_step", 0, (double)R * c.vocab * 4);
        if (greedy)
        {
            TRY(launch_greedy_select(d.logits, m->ldl, c.vocab, d.seq, Lm, t, Lm, c.eos, c.pad, d.finished, d.lens, R, s,
                                     coca ? c.min_len : 0, coca ? 1 : 0, d.map));
            if (compact) TRY(launch_compact_rows(d.finished, R, m->live, m->n_live, s));
        }
        else
            TRY(launch_beam_step(d.beam, d.logits, m->ldl, c.vocab, B, K, Lm, cur_len, c.eos, lp, d.anc, Lm, s,
                                 coca ? BEAM_LEGACY_RAW : BEAM_HF_V5, coca ? c.min_len : 0));
        bool done;
        TRY(poll_all_finished(m, t, Lm - 1, d.finished, R, greedy ? nullptr : beam_active_flag_p(d.beam, B, K, Lm), s, &done));
        if (done) break;
    }
    if (greedy) {
        hipLaunchKernelGGL(copy_i32_kernel, dim3(64), dim3(256), 0, s, d.seq, out_ids, (size_t)R * Lm);
        if (out_len) hipLaunchKernelGGL(copy_i32_kernel, dim3(4), dim3(256), 0, s, d.lens, out_len, (size_t)R);
        CAP_HIP_CHECK(hipGetLastError());
    } else {
        TRY(launch_beam_finalize(d.beam, B, K, Lm, out_ids, out_len, out_scores, s));
    }
    return 0;
}

}  // namespace

// Everything a handle owns - also the exit of every failed cap_create (streams and events included).
static void release_captioner(Captioner* m) {
    for (auto& r : m->prof_recs) { (void)hipEventDestroy(r.e0); (void)hipEventDestroy(r.e1); }
    for (void* p : m->allocs) (void)hipFree(p);
    if (m->ws && m->ws->refs.fetch_sub(1) == 1) {        // last handle on these weights
        for (void* p : m->ws->ptrs) (void)hipFree(p);
        delete m->ws;
    }
    if (m->stage) (void)hipFree(m->stage);
    if (m->absmax_dev) (void)hipFree(m->absmax_dev);
    if (m->host_flag) (void)hipHostFree(m->host_flag);
    delete m;
}

// ================================================================================================ C ABI
extern "C" {

const char* cap_last_error(void) { return g_err; }
int cap_version(void) { return 1; }

static int create_impl(const CapConfig* cfg, Captioner* share, CapHandle* out) {
    if (!cfg || !out) { cap_set_error("cap_create: null argument"); return -1; }
    if (cfg->struct_size != (int)sizeof(CapConfig)) {
        cap_set_error("cap_create: CapConfig size mismatch (caller %d, library %d)", cfg->struct_size, (int)sizeof(CapConfig));
        return -1;
    }
    if (cfg->arch != CAP_ARCH_BLIP && cfg->arch != CAP_ARCH_COCA && cfg->arch != CAP_ARCH_MINILM && cfg->arch != CAP_ARCH_BLIP2) {
        cap_set_error("cap_create: unknown arch %d", cfg->arch);
        return -1;
    }
    if (cfg->compute_dtype != CAP_F32 && cfg->compute_dtype != CAP_BF16 && cfg->compute_dtype != CAP_F32_SPLIT) {
        cap_set_error("cap_create: unknown dtype");
        return -1;
    }
    if (cfg->compute_dtype == CAP_F32_SPLIT && cfg->arch == CAP_ARCH_MINILM) {
        cap_set_error("cap_create: CAP_F32_SPLIT is built for the captioner architectures (the sentence encoder takes CAP_F32 or CAP_BF16)");
        return -1;
    }
    const bool text_only = cfg->arch == CAP_ARCH_MINILM;
    if (text_only) {
        const int hd = cfg->t_heads > 0 ? cfg->t_hidden / cfg->t_heads : 0;
        if ((hd != 32 && hd != 64) || hd * cfg->t_heads != cfg->t_hidden || cfg->t_hidden % 64 || cfg->t_ffn % 64 ||
            cfg->t_hidden > 1024 || cfg->t_layers < 1 || cfg->max_batch < 1 || cfg->max_len < 1 || cfg->max_len > cfg->max_pos ||
            cfg->max_len > 512 || cfg->vocab < 1) {
            cap_set_error("cap_create: sentence encoder needs head_dim 32 or 64, widths multiple of 64 (hidden <= 1024), "
                          "1 <= max_len <= min(max_pos, 512)");
            return -1;
        }
    } else if (cfg->arch == CAP_ARCH_BLIP2) {
        auto hd_ok = [](int w, int h) { return h > 0 && w % h == 0 && (w / h) % 8 == 0 && w / h >= 8 && w / h <= 128; };
        if (!hd_ok(cfg->v_hidden, cfg->v_heads) || !hd_ok(cfg->q_hidden, cfg->q_heads) || !hd_ok(cfg->t_hidden, cfg->t_heads) ||
            cfg->v_hidden % 64 || cfg->v_mlp % 64 || cfg->q_hidden % 64 || cfg->q_ffn % 64 || cfg->t_hidden % 64 || cfg->t_ffn % 64 ||
            cfg->t_hidden > 256 * 12 || cfg->image_size % cfg->patch_size || cfg->q_layers < 1 || cfg->q_cross_freq < 1 ||
            cfg->num_query_tokens < 1 || cfg->max_batch < 1 || cfg->max_beams != 1 || cfg->max_len < 1 ||
            cfg->num_query_tokens + 1 + cfg->max_len > cfg->max_pos) {
            cap_set_error("cap_create: BLIP-2 needs head dims that are multiples of 8 (<= 128), widths multiple of 64 (OPT hidden <= 3072), "
                          "max_beams 1 and num_query_tokens + 1 + max_len <= max_pos");
            return -1;
        }
    } else {
        if (cfg->arch == CAP_ARCH_COCA) {
            const int hd = cfg->pool_heads > 0 ? cfg->embed_dim / cfg->pool_heads : 0;
            if (cfg->embed_dim != cfg->t_hidden || cfg->pool_queries < 2 || cfg->mm_layers < 1 || (hd != 64 && hd != 96) ||
                hd * cfg->pool_heads != cfg->embed_dim) {
                cap_set_error("cap_create: CoCa needs embed_dim == t_hidden, pooler head_dim 64 or 96, mm_layers >= 1");
                return -1;
            }
        }
        if (cfg->v_hidden != cfg->v_heads * 64 || cfg->t_hidden != cfg->t_heads * 64) {
            cap_set_error("cap_create: head_dim must be 64 (v %d/%d, t %d/%d)", cfg->v_hidden, cfg->v_heads, cfg->t_hidden, cfg->t_heads);
            return -1;
        }
        if (cfg->image_size % cfg->patch_size || cfg->max_batch < 1 || cfg->max_beams < 1 || cfg->max_beams > 8 ||
            cfg->max_len < 2 || cfg->max_len > cfg->max_pos) {
            cap_set_error("cap_create: bad geometry/capacity");
            return -1;
        }
        if (cfg->v_hidden % 64 || cfg->v_mlp % 64 || cfg->t_ffn % 64) { cap_set_error("cap_create: widths must be multiples of 64"); return -1; }
    }
    if (cfg->weight_int8) {
        const int T = cfg->t_hidden, G = cfg->t_ffn;
        if (cfg->weight_int8 != 1 || cfg->arch != CAP_ARCH_BLIP2 || cfg->compute_dtype != CAP_BF16) {
            cap_set_error("cap_create: weight_int8 (load_in_8bit) is built for CAP_ARCH_BLIP2 with CAP_BF16 activations");
            return -1;
        }
        if (skinny_i8_plan(3 * T, T, true) < 1 || skinny_i8_plan(G, T, true) < 1 || skinny_i8_plan(T, T, false) < 1 || skinny_i8_plan(T, G, false) < 1) {
            cap_set_error("cap_create: weight_int8 needs OPT widths the int8 weight stream takes (hidden %d, ffn %d: multiples of 256 that "
                          "split into waves of at most 320 k)", T, G);
            return -1;
        }
    }
    int dev = 0;
    CAP_HIP_CHECK(hipGetDevice(&dev));
    if (share) {
        // same model, same arithmetic, same GPU; only the capacity of the arena may differ
        CapConfig a = *cfg, b = share->c;
        a.max_batch = b.max_batch = 0; a.max_beams = b.max_beams = 0; a.max_len = b.max_len = 0;
        if (memcmp(&a, &b, sizeof(a)) != 0 || share->ws->device != dev) {
            cap_set_error("cap_create_shared: the new handle must describe the same model, compute dtype and GPU as the handle "
                          "whose weights it shares");
            return -1;
        }
    }
    Captioner* m = new Captioner();
    m->c = *cfg;
    m->wq8 = cfg->weight_int8 != 0;
    if (share) { m->ws = share->ws; m->ws->refs.fetch_add(1); m->replay = true; }
    else { m->ws = new WeightStore(); m->ws->device = dev; }
    m->dt = cfg->compute_dtype == CAP_BF16 ? CAP_DT_BF16 : CAP_DT_F32;
    m->gdt = cfg->compute_dtype == CAP_F32_SPLIT ? CAP_DT_G8 : m->dt;
    m->esz = m->dt == CAP_DT_BF16 ? 2 : 4;            // a G8 element is 4 bytes like fp32
    const int g = text_only ? 0 : cfg->image_size / cfg->patch_size;
    // (the KV16 layout is read by the chunked cross-attention kernels: more than 32 keys per image - every real geometry; the
    // fixture-sized ones keep fp32 rows)
    m->kv16 = m->gdt == CAP_DT_G8 && !cfg->cross_kv_fp32 &&
              ((cfg->arch == CAP_ARCH_BLIP && g * g + 1 > 32) || (cfg->arch == CAP_ARCH_COCA && cfg->pool_queries - 1 > 32));
    m->kvrow = m->kv16 ? 132 : 64 * m->esz;
    m->P = g * g; m->NT = m->P + 1;
    m->Kpatch = text_only ? 0 : 3 * cfg->patch_size * cfg->patch_size;
    m->Kpad = (m->Kpatch + 63) / 64 * 64;
    const int built = text_only ? (build_minilm(m) != 0)
                      : cfg->arch == CAP_ARCH_BLIP2 ? (build_blip2(m) != 0)
                      : cfg->arch == CAP_ARCH_COCA ? (build_coca(m) != 0 || build_arena_coca(m) != 0)
                                                   : (build_blip(m) != 0 || build_arena(m) != 0);
    if (built) {
        release_captioner(m);       // the message of the failing step is kept
        return -1;
    }
    if (m->replay && m->wcur != m->ws->ptrs.size()) {
        cap_set_error("cap_create_shared: the shared store holds %zu buffers, this configuration uses %zu", m->ws->ptrs.size(), m->wcur);
        release_captioner(m);
        return -1;
    }
    *out = (CapHandle)m;
    return 0;
}

int cap_create(const CapConfig* cfg, CapHandle* out) { return create_impl(cfg, nullptr, out); }

int cap_create_shared(const CapConfig* cfg, CapHandle weights_of, CapHandle* out) {
    if (!weights_of) { cap_set_error("cap_create_shared: null handle"); return -1; }
    return create_impl(cfg, (Captioner*)weights_of, out);
}

int cap_destroy(CapHandle h) {
    if (!h) return 0;
    (void)hipDeviceSynchronize();
    release_captioner((Captioner*)h);
    return 0;
}

int cap_set_early_exit(CapHandle h, int poll_steps) {
    Captioner* m = (Captioner*)h;
    if (!m || poll_steps < 0) { cap_set_error("cap_set_early_exit: null handle or negative interval"); return -1; }
    if (poll_steps > 0 && !m->host_flag) {
        CAP_HIP_CHECK(hipHostMalloc((void**)&m->host_flag, sizeof(int), hipHostMallocMapped));
        CAP_HIP_CHECK(hipHostGetDevicePointer((void**)&m->host_flag_dev, m->host_flag, 0));
        *m->host_flag = 1;
    }
    m->poll = poll_steps;
    return 0;
}

int cap_last_decode_steps(CapHandle h) { return h ? ((Captioner*)h)->last_steps : -1; }

int cap_set_decode_path(CapHandle h, int path) {
    Captioner* m = (Captioner*)h;
    if (!m) { cap_set_error("cap_set_decode_path: null handle"); return -1; }
    if (path < 0 || path > 2) { cap_set_error("cap_set_decode_path: path must be 0 (by row count), 1 (batch kernels, one launch per operation) or 2 (small-batch kernels), got %d", path); return -1; }
    m->decode_path = path;
    return 0;
}
int cap_last_decode_path(CapHandle h) { return h ? ((Captioner*)h)->last_path : -1; }

int cap_set_row_compaction(CapHandle h, int on) {
    Captioner* m = (Captioner*)h;
    if (!m || (on != 0 && on != 1)) { cap_set_error("cap_set_row_compaction: null handle, or a value other than 0 / 1"); return -1; }
    m->compaction = on;
    return 0;
}
int cap_last_row_compaction(CapHandle h) { return h ? ((Captioner*)h)->last_compacted : -1; }
int cap_cross_cache_kind(CapHandle h) {
    const Captioner* m = (const Captioner*)h;
    if (!m) return -1;
    return m->kv16 ? 2 : (m->dt == CAP_DT_BF16 ? 1 : 0);
}

size_t cap_device_bytes(CapHandle h) { return h ? ((Captioner*)h)->dev_bytes : 0; }

int cap_load_weight(CapHandle h, const char* name, const float* data, int on_device, int ndim, const int64_t* shape,
                    void* stream) {
    Captioner* m = (Captioner*)h;
    if (!m || !name || !data) { cap_set_error("cap_load_weight: null argument"); return -1; }
    hipStream_t s = (hipStream_t)stream;
    int64_t n = 1;
    for (int i = 0; i < ndim; ++i) n *= shape[i];
    auto range = m->ws->slots.equal_range(name);
    if (range.first == range.second) return 1;   // not a tensor this architecture stores (e.g. tied decoder.weight)
    const float* src = data;
    if (!on_device) {
        if ((size_t)n > m->stage_elems) {
            if (m->stage) { CAP_HIP_CHECK(hipStreamSynchronize(s)); (void)hipFree(m->stage); }
            CAP_HIP_CHECK(hipMalloc((void**)&m->stage, (size_t)n * 4));
            m->stage_elems = n;
        }
        CAP_HIP_CHECK(hipMemcpyAsync(m->stage, data, (size_t)n * 4, hipMemcpyHostToDevice, s));
        src = m->stage;
    }
    bool any_g8 = false;
    for (auto it = range.first; it != range.second; ++it) any_g8 |= it->second.dtype == CAP_DT_G8;
    if (any_g8) {
        // split mode stores 4096 * w as two fp16 halves: a tensor that does not fit fp16's range would be clipped silently -
        // refuse it instead (real checkpoints sit far below the bound; this is the guard for the ones that do not)
        if (!m->absmax_dev) CAP_HIP_CHECK(hipMalloc((void**)&m->absmax_dev, 256));
        CAP_HIP_CHECK(hipMemsetAsync(m->absmax_dev, 0, 4, s));
        TRY(launch_absmax_f32(src, (size_t)n, m->absmax_dev, s));
        unsigned int bits = 0;
        CAP_HIP_CHECK(hipMemcpyAsync(&bits, m->absmax_dev, 4, hipMemcpyDeviceToHost, s));
        CAP_HIP_CHECK(hipStreamSynchronize(s));
        float amax;
        memcpy(&amax, &bits, 4);
        if (!(amax * G8_WSCALE <= G8_AMAX)) {
            cap_set_error("cap_load_weight: %s has max |w| = %g; the split-fp16 mode (CAP_F32_SPLIT / dtype \"f32s\") stores "
                          "%g * w in fp16 and takes |w| <= %g (no NaN) - load this checkpoint with CAP_F32 or CAP_BF16",
                          name, (double)amax, (double)G8_WSCALE, (double)(G8_AMAX / G8_WSCALE));
            return -1;
        }
    }
    for (auto it = range.first; it != range.second; ++it) {
        Slot& sl = it->second;
        if (sl.rows * sl.cols != n) {
            cap_set_error("cap_load_weight: %s has %lld elements, expected %lld x %lld", name, (long long)n,
                          (long long)sl.rows, (long long)sl.cols);
            return -1;
        }
        if (sl.dtype == CAP_DT_I8W) TRY(launch_quant_i8_pack(src, sl.dst, (float*)sl.aux, (int)sl.rows, (int)sl.cols, s));
        else TRY(launch_convert2d(sl.dtype, src, sl.dst, (int)sl.rows, (int)sl.cols, sl.dst_ld, s, sl.dtype == CAP_DT_G8 ? G8_WSCALE : 1.0f));
        sl.loaded = true;
    }
    CAP_HIP_CHECK(hipStreamSynchronize(s));
    return 0;
}

int cap_finalize_weights(CapHandle h) {
    Captioner* m = (Captioner*)h;
    if (!m) { cap_set_error("null handle"); return -1; }
    int missing = 0;
    std::string names;
    for (auto& kv : m->ws->slots)
        if (!kv.second.loaded) {
            if (missing < 8) names += (missing ? ", " : "") + kv.first;
            ++missing;
        }
    if (missing) cap_set_error("%d tensors not loaded: %s%s", missing, names.c_str(), missing > 8 ? ", ..." : "");
    return missing;
}

static int check_call(Captioner* m, int B, int K, int Lm, int fmt) {
    if (!m) { cap_set_error("null handle"); return -1; }
    if (cap_finalize_weights((CapHandle)m) != 0) return -1;
    if (B < 1 || B > m->c.max_batch || K < 1 || K > m->c.max_beams || Lm < (m->c.arch == CAP_ARCH_BLIP2 ? 1 : 2) || Lm > m->c.max_len) {
        cap_set_error("request B=%d beams=%d max_len=%d exceeds the handle's capacity (%d, %d, %d)", B, K, Lm,
                      m->c.max_batch, m->c.max_beams, m->c.max_len);
        return -1;
    }
    if (fmt != CAP_PIX_F32_NCHW && fmt != CAP_PIX_U8_NHWC) { cap_set_error("unknown pixel format %d", fmt); return -1; }
    if (m->c.arch == CAP_ARCH_MINILM) { cap_set_error("this handle is a sentence encoder: use cap_embed_text"); return -1; }
    return 0;
}

int cap_embed_text(CapHandle h, const int32_t* ids, const int32_t* lens, int B, int L, float* out, void* stream) {
    Captioner* m = (Captioner*)h;
    if (!m) { cap_set_error("null handle"); return -1; }
    if (m->c.arch != CAP_ARCH_MINILM) { cap_set_error("cap_embed_text: the handle is not a sentence encoder"); return -1; }
    if (cap_finalize_weights(h) != 0) return -1;
    if (!ids || !lens || !out) { cap_set_error("cap_embed_text: null buffer"); return -1; }
    if (B < 1 || B > m->c.max_batch || L < 1 || L > m->c.max_len) {
        cap_set_error("cap_embed_text: B=%d L=%d exceeds the handle's capacity (%d, %d)", B, L, m->c.max_batch, m->c.max_len);
        return -1;
    }
    return run_text_encoder(m, ids, lens, B, L, out, (hipStream_t)stream);
}

int cap_encode(CapHandle h, const void* pixels, int pixel_fmt, int B, float* out_embeds, void* stream) {
    Captioner* m = (Captioner*)h;
    TRY(check_call(m, B, 1, 2, pixel_fmt));
    if (!pixels || !out_embeds) { cap_set_error("cap_encode: null buffer"); return -1; }
    if (m->c.arch == CAP_ARCH_COCA) {    // out_embeds: fp32 [B, pool_queries, embed_dim] (row 0 pooled token, rows 1.. image_embs)
        TRY(run_encoder(m, pixels, pixel_fmt, B, nullptr, (hipStream_t)stream));
        return run_coca_pool(m, B, out_embeds, (hipStream_t)stream);
    }
    return run_encoder(m, pixels, pixel_fmt, B, out_embeds, (hipStream_t)stream);
}

int cap_generate(CapHandle h, const void* pixels, int pixel_fmt, int B, int num_beams, int max_len, float length_penalty,
                 int32_t* out_ids, int32_t* out_len, float* out_scores, float* out_step_logits, void* stream) {
    Captioner* m = (Captioner*)h;
    TRY(check_call(m, B, num_beams, max_len, pixel_fmt));
    if (!pixels || !out_ids) { cap_set_error("cap_generate: null buffer"); return -1; }
    if (m->c.arch == CAP_ARCH_BLIP2) {
        if (num_beams != 1) { cap_set_error("cap_generate: BLIP-2 supports greedy decoding (num_beams = 1)"); return -1; }
        return run_generate_blip2(m, pixels, pixel_fmt, B, max_len, out_ids, out_len, out_step_logits, (hipStream_t)stream);
    }
    // CoCa: num_beams == 1 is the reference's top-k(1) loop (coca.py:29), num_beams > 1 its `_generate_beamsearch` with one
    // beam group (coca_model.py:335-482; length_penalty is the scorer's: pass 1.0 for the reference's default)
    return run_generate(m, pixels, pixel_fmt, B, num_beams, max_len, length_penalty, out_ids, out_len, out_scores,
                        out_step_logits, (hipStream_t)stream);
}

int cap_generate_groups(CapHandle h, const void* pixels, int pixel_fmt, int B, int num_beams, int num_beam_groups, int max_len,
                        float length_penalty, int32_t* out_ids, int32_t* out_len, float* out_scores, void* stream) {
    Captioner* m = (Captioner*)h;
    TRY(check_call(m, B, num_beams, max_len, pixel_fmt));
    if (!pixels || !out_ids) { cap_set_error("cap_generate_groups: null buffer"); return -1; }
    if (m->c.arch != CAP_ARCH_COCA) {
        cap_set_error("cap_generate_groups: beam groups are the CoCa loop's (coca_model.py:335-482); HF's group beam search for the "
                      "other architectures needs a diversity penalty, which this library does not implement");
        return -1;
    }
    if (num_beam_groups < 1 || num_beam_groups > num_beams || num_beams % num_beam_groups != 0) {
        cap_set_error("cap_generate_groups: num_beams (%d) must be a multiple of num_beam_groups (%d) (BeamSearchScorer's own check)",
                      num_beams, num_beam_groups);
        return -1;
    }
    // The reference's loop runs its groups one after the other on the SAME logits with only MinLength / RepetitionPenalty(1.0)
    // as processors (coca_model.py:236-241: no HammingDiversity processor), every group starts from the same scores (:380-384),
    // and finalize picks the best hypothesis over all groups of an image: the groups are identical searches of
    // num_beams / num_beam_groups beams, and the result is that of ONE of them.  That one is what runs here.
    const int sub = num_beams / num_beam_groups;
    return run_generate(m, pixels, pixel_fmt, B, sub, max_len, length_penalty, out_ids, out_len, out_scores, nullptr,
                        (hipStream_t)stream, /*force_beam=*/true);
}

long long cap_g8_saturations(int reset) {
    if (hipDeviceSynchronize() != hipSuccess) { cap_set_error("cap_g8_saturations: device synchronisation failed"); return -1; }
    unsigned long long total = 0;
    if (cap_g8_clamped_gemm(&total, reset) != 0 || cap_g8_clamped_gemm_pp(&total, reset) != 0 || cap_g8_clamped_elementwise(&total, reset) != 0 ||
        cap_g8_clamped_attention(&total, reset) != 0 || cap_g8_clamped_decode_small(&total, reset) != 0)
        return -1;
    return (long long)total;
}

int cap_profile_enable(CapHandle h, int on) {
    Captioner* m = (Captioner*)h;
    if (!m) return -1;
    for (auto& r : m->prof_recs) { (void)hipEventDestroy(r.e0); (void)hipEventDestroy(r.e1); }
    m->prof_recs.clear();
    m->prof = on != 0;
    return 0;
}

int cap_profile_report(CapHandle h, char* buf, size_t buf_bytes) {
    Captioner* m = (Captioner*)h;
    if (!m || !buf) return -1;
    CAP_HIP_CHECK(hipDeviceSynchronize());
    struct Agg { long n = 0; double ms = 0, flops = 0, bytes = 0; };
    std::map<std::string, Agg> agg;
    for (auto& r : m->prof_recs) {
        float ms = 0.f;
        if (hipEventElapsedTime(&ms, r.e0, r.e1) != hipSuccess) continue;
        Agg& a = agg[r.tag];
        a.n++; a.ms += ms; a.flops += r.flops; a.bytes += r.bytes;
    }
    std::string out = "{";
    bool first = true;
    for (auto& kv : agg) {
        char line[512];
        snprintf(line, sizeof(line), "%s\"%s\": {\"launches\": %ld, \"ms\": %.6f, \"flops\": %.6e, \"bytes\": %.6e}",
                 first ? "" : ", ", kv.first.c_str(), kv.second.n, kv.second.ms, kv.second.flops, kv.second.bytes);
        out += line;
        first = false;
    }
    out += "}";
    if (out.size() + 1 > buf_bytes) { cap_set_error("cap_profile_report: buffer too small (%zu needed)", out.size() + 1); return -1; }
    memcpy(buf, out.c_str(), out.size() + 1);
    return 0;
}

// ---- single-kernel entry points.  dtype 2 (CAP_F32_SPLIT) = the split mode's convention: GEMM operands / kernel outputs
// that feed a GEMM are G8 (weights scaled by G8_WSCALE = 4096: cap_op_convert_weight), everything else is fp32.
static int dt_of(int dtype) { return dtype == CAP_BF16 ? CAP_DT_BF16 : dtype == CAP_F32_SPLIT ? CAP_DT_G8 : CAP_DT_F32; }
static int in_dt_of(int dtype) { return dtype == CAP_BF16 ? CAP_DT_BF16 : CAP_DT_F32; }
int cap_op_gemm(int dtype, const void* A, const void* W, const float* bias, const float* resid, void* C, int M, int N,
                int K, int gelu, int out_f32, int tile, void* stream) {
    GemmParams p;
    memset(&p, 0, sizeof(p));
    p.A = A; p.lda = K; p.W = W; p.ldw = K; p.C = C; p.ldc = N; p.bias = bias; p.resid = resid; p.ldr = N;
    p.M = M; p.N = N; p.K = K; p.gelu = gelu; p.out_f32 = out_f32; p.epi = EPI_STORE; p.splitk = 1;
#ifdef CAP_EXPERIMENTS
    if (tile == 9 || tile == 13 || tile == 14 || tile == 21) { p.aux = resid; p.resid = nullptr; }   // instrumented kernels: `resid` is the cycle-count buffer
#endif
    return launch_gemm(dt_of(dtype), p, tile, (hipStream_t)stream);
}
int cap_op_gemm_partial(int dtype, const void* A, const void* W, float* part, int M, int N, int K, int splitk, int tile,
                        void* stream) {
    GemmParams p;
    memset(&p, 0, sizeof(p));
    p.A = A; p.lda = K; p.W = W; p.ldw = K; p.C = part; p.ldc = N; p.M = M; p.N = N; p.K = K;
    p.out_f32 = 1; p.epi = EPI_PARTIAL; p.splitk = splitk;
    return launch_gemm(dt_of(dtype), p, tile, (hipStream_t)stream);
}
int cap_op_layernorm(int dtype, const float* in, const float* gamma, const float* beta, float eps, void* out_t,
                     float* out_f, int M, int D, void* stream) {
    return launch_layernorm(dt_of(dtype), in, D, gamma, beta, eps, out_t, out_f, M, D, (hipStream_t)stream);
}
int cap_op_vit_attention(int dtype, const void* qkv, void* ctx, int B, int N, int H, int impl, void* stream) {
    // dtype CAP_F32_SPLIT: impl 3 = G8 q|k|v in (what the split mode's qkv GEMM writes; the split-fp16 MFMA kernel), any other
    // impl = fp32 q|k|v in; the context is G8 either way
    if (dtype == CAP_F32_SPLIT && (impl == 3 || impl == 5))      // 5: the one-workgroup-per-unit kernel where 3 would pick the persistent one
        return launch_vit_attention(CAP_DT_G8, qkv, ctx, B, N, H, impl == 5 ? 5 : 0, (hipStream_t)stream, 64, 0, CAP_DT_G8);
    return launch_vit_attention(in_dt_of(dtype), qkv, ctx, B, N, H, impl, (hipStream_t)stream, 64, 0, dt_of(dtype));
}
int cap_op_vit_attention_hd(int dtype, const void* qkv, void* ctx, int B, int N, int H, int head_dim, int impl, void* stream) {
    // impl bit 8: causal mask (decoder prefill)
    return launch_vit_attention(in_dt_of(dtype), qkv, ctx, B, N, H, impl & 7, (hipStream_t)stream, head_dim, (impl >> 3) & 1,
                                dt_of(dtype));
}
int cap_crop_resize_tables(const int32_t* rects, const int32_t* geom, int n, int S, int KH, int KV, int32_t* hb, int32_t* hk,
                           int32_t* vb, int32_t* vk, void* stream) {
    return launch_crop_resize_tables(rects, geom, n, S, KH, KV, hb, hk, vb, vk, (hipStream_t)stream);
}
int cap_crop_resize_u8(const uint8_t* frame, int H, int W, int bgr, const int32_t* rects, const int32_t* hb, const int32_t* hk,
                       int KH, const int32_t* vb, const int32_t* vk, int KV, int n, int S, uint8_t* out, void* stream) {
    return launch_crop_resize_u8(frame, H, W, bgr, rects, hb, hk, KH, vb, vk, KV, n, S, out, (hipStream_t)stream);
}
int cap_crop_resize_u8_frames(const uint8_t* packed, const int64_t* frames, int bgr, const int32_t* rects, const int32_t* hb, const int32_t* hk,
                              int KH, const int32_t* vb, const int32_t* vk, int KV, int n, int S, uint8_t* out, void* stream) {
    if (!frames) { cap_set_error("crop_resize_frames: null frame table"); return -1; }
    return launch_crop_resize_u8(packed, 0, 0, bgr, rects, hb, hk, KH, vb, vk, KV, n, S, out, (hipStream_t)stream, (const long long*)frames);
}
int cap_op_reduce_layernorm(int dtype, const float* part, int S, const float* bias, const float* resid, const float* gamma,
                            const float* beta, float eps, void* out_t, float* out_f, float* y_out, int M, int D,
                            int per_row_block, void* stream) {
    return launch_reduce_layernorm(dt_of(dtype), part, S, bias, resid, gamma, beta, eps, out_t, out_f, y_out, M, D,
                                   (hipStream_t)stream, per_row_block != 0, false);
}
int cap_op_gemm_skinny(const void* A, const void* W, const float* bias, int act, void* out, float* part, int M, int N, int K,
                       void* stream) {
    return launch_gemm_skinny(A, K, W, K, bias, act, out, N, part, M, N, K, (hipStream_t)stream);
}
int cap_op_gemm_skinny_slices(int N, int K, int finished) { return skinny_plan(N, K, finished != 0); }
int cap_op_quant_i8_pack(const float* W, void* packed, float* scale, int rows, int cols, void* stream) {
    return launch_quant_i8_pack(W, packed, scale, rows, cols, (hipStream_t)stream);
}
int cap_op_gemm_skinny_i8(const void* A, const void* packed, const float* scale, const float* bias, int act, void* out, float* part, int M,
                          int N, int K, void* stream) {
    return launch_gemm_skinny_i8(A, K, packed, scale, bias, act, out, N, part, M, N, K, (hipStream_t)stream);
}
int cap_op_gemm_skinny_i8_slices(int N, int K, int finished) { return skinny_i8_plan(N, K, finished != 0); }
int cap_op_decode_attention(int dtype, const void* q, const void* kbase, const void* vbase, const int32_t* anc,
                            int anc_ld, int rows_per_kv, int kv_ld, int n_keys, void* out, int R, int H, int impl,
                            void* stream) {
    // impl bit 16: kbase / vbase are KV16 blocks (cap_op_pack_kv16) whose row 0 is the launch's first K/V row
    return launch_decode_attention(in_dt_of(dtype), q, kbase, vbase, anc, anc_ld, rows_per_kv, kv_ld, n_keys, out, R, H, impl & 15,
                                   (hipStream_t)stream, nullptr, 0, nullptr, 0, 0, 0, dt_of(dtype), nullptr, (impl >> 4) & 1, 0);
}
int cap_op_gemm_crosskv(int dtype, const void* A, const void* W, const float* bias, void* cache, int n_img, int tokens, int heads,
                        int layers, int K, int kv16, void* stream) {
    GemmParams p;
    memset(&p, 0, sizeof(p));
    p.A = A; p.lda = K; p.W = W; p.ldw = K; p.C = cache; p.ldc = 0; p.bias = bias;
    p.M = n_img * tokens; p.N = layers * 2 * heads * 64; p.K = K; p.out_f32 = 1; p.epi = EPI_CROSSKV; p.splitk = 1;
    p.p0 = tokens; p.p1 = heads; p.p2 = n_img; p.kv16 = kv16;
    return launch_gemm(dt_of(dtype), p, 0, (hipStream_t)stream);
}
int cap_op_pack_kv16(const float* src, void* dst, size_t n_rows, void* stream) {
    return launch_pack_kv16(src, dst, n_rows, (hipStream_t)stream);
}
int cap_op_beam_candidates(const float* logits, int ld, int V, int B, int K, int legacy_raw, int masked_id, float* out_val,
                           int32_t* out_idx, void* stream) {
    if (!logits || !out_val || !out_idx || B < 1 || K < 1 || V < 1 || ld < V) { cap_set_error("cap_op_beam_candidates: bad arguments"); return -1; }
    void* st = nullptr;
    CAP_HIP_CHECK(hipMalloc(&st, beam_state_bytes(B, K, 4)));
    int rc = beam_candidates_only(st, logits, ld, V, B, K, legacy_raw ? BEAM_LEGACY_RAW : BEAM_HF_V5, masked_id, out_val, out_idx,
                                  (hipStream_t)stream);
    if (hipStreamSynchronize((hipStream_t)stream) != hipSuccess) rc = -1;
    (void)hipFree(st);
    return rc;
}
int cap_op_convert(int dtype, const float* src, void* dst, size_t n, void* stream) {
    return launch_convert(dt_of(dtype), src, dst, n, (hipStream_t)stream);
}
int cap_op_convert_weight(int dtype, const float* src, void* dst, int rows, int cols, void* stream) {
    return launch_convert2d(dt_of(dtype), src, dst, rows, cols, cols, (hipStream_t)stream, dtype == CAP_F32_SPLIT ? G8_WSCALE : 1.0f);
}

}  // extern "C"
