/* C ABI of the MI355X-native captioner forward path (libcaptioner_hip.so).
 *
 * Nothing like this exists in the reference (it is 100 % Python): every entry point below replaces a
 * piece of Python/torch that the reference's captioner wrappers run on CPU, and is what a binding for this path
 * (ctypes here; pybind11/cffi equally) has to call.  Plain pointers and sizes only - no torch types.
 *
 *   reference interface replaced                                              entry point
 *   -------------------------------------------------------------------------------------------------------------
 *   model construction  captioner/models/blip2/blip2.py:17-22 (from_pretrained),    cap_create, cap_load_weight,
 *                       captioner/models/coca/factory.py:183-338 (create_model),     cap_finalize_weights
 *                       utils/predictor_utils.py:182-185 (load_state_dict)
 *   image tower         coca_model.py:152-155 `_encode_image`;                       cap_encode
 *                       HF modeling_blip.py:901-906 `vision_model(pixel_values)`
 *   generate            blip2.py:26 `model.generate(..., output_logits=True)`;       cap_generate
 *                       coca.py:29 `model.generate(x, generation_type=...)`;
 *                       coca_model.py:205-333 (greedy/top-k loop), :335-482 (beam);
 *                       blip2.py:26 BLIP-2 OPT (CAP_ARCH_BLIP2): out_ids = the max_len NEW tokens (HF's sequences
 *                       minus the 32 image placeholders and BOS), logits as HF's `output_logits`
 *   caption embedding   agents/goal_exploration/goal_exploration.py:57,102 and                cap_embed_text
 *                       detector/pseudolabeler.py:568,677 `SentenceTransformer("all-MiniLM-L6-v2").encode(caption)`  (CAP_ARCH_MINILM handle)
 *   one crop per call   coca.py:27-33, blip2.py:24-29, goal_exploration.py:95-105,     cap_generate (rows <= 16: fused
 *                       pseudolabeler.py:673-676 (the callers hand over ONE image)      launches), cap_set_decode_path
 *   greedy stopping     HF generation/utils.py:2894-2937 (a finished row keeps its slot and   cap_set_row_compaction (the open
 *                       is fed pad tokens; `unfinished_sequences.max() == 0` ends the loop)  rows only), cap_set_early_exit
 *   batches of crops    detector/pseudolabeler.py:664-711, scripts/run_pseudolabeler.py:77-107  cap_create_shared (n engines on one
 *                       (one generate per crop; here: micro-batches merged into passes)          weight store: engine.EnginePool)
 *   load options        blip2.py:19-22 `load_in_8bit=True, torch_dtype=float16`;       CapConfig.weight_int8 (int8 Linear weights as
 *                       evaluate_finetuned_model.py:147-148 `PeftModel.from_pretrained`  bitsandbytes stores them, quantised by
 *                                                                                      cap_load_weight), CapConfig.compute_dtype,
 *                                                                                      CapConfig.cross_kv_fp32 (host side:
 *                                                                                      weights.merge_peft_lora)
 *   device move/free    predictor_utils.py:187 `.to(...)`; object lifetime            cap_destroy
 *   errors              Python exceptions (utils_captioner.py:6, factory.py:231,309)  int return codes + cap_last_error
 *
 * Conventions: every function returns 0 on success, non-zero on failure (message via cap_last_error(), thread-local).
 * All device buffers are caller-owned; the library owns weights, KV caches and workspace, sized at cap_create from
 * max_batch / max_beams / max_len.  One handle per stream; a handle is not thread-safe.  Calls enqueue work on the
 * given hipStream_t (passed as void*) and return without synchronising, except where noted.
 */
#ifndef CAPTIONER_HIP_H
#define CAPTIONER_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct CapHandle_s* CapHandle;

enum { CAP_ARCH_BLIP = 0, CAP_ARCH_COCA = 1, CAP_ARCH_MINILM = 2, CAP_ARCH_BLIP2 = 3 };
/* Arithmetic of the GEMM / attention operands (accumulation is fp32 in every mode):
 *   CAP_F32        fp32 operands, exact fp32 products on the fp32 MFMA pipe (v_mfma_f32_32x32x2_f32)
 *   CAP_BF16       bf16 operands on the bf16 MFMA pipe - fastest, but not token-identical to an fp32 reference
 *   CAP_F32_SPLIT  fp32 values carried into every GEMM as two fp16 halves (hi + lo = x to 2^-23), each product formed as
 *                  hi.hi + hi.lo + lo.hi on the fp16 MFMA pipe: products good to ~2^-21 (fp32: 2^-24, bf16: 2^-9) at 3/16
 *                  of the fp32 pipe's cost; LayerNorm, softmax, attention, residual stream and the self-attention K/V cache are
 *                  fp32 as in CAP_F32.  The CROSS-attention K/V cache (the decode side's HBM stream) is KV16 - per token and
 *                  head 64 int16 and one fp32 scale, 15 value bits relative to the row's largest element - whenever an image has
 *                  more than 32 tokens (every real BLIP / CoCa geometry); CapConfig.cross_kv_fp32 = 1 keeps fp32 rows instead
 *                  (1.9x the bytes), cap_cross_cache_kind() reports what a handle uses.  Token-identical to the fp32 reference
 *                  on every golden fixture; the default of the plugin and of bench.py.  Every captioner architecture (BLIP, BLIP-2, CoCa); not the sentence encoder.  The mode has a
 *                  finite range - see cap_g8_saturations. */
enum { CAP_F32 = 0, CAP_BF16 = 1, CAP_F32_SPLIT = 2 };
enum { CAP_PIX_F32_NCHW = 0, CAP_PIX_U8_NHWC = 1 }; /* normalised fp32 [B,3,H,W] | raw RGB uint8 [B,H,W,3] */

typedef struct CapConfig {
    int32_t struct_size;          /* sizeof(CapConfig), for ABI checking */
    int32_t arch;                 /* CAP_ARCH_BLIP | CAP_ARCH_COCA | CAP_ARCH_MINILM (sentence encoder: only the t_* / vocab /
                                     max_pos / max_batch / max_len fields are read; max_len = tokens per sentence) */
    int32_t compute_dtype;        /* CAP_F32 | CAP_BF16 | CAP_F32_SPLIT (see the enum) */
    /* vision tower */
    int32_t image_size, patch_size, v_hidden, v_layers, v_heads, v_mlp;
    float v_eps;
    /* text decoder */
    int32_t t_hidden, t_layers, t_heads, t_ffn, vocab, max_pos;
    float t_eps;
    int32_t bos, eos, pad;
    /* capacity of the library-owned arena */
    int32_t max_batch, max_beams, max_len;
    /* raw-pixel normalisation for CAP_PIX_U8_NHWC: (x/255 - mean[c]) / std[c] */
    float pix_mean[3], pix_std[3];
    /* CAP_ARCH_COCA only (open_clip coca_ViT-L-14.json): attentional pooler output width / queries / heads, number of
     * multimodal decoder layers (t_layers = unimodal text layers), MinLength of the decode loop.  For CoCa
     * bos = start-of-text id, max_pos = context_length + 1, max_len = generate()'s seq_len. */
    int32_t embed_dim, pool_queries, pool_heads, mm_layers, min_len;
    /* CAP_ARCH_BLIP2 only (HF Blip2Config): Q-Former geometry, cross-attention on layers i % q_cross_freq == 0, number of
     * query tokens.  v_* = ViT-g (head_dim = v_hidden / v_heads, any multiple of 8 up to 128), t_* = OPT decoder (pre-LN,
     * ReLU, learned positions with offset 2; max_pos = rows of embed_positions - 2), max_len = new tokens per caption,
     * bos/eos/pad = OPT ids as generate() uses them. */
    int32_t q_hidden, q_layers, q_heads, q_ffn, q_cross_freq, num_query_tokens;
    float q_eps;
    /* CAP_F32_SPLIT only: 1 = the cross-attention K/V cache keeps fp32 rows (as CAP_F32 does) instead of KV16.  0 (default): KV16
     * for images of more than 32 tokens.  Ignored by the other modes (bf16 rows / fp32 rows). */
    int32_t cross_kv_fp32;
    /* CAP_ARCH_BLIP2 + CAP_BF16 only: 1 = the reference's own load mode for BLIP-2 (captioner/models/blip2/blip2.py:19-22,
     * `load_in_8bit=True`): the OPT decoder layers' Linear weights (q / k / v / out_proj / fc1 / fc2 - where the bytes of a decode
     * step are) are kept as bitsandbytes keeps a Linear8bitLt weight - one signed byte per element, q = rint(w * 127 /
     * absmax(row)), and the row's absmax / 127 in fp32 - quantised on the device by cap_load_weight from the fp32 tensor it is
     * given; the GEMMs stream the bytes and multiply the row sums by the scales.  Activations stay bf16 (bitsandbytes' int8
     * activation path with fp16 outlier columns is not restated: weights-only, "W8A16").  lm_head stays bf16 (HF does not
     * convert it either).  Needs OPT widths the int8 weight stream takes (opt-2.7b's 2560 / 10240 are). */
    int32_t weight_int8;
} CapConfig;

const char* cap_last_error(void);
int cap_version(void);

int cap_create(const CapConfig* cfg, CapHandle* out);
/* A further handle on the SAME weights (read-only once loaded): own arena / KV caches / workspace, sized by cfg's max_batch,
 * max_beams, max_len; everything else in cfg must equal the configuration of `weights_of` (same model, compute dtype, GPU).
 * What a pool of engines on several streams uses: n handles cost one copy of the weights + n arenas.  Tensors loaded through
 * any of the handles are seen by all.  The weights are freed when the last handle referencing them is destroyed (any order).
 * Replaces: one `model.to(device)` copy per worker (reference utils/predictor_utils.py:187). */
int cap_create_shared(const CapConfig* cfg, CapHandle weights_of, CapHandle* out);
int cap_destroy(CapHandle h);

/* Stream one fp32 tensor of the checkpoint into the library.  Names: HuggingFace BLIP state-dict keys for CAP_ARCH_BLIP;
 * open_clip CoCa keys (`visual.*`, `text.*`, `text_decoder.*`) for CAP_ARCH_COCA plus a few tensors the host derives
 * once at load (embodied_captioning_amd/coca_weights.py): `derived.pool_q` (ln_q(query) projected), `derived.pool_kv.*`
 * (k|v projection of the pooler fused), `derived.cross_q.{i}.*`, `derived.cross_kv.*` (all layers' cross k|v projections
 * with ln_1_kv folded in), `derived.vocab.weight` (text_projection transposed).  `data` is a
 * host pointer (on_device = 0) or a device pointer (on_device = 1); the library converts to its compute layout
 * (bf16 cast, q/k/v and cross-K/V concatenation) on `stream` and synchronises before returning.
 * Unknown names return 1 (not an error for tied/duplicate heads, see cap_finalize_weights). */
int cap_load_weight(CapHandle h, const char* name, const float* data, int on_device, int ndim, const int64_t* shape,
                    void* stream);
/* Returns 0 when every tensor the architecture needs has been loaded; otherwise the count of missing tensors
 * (names in cap_last_error()). */
int cap_finalize_weights(CapHandle h);

/* Early exit of cap_generate's decode loop, as HF generate stops once every caption is finished (the remaining steps
 * would only write pad, so the outputs are the same either way).  poll_steps > 0: after every poll_steps-th step one
 * tiny kernel reports the number of open captions through a host-mapped word and the stream is synchronised.
 * 0 (default): never look - no host synchronisation inside cap_generate, which can then be captured in a graph.  With early
 * exit, rows of out_step_logits beyond the last executed step are left untouched.  Reference: HF
 * GenerationMixin._sample / _beam_search stopping criteria behind captioner/models/blip/blip.py:30 `model.generate`. */
int cap_set_early_exit(CapHandle h, int poll_steps);
/* Decode steps the last cap_generate on this handle ran (max_len - 1 without early exit; diagnostics and tests). */
int cap_last_decode_steps(CapHandle h);

/* Which kernels the decode steps of cap_generate run on (CAP_ARCH_BLIP, split and bf16 modes).  The reference calls its captioner
 * with ONE crop per call (captioner/models/coca/coca.py:27-33, blip2/blip2.py:24-29, agents/goal_exploration/goal_exploration.py:
 * 95-105; BASELINE config 1: 8 crops): for images x beams <= 16 rows a decoder layer-step runs as 6 fused launches
 * (csrc/decode_small.hip) instead of the batch path's 11.  Both paths form the same sums in the same order: tokens, logits and
 * scores have the same bits (tests/test_small_decode_gpu.py).
 *   path 0 (default): by row count;  1: always the batch kernels;  2: always the small-batch kernels - cap_generate then fails for
 *   calls they do not take (more than 16 rows, more than 32 positions [checked at entry], CAP_F32, other architectures). */
int cap_set_decode_path(CapHandle h, int path);
/* 1 = batch kernels, 2 = small-batch kernels: what the last decode step of the last cap_generate ran on (0 before any). */
int cap_last_decode_path(CapHandle h);
/* Row compaction of the greedy decode loop (CAP_ARCH_BLIP, batch kernels, more than 16 rows, max_len <= 33, no per-step logits):
 * after every token selection the rows of the captions still open are packed to the front and every kernel of the next step
 * works on those rows only (HF's greedy loop, generation/utils.py:2894-2937, keeps feeding pad tokens to a finished row; its
 * outputs are never read).  A caption's arithmetic does not depend on the row it sits in, so tokens and lengths are the bits of the
 * uncompacted loop (tests/test_merged_passes_gpu.py).  on = 1 (default) / 0. */
int cap_set_row_compaction(CapHandle h, int on);
/* 1 if the decode loop of the last cap_generate ran compacted, 0 if not (-1: null handle). */
int cap_last_row_compaction(CapHandle h);
/* Layout of the handle's cross-attention K/V cache: 0 = fp32 rows, 1 = bf16 rows, 2 = KV16 (int16 + one fp32 scale per 64-wide
 * head row; CAP_F32_SPLIT unless CapConfig.cross_kv_fp32).  -1 for a null handle. */
int cap_cross_cache_kind(CapHandle h);

/* Object crops of one frame, resized for the captioner ON THE DEVICE, bit-exact with Pillow's
 * `Image.crop(box).resize((S, S), Image.BICUBIC)` - what the reference does to every detected box on the host before the
 * captioner sees it (detector/pseudolabeler.py:670-675 expand + crop, BGR->RGB at :670; HF BlipImageProcessor.resize).
 *   frame  uint8 [H, W, 3] (device), bgr != 0: channels are swapped to RGB on the way
 *   rects  int32 [n, 4] = x1, y1, x2, y2 of each (already expanded) box; parts outside the frame read as zeros, as
 *          Image.crop pads them
 *   hb, vb int32 [n, S, 2] = first input index (relative to the crop) and tap count of every output column / row
 *   hk, vk int32 [n, S, KH] / [n, S, KV] = Pillow's 22-bit integer coefficients (normalize_coeffs_8bpc), zero padded
 *   out    uint8 [n, S, S, 3] RGB -> feed to cap_generate / cap_encode as CAP_PIX_U8_NHWC
 * The tables are O(S) doubles per box: cap_crop_resize_tables fills them on the device (fp64 without contraction - equal
 * to Pillow's bit for bit), or the host builds them (embodied_captioning_amd/preprocess.py::pil_bicubic_coeffs).  All
 * pointers are device pointers. */
/* geom int32 [n, 4] = (resized width, resized height, left, top): out = the window [left, left+S) x [top, top+S) of the crop
 * resized to (width, height); (S, S, 0, 0) for the plain square resize.  KH / KV >= 2 ceil(2 max(scale, 1)) + 1 of the
 * widest / tallest box (scale = crop size / resized size). */
int cap_crop_resize_tables(const int32_t* rects, const int32_t* geom, int n, int S, int KH, int KV, int32_t* hb, int32_t* hk,
                           int32_t* vb, int32_t* vk, void* stream);
int cap_crop_resize_u8(const uint8_t* frame, int H, int W, int bgr, const int32_t* rects, const int32_t* hb,
                       const int32_t* hk, int KH, const int32_t* vb, const int32_t* vk, int KV, int n, int S, uint8_t* out,
                       void* stream);
/* The same for a LIST of images (what generate_batch / caption_batch receive: PIL crops of different sizes): `packed` holds the n
 * images' bytes back to back, frames int64 [n, 3] = (byte offset, height, width) of image b, rects[b] its rectangle inside it
 * ((0, 0, W, H) for the whole image) - one upload and one launch for the whole list. */
int cap_crop_resize_u8_frames(const uint8_t* packed, const int64_t* frames, int bgr, const int32_t* rects, const int32_t* hb,
                              const int32_t* hk, int KH, const int32_t* vb, const int32_t* vk, int KV, int n, int S, uint8_t* out,
                              void* stream);

/* Image tower.  pixels: B frames in `pixel_fmt`; out_embeds: fp32 [B, tokens, v_hidden] (device). */
int cap_encode(CapHandle h, const void* pixels, int pixel_fmt, int B, float* out_embeds, void* stream);

/* Encoder + autoregressive decode.
 *   num_beams == 1: greedy (HF `_sample` with do_sample=False); num_beams > 1: HF v5 beam search.
 *   out_ids     int32 [B, max_len]   token ids incl. BOS; rows are padded after their end
 *   out_len     int32 [B]            tokens in each row incl. BOS and EOS (may be NULL)
 *   out_scores  fp32  [B]            beam `sequences_scores`; untouched for greedy (may be NULL)
 *   out_step_logits fp32 [max_len-1, B*num_beams, vocab]  raw per-step logits (may be NULL).  Greedy: a caption's rows
 *               are meaningful up to and including the step that produced its EOS; later steps of that row are
 *               unspecified (the attention kernels skip ended captions; HF feeds them pad and ignores the result)
 * max_len <= cfg.max_len, B <= cfg.max_batch, num_beams <= cfg.max_beams. */
int cap_generate(CapHandle h, const void* pixels, int pixel_fmt, int B, int num_beams, int max_len,
                 float length_penalty, int32_t* out_ids, int32_t* out_len, float* out_scores,
                 float* out_step_logits, void* stream);

/* CoCa's `_generate_beamsearch` with beam GROUPS (reference coca_model.py:335-482; `generate()` defaults num_beams = 6,
 * num_beam_groups = 3, :218-219): num_beams % num_beam_groups == 0, each group a beam search of num_beams / num_beam_groups
 * beams, the best hypothesis over an image's groups returned.  The reference attaches no diversity processor (:236-241), so
 * its groups are identical searches and the result equals ONE search of num_beams / num_beam_groups beams - which is what this
 * entry point runs (a group of one beam runs as a 1-beam BEAM search, not as cap_generate's greedy loop).  Outputs as
 * cap_generate; B <= max_batch, num_beams <= max_beams.  CAP_ARCH_COCA only. */
int cap_generate_groups(CapHandle h, const void* pixels, int pixel_fmt, int B, int num_beams, int num_beam_groups, int max_len,
                        float length_penalty, int32_t* out_ids, int32_t* out_len, float* out_scores, void* stream);

/* Sentence encoder (CAP_ARCH_MINILM handle; weights by HF BertModel names as sentence-transformers stores them):
 * WordPiece ids int32 [B, L] incl. [CLS]/[SEP] (rows padded with any valid id), lens int32 [B] = valid tokens per row
 * -> out fp32 [B, t_hidden]: mean of the last hidden states over the valid tokens, L2-normalised
 * (sentence-transformers Pooling(mean) + Normalize).  All pointers are device pointers. */
int cap_embed_text(CapHandle h, const int32_t* ids, const int32_t* lens, int B, int L, float* out, void* stream);

/* Range guard of CAP_F32_SPLIT.  Weights: cap_load_weight refuses (returns -1, message names the tensor) a tensor bound for
 * a GEMM-operand slot whose max |w| exceeds 65000 / 4096 = 15.87 or that holds a NaN - nothing is clipped silently.
 * Activations: every kernel that writes a GEMM operand clamps to +-65000 (fp16's range) and COUNTS what it clamped; this
 * returns the count on the current device since the last reset (groups of four adjacent elements count once), -1 on error.
 * It synchronises the device.  0 means every value of every generate / encode since the reset was inside the envelope in
 * which the mode is fp32-grade; anything else means "run this checkpoint / input in CAP_F32".
 * Reference behaviour replaced: none (fp32 torch on the CPU has no such range) - this is the honesty clause of the fast mode. */
long long cap_g8_saturations(int reset);

/* Bytes of device memory this handle allocated: its arena, plus the weights if it is the handle that created them
 * (cap_create); a cap_create_shared handle reports its arena only. */
size_t cap_device_bytes(CapHandle h);

/* Per-kernel timing with HIP events on the launch stream (bench.py's roofline leg).  While enabled every launch of the
 * tagged kernels is bracketed by an event pair; cap_profile_report synchronises the stream and writes a JSON object
 * {"tag": {"launches": n, "ms": total, "flops": f, "bytes": b}, ...} into buf. */
int cap_profile_enable(CapHandle h, int on);
int cap_profile_report(CapHandle h, char* buf, size_t buf_bytes);

/* ---- single-kernel entry points (used by tests/ to check each kernel against a plain reference) ----
 * dtype CAP_F32_SPLIT follows the mode's convention: GEMM operands (A, W, and a non-fp32 C) and the outputs of kernels that
 * feed a GEMM (layernorm out_t, attention ctx / out) are G8 = 4 bytes per element, every 8 consecutive elements of a row
 * stored as [8 fp16 hi | 8 fp16 lo]; q|k|v, caches and partial sums are fp32. */
int cap_op_gemm(int dtype, const void* A, const void* W, const float* bias, const float* resid, void* C, int M, int N,
                int K, int gelu, int out_f32, int tile, void* stream);
/* split-K form of the decode GEMMs: part[z][M][N] fp32 = A[:, z-th K slice] . W[:, z-th K slice]^T, no bias (the consumer -
 * cap_op_reduce_layernorm, or the decode attention kernels - sums the slices in order).  K % (slab * splitk) == 0 with slab =
 * 32 (fp32, split) / 64 (bf16); tile as cap_op_gemm (2 = the 64x64 decode tile). */
int cap_op_gemm_partial(int dtype, const void* A, const void* W, float* part, int M, int N, int K, int splitk, int tile,
                        void* stream);
int cap_op_layernorm(int dtype, const float* in, const float* gamma, const float* beta, float eps, void* out_t,
                     float* out_f, int M, int D, void* stream);
int cap_op_vit_attention(int dtype, const void* qkv, void* ctx, int B, int N, int H, int impl, void* stream);
/* heads of any width (BLIP-2's ViT-g/14: 88); impl 1 forces the scalar kernel, 0 picks the MFMA kernel where one exists;
 * impl | 8: causal mask (query i sees keys 0..i: the OPT prefill) */
int cap_op_vit_attention_hd(int dtype, const void* qkv, void* ctx, int B, int N, int H, int head_dim, int impl,
                            void* stream);
/* split-K consumer: y = sum_z part[z][M][D] + bias + resid (-> y_out, may alias resid), LayerNorm(y) -> out_t (dtype) /
 * out_f (fp32); any output may be NULL.  per_row_block: the decoder's workgroup-per-row kernels (few rows). */
int cap_op_reduce_layernorm(int dtype, const float* part, int S, const float* bias, const float* resid,
                            const float* gamma, const float* beta, float eps, void* out_t, float* out_f, float* y_out,
                            int M, int D, int per_row_block, void* stream);
/* bf16 weight-streaming GEMM for a handful of rows (decode step of a large LM): out (bf16) = act(A W^T + bias) when part
 * is NULL, else part[z][M][N] fp32 slice sums.  Returns the slice count used (>= 1) or -1; cap_op_gemm_skinny_slices
 * tells it beforehand (0 = the shape does not fit the kernel). */
int cap_op_gemm_skinny(const void* A, const void* W, const float* bias, int act, void* out, float* part, int M, int N,
                       int K, void* stream);
int cap_op_gemm_skinny_slices(int N, int K, int finished);
/* The int8 form (CapConfig.weight_int8).  cap_op_quant_i8_pack: fp32 W [rows, cols] (rows % 16 == 0, cols % 64 == 0) ->
 * packed (rows * cols bytes, MFMA fragment order: block (t, s) = rows 16 t.., k = 64 s.. is the KiB at (t * cols / 64 + s) * 1024,
 * lane r + 16 g owns bytes 16 l..: k = 64 s + 8 g.. + 7, then k = 64 s + 32 + 8 g.. + 7) and scale [rows] = absmax(row) / 127.
 * cap_op_gemm_skinny_i8: as cap_op_gemm_skinny with (packed, scale) for W; N % 32 == 0; cap_op_gemm_skinny_i8_slices = its plan. */
int cap_op_quant_i8_pack(const float* W, void* packed, float* scale, int rows, int cols, void* stream);
int cap_op_gemm_skinny_i8(const void* A, const void* packed, const float* scale, const float* bias, int act, void* out, float* part,
                          int M, int N, int K, void* stream);
int cap_op_gemm_skinny_i8_slices(int N, int K, int finished);
int cap_op_decode_attention(int dtype, const void* q, const void* kbase, const void* vbase, const int32_t* anc,
                            int anc_ld, int rows_per_kv, int kv_ld, int n_keys, void* out, int R, int H, int impl,
                            void* stream);        /* impl | 16: kbase / vbase are KV16 blocks (no ancestry, > 32 keys) */
/* The split mode's cross-attention K/V cache layout: fp32 rows [n_rows, 64] -> one KV16 block: per row 64 int16 and one fp32 scale
   (x ~ q * scale, scale = max|x| / 32767 over the row), rows in groups of 32 = [32 x 128 bytes][32 scales] = 4224 bytes; dst holds
   (n_rows + 31) / 32 groups.  The cross-K/V GEMM's epilogue writes this layout; the op exists for the kernel tests. */
int cap_op_pack_kv16(const float* src, void* dst, size_t n_rows, void* stream);
/* The cross-K/V GEMM as the image side runs it (replaces the per-layer key / value Linear calls of HF:modeling_blip_text.py:161-175):
   A [n_img * tokens, K] x W [layers * 2 * heads * 64, K]^T + bias -> cache [layer][k | v][image][head][token][64]; kv16 = 1 (CAP_F32_SPLIT
   only): every (layer, k | v) block is a KV16 block of (n_img * heads * tokens + 31) / 32 groups, else fp32 / bf16 rows. */
int cap_op_gemm_crosskv(int dtype, const void* A, const void* W, const float* bias, void* cache, int n_img, int tokens, int heads,
                        int layers, int K, int kv16, void* stream);
/* The candidate selection of a beam step alone (first step: running score 0 for beam 0 of an item, -1e9 for the others):
 * logits fp32 [B*K][ld] -> the 2K best (score, token) per row, best first, ties to the lower token id; legacy_raw: scores are
 * raw logits + running score (CoCa), else log-softmax + running score (HF v5); masked_id >= 0: that token scores -inf (legacy
 * MinLength).  Synchronises the stream. */
int cap_op_beam_candidates(const float* logits, int ld, int V, int B, int K, int legacy_raw, int masked_id, float* out_val,
                           int32_t* out_idx, void* stream);
int cap_op_convert(int dtype, const float* src, void* dst, size_t n, void* stream);
/* Weight upload as cap_load_weight does it: dst [rows, cols] in the GEMM-operand type of `dtype` (CAP_F32_SPLIT: G8 halves of
 * 4096 * w - the split GEMM's epilogue divides by 4096; a G8 buffer is 4 bytes per element, cols % 8 == 0). */
int cap_op_convert_weight(int dtype, const float* src, void* dst, int rows, int cols, void* stream);

#ifdef __cplusplus
}
#endif
#endif
