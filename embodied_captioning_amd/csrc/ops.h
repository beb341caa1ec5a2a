// Launchers of the non-GEMM kernels (elementwise.hip, attention.hip, beam.hip).  All launch on `stream`,
// never allocate or synchronise (graph-capturable), and return 0 / -1 (+ cap_set_error).
#pragma once
#include "common.h"

// Row compaction of the greedy decode loop (captioner.hip, run_decoder_step): the rows of captions that are still open, in
// row order, and their count - both on the device, rebuilt after every token selection.  A kernel given a RowMap works on
// COMPACT rows c < *n (activations, split-K slabs, logits) and reaches what a caption owns across steps - its tokens, its
// self-attention cache rows, its image's cross K/V - through live[c]; compact rows from *n on return at once.  Null pointers =
// no map (every row, c = row).  A caption's arithmetic does not depend on the row it sits in (batch invariance), so the
// compacted loop produces the bits of the uncompacted one.
struct RowMap {
    const int* live = nullptr;     // int32 [R]: live[c] = row of the batch
    const int* n = nullptr;        // int32: open rows
};

// ---- elementwise.hip -------------------------------------------------------------------------
// fp32 -> T copy (weight upload / activation cast); dst rows may be padded: dst[r*dst_ld + c] = src[r*cols + c]
int launch_convert(int dtype, const float* src, void* dst, size_t n, hipStream_t s);
int launch_convert2d(int dtype, const float* src, void* dst, int rows, int cols, int dst_ld, hipStream_t s, float scale = 1.0f);
// dst[cols][rows] = src[rows][cols]^T
int launch_convert2d_t(int dtype, const float* src, void* dst, int rows, int cols, hipStream_t s, float scale = 1.0f);
// im2col-free patch gather: pixels -> A[B*P, Kpad] (T).  fmt 0: fp32 NCHW normalised; fmt 1: u8 NHWC raw RGB,
// normalised on the fly with (x/255 - mean[c]) / std[c].
int launch_patchify(int dtype, const void* pixels, int fmt, int B, int img, int ps, int Kpad, void* out,
                    const float* mean, const float* stdv, hipStream_t s);
// X[b,0,:] = cls + pos[0]
int launch_cls_rows(const float* cls, const float* pos, float* X, int B, int tokens, int D, hipStream_t s);
// row LayerNorm: in fp32 [M,D]; writes out_t (T, optional) and out_f (fp32, optional; may alias in)
int launch_layernorm(int dtype, const float* in, int ld_in, const float* gamma, const float* beta, float eps,
                     void* out_t, float* out_f, int M, int D, hipStream_t s);
// split-K / residual consumer: y = sum_z part[z][M][D] + bias + resid (fixed order); y_out (optional, may alias resid)
// receives y, then LayerNorm(y) goes to out_t / out_f as above (out_f may alias resid when y_out is null)
int launch_reduce_layernorm(int dtype, const void* part, int S, const float* bias, const float* resid,
                            const float* gamma, const float* beta, float eps, void* out_t, float* out_f, float* y_out,
                            int M, int D, hipStream_t s, bool per_row_block = false, bool part_in_t = false,
                            const int* n_rows = nullptr);   // device int32: rows from *n_rows on are left alone (RowMap::n)
// decoder embeddings: x = LN(word[tok] + pos[t]); tok = seq[row*seq_ld + t]
int launch_reduce_bias_act(int dtype, const float* part, int S, const float* bias, void* out, int M, int N, int act, hipStream_t s);
int launch_embed_tokens(int dtype, const int* ids, int L, const float* word, const float* pos, const float* type0,
                        const float* gamma, const float* beta, float eps, void* out_t, float* out_f, int R, int D,
                        hipStream_t s, int V);   // ids outside [0, V) are clamped
int launch_mean_pool_normalize(const float* x, const int* lens, int B, int L, int D, float* out, hipStream_t s);
int launch_text_attention(int dtype, const void* qkv, const int* lens, void* ctx, int B, int L, int H, int head_dim,
                          hipStream_t s);
int launch_embed(int dtype, const int* seq, int seq_ld, int t, const float* word, const float* pos,
                 const float* gamma, const float* beta, float eps, void* out_t, float* out_f, int R, int D,
                 hipStream_t s, float* y_out = nullptr,    // y_out: the un-normalised sum (pre-LN residual stream)
                 RowMap map = RowMap());                    // compact output row c reads the token of row map.live[c]
// greedy selection: argmax (lowest index wins ties), pad after EOS, append at seq[row][t+1], track finished/len
int launch_greedy_select(const float* logits, int ld, int V, int* seq, int seq_ld, int t, int max_len, int eos,
                         int pad, int* finished, int* out_len, int R, hipStream_t s, int min_len = 0, int force_eos = 0,
                         RowMap map = RowMap());            // logits row c belongs to row map.live[c] (seq / finished / out_len)
// live[] = the rows with finished[r] == 0 in ascending order, *n_live = their count (one workgroup; stable)
int launch_compact_rows(const int* finished, int R, int* live, int* n_live, hipStream_t s);
int launch_fill_i32(int* p, int v, size_t n, hipStream_t s);
int launch_fill_f32(float* p, float v, size_t n, hipStream_t s);
int launch_copy_f32(const float* src, float* dst, size_t n, hipStream_t s);

// max |src[i]| as the bit pattern of a non-negative float, atomically max-ed into *out_bits (caller zeroes it)
int launch_absmax_f32(const float* src, size_t n, unsigned int* out_bits, hipStream_t s);
// per-file readers of the G8 clamp counter (common.h)
int cap_g8_clamped_gemm(unsigned long long* total, int reset);
int cap_g8_clamped_gemm_pp(unsigned long long* total, int reset);
int cap_g8_clamped_elementwise(unsigned long long* total, int reset);
int cap_g8_clamped_attention(unsigned long long* total, int reset);

// ---- attention.hip ---------------------------------------------------------------------------
// ViT self-attention over a fused qkv buffer [B*N, 3*H*64] (T) -> ctx [B*N, H*64] (T); scale = 1/8.
// impl 0 = auto (MFMA for bf16 when N <= 256, scalar otherwise), 1 = scalar, 2 = MFMA.
int launch_vit_attention(int dtype, const void* qkv, void* ctx, int B, int N, int H, int impl, hipStream_t s, int head_dim = 64,
                         int causal = 0,    // causal: query i sees keys 0..i (decoder prefill over fused q|k|v rows)
                         int out_dtype = -1);   // type of ctx when it differs from qkv's (split mode: fp32 in, G8 out)
// split mode: can the q|k|v buffer be G8 (launch_vit_attention(CAP_DT_G8, ...): the split-fp16 MFMA kernel) for N tokens?
// (Yes for every N since the kernel walks the keys in chunks; kept as the switch to the fp32 kernels: fp32 in, G8 out.)
bool vit_attention_takes_g8(int N);
int launch_generic_attention(int dtype, const void* q, long ldq, long qbs, const void* k, long ldk, long kbs, const void* v,
                             long ldv, long vbs, void* out, long ldo, long obs, int B, int Lq, int Lk, int H, int hd,
                             int causal_off, hipStream_t s, int out_dtype = -1);   // out_dtype: see launch_vit_attention
int launch_opt_prefill_inputs(const float* proj, const float* tok, const float* pos, float* x, int B, int nq, int T, int bos,
                              hipStream_t s);
int launch_opt_token_inputs(const int* seq, int seq_ld, int cur, const float* tok, const float* pos, float* x, int B, int T,
                            hipStream_t s);
// decode step of a pre-LN decoder: q|k|v row [B, 3T] -> appends k, v to caches [B][Lmax][T] at `past`, out [B, T]
int launch_opt_decode_attention(int dtype, const void* qkv, void* kc, void* vc, void* out, int B, int T, int H, int Lmax,
                                int past, hipStream_t s, int out_dtype = -1);
int launch_kv_append(int dtype, const void* qkv, void* kc, void* vc, int B, int L, int T, int Lmax, int pos0, hipStream_t s);
int launch_rows_broadcast(int dtype, const float* src, float* dst_f, void* dst_t, int B, int n, int D, hipStream_t s);
// single-query decode attention. q [R, H*64] (T).  K/V of row r, head h, position j at
//   kbase + (((size_t)src(r,j) * H + h) * kv_ld + j) * 64   where src(r,j) = anc ? anc[r*anc_ld + j] : r / rows_per_kv
// n_keys positions; out [R, H*64] (T).  impl 0 = fast kernels (wave-per-head, chunked online softmax),
// impl 1 = simple two-pass block kernel (independent implementation kept for cross-checks).
// Fused producer (optional, impl 0 only): q_part != nullptr -> q = sum_z q_part[z][R][q_ld][q_col0 + ...] + q_bias;
// append_kv -> the new position's k/v are finished the same way (columns q_col0 + H*64, + 2*H*64), written to the
// cache of the row itself and attended (self-attention with n_keys <= 32).
int launch_decode_attention(int dtype, const void* q, const void* kbase, const void* vbase, const int* anc,
                            int anc_ld, int rows_per_kv, int kv_ld, int n_keys, void* out, int R, int H, int impl,
                            hipStream_t s, const float* q_part = nullptr, int q_S = 0, const float* q_bias = nullptr,
                            int q_ld = 0, int q_col0 = 0, int append_kv = 0, int out_dtype = -1,   // out_dtype: see launch_vit_attention
                            const int* skip_rows = nullptr,    // int32 [R] or null: rows with a non-zero flag are left untouched
                            int kv16 = 0,                      // 1: kbase / vbase are the bases of KV16 blocks (common.h; fp32 q, no
                            size_t kv_row0 = 0,                // ancestry, > 32 keys); kv_row0 = index of the launch's first row in them
                            RowMap map = RowMap());            // q_part / out rows are compact, caches and ancestry belong to map.live[c]
                                                               // (wave / online kernels, impl 0)
// fp32 rows [n_rows, 64] -> one KV16 block of kv16_block_bytes(n_rows) bytes: what the cross-K/V GEMM's epilogue writes, as a
// kernel of its own (tests)
int launch_pack_kv16(const float* src, void* dst, size_t n_rows, hipStream_t s);

// attentional pooler (CoCa): fixed projected queries qp fp32 [Q, E] shared by every image; kv (T) [B*N, 2E] with K in
// columns [0,E) and V in [E,2E); heads of E/heads dims (64 or 96); out (T) [B*Q, E].  scale = 1/sqrt(head_dim).
int launch_pool_attention(int dtype, const float* qp, const void* kv, void* out, int B, int N, int Q, int E, int heads,
                          hipStream_t s, int out_dtype = -1);

// ---- gemm_skinny.hip ---------------------------------------------------------------------------
// Weight-streaming bf16 GEMM for a handful of rows (decode step of a large LM).  skinny_plan: K slices for (N, K) in the
// finished (bias + act -> bf16, one slice) or partial form, 0 if the shape does not fit.  launch: part == nullptr ->
// out (bf16) = act(A W^T + bias), act 0 none / 1 GELU / 2 ReLU; part != nullptr -> part[z][M][N] fp32 slice sums for
// launch_reduce_layernorm / launch_reduce_bias_act.  Returns the slice count, or -1.
int skinny_plan(int N, int K, bool finished, int* nw_out = nullptr, int* tr_out = nullptr, int M = 32);
int launch_gemm_skinny(const void* A, int lda, const void* W, int ldw, const float* bias, int act, void* out, int ldc,
                       float* part, int M, int N, int K, hipStream_t s);

// int8 weights (BLIP-2 load_in_8bit): W as row-quantised signed bytes in MFMA fragment order + fp32 row scales (layout and
// arithmetic: gemm_skinny.hip).  launch_quant_i8_pack: fp32 [rows, cols] (rows % 16 == 0, cols % 64 == 0) -> dst (rows * cols
// bytes) and scale [rows].  skinny_i8_plan / launch_gemm_skinny_i8: as the bf16 pair, N % 32 == 0.
int skinny_i8_plan(int N, int K, bool finished, int* nw_out = nullptr);
int launch_quant_i8_pack(const float* src, void* dst, float* scale, int rows, int cols, hipStream_t s);
int launch_gemm_skinny_i8(const void* A, int lda, const void* Wp, const float* wscale, const float* bias, int act, void* out, int ldc,
                          float* part, int M, int N, int K, hipStream_t s);
// the bytes as a row-major bf16 matrix of the integers (for the tiled GEMM) and part[m][n] *= scale[n] on its fp32 output
int launch_dequant_i8_rowmajor(const void* packed, void* dst_bf16, int rows, int cols, hipStream_t s);
int launch_scale_cols(float* part, const float* scale, int M, int N, hipStream_t s);

// ---- preprocess.hip ----------------------------------------------------------------------------
// n boxes of one uint8 HWC frame -> out uint8 [n, S, S, 3] RGB, bit-exact with Pillow's crop + BICUBIC resize.
// rects int32 [n, 4] (x1, y1, x2, y2, inside the frame); hb/vb int32 [n, S, 2] (first tap, tap count) and hk/vk int32
// [n, S, KH|KV] integer coefficients built by the host (embodied_captioning_amd/preprocess.py).  All device pointers.
// the tables themselves, on the device (fp64, no contraction: equal to the host's bit for bit); geom int32 [n, 4] =
// (resized width, resized height, left, top) of the kept S x S window
int launch_crop_resize_tables(const int* rects, const int* geom, int n, int S, int KH, int KV, int* hb, int* hk, int* vb, int* vk,
                              hipStream_t s);
int launch_crop_resize_u8(const uint8_t* frame, int H, int W, int bgr, const int* rects, const int* hb, const int* hk, int KH,
                          const int* vb, const int* vk, int KV, int n, int S, uint8_t* out, hipStream_t s,
                          const long long* frames = nullptr);     // frames int64 [n, 3] = (byte offset in `frame`, H, W) of box b's own frame

// ---- beam.hip --------------------------------------------------------------------------------
size_t beam_state_bytes(int B, int K, int max_len);
int beam_candidates_only(void* state, const float* logits, int ld, int V, int B, int K, int mode, int eos_mask, float* out_val,
                         int* out_idx, hipStream_t s);     // test hook: candidate selection of the first step alone
// mode: BEAM_HF_V5 (log-softmax scores, HF 5.x `_beam_search`) or BEAM_LEGACY_RAW (the reference's CoCa loop: raw-logit
// scores, HF's pre-5.x BeamSearchScorer with one group, MinLength(min_len) on EOS) - see beam.hip
enum { BEAM_HF_V5 = 0, BEAM_LEGACY_RAW = 1 };
int launch_beam_init(void* state, int B, int K, int max_len, int bos, int pad, int eos, hipStream_t s, int mode = BEAM_HF_V5);
int launch_beam_step(void* state, const float* logits, int ld, int V, int B, int K, int max_len, int cur_len,
                     int eos, float length_penalty, int* anc, int anc_ld, hipStream_t s, int mode = BEAM_HF_V5, int min_len = 0);
int launch_beam_finalize(void* state, int B, int K, int max_len, int* out_ids, int* out_len, float* out_scores,
                         hipStream_t s);
// device pointer: int32, 1 while the beam loop of this state is still running (HF `is_done.all()` not yet true)
const int* beam_active_flag_p(void* state, int B, int K, int max_len);
// device pointer: int32 [B*K, max_len] running sequences of the given parity (= cur_len & 1)
const int* beam_running_tokens_p(void* state, int B, int K, int max_len, int parity);
