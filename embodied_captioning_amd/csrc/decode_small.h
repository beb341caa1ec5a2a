// Launchers of the small-batch decode kernels (decode_small.hip): rows = images x beams <= SMALL_MAX_ROWS.
#pragma once
#include "common.h"

constexpr int SMALL_MAX_ROWS = 16;

enum { SMALL_PRO_GLOBAL = 0,      // A [R, K] in the operand type from global memory
       SMALL_PRO_LN = 1,          // A = LayerNorm(sum of split-K slabs + bias + residual): the consumer of the previous GEMM
       SMALL_PRO_SELFATTN = 2 };  // A = self-attention context of the K slice's heads (q|k|v from partial sums, k/v appended)
enum { SMALL_EPI_PARTIAL = 0,     // out_part[kz][R][N]: raw split-K slabs
       SMALL_EPI_ACT_T = 1,       // act(acc + bias) -> operand type [R, ldc]
       SMALL_EPI_ACT_F32 = 2 };   // act(acc + bias) -> fp32 [R, ldc]
enum { SMALL_KV_F32 = 0, SMALL_KV_BF16 = 1, SMALL_KV_KV16 = 2 };

struct SmallLN {
    const float* part; int S;         // fp32 [S][R][D] slabs (S = 1: a finished fp32 row)
    const float* bias;                // [D] or null
    const float* resid;               // fp32 [R][D] or null: the row the batch path keeps in `dx`
    const float* gamma; const float* beta; float eps;
    float* x_out;                     // fp32 [R][D] or null: written by ONE workgroup (must not alias resid) - the LayerNorm output
    int x_is_sum;                     // (post-LN decoders: BLIP), or with x_is_sum the un-normalised sum y itself (pre-LN: CoCa,
                                      // whose residual stream is the running sum)
};
struct SmallSA {
    const float* qkv_part; const float* qkv_bias; int qkv_S;   // fp32 [qkv_S][R][3 H 64] partial sums of q|k|v, bias [3 H 64]
    void* kc; void* vc;               // self-attention cache of the layer: [R][H][kv_ld][64] in the attention's value type
    const int* anc; int anc_ld;       // ancestry table (beams) or null
    int kv_ld, n_keys, H;             // n_keys = t + 1 positions, the newest one from qkv_part
    const int* skip;                  // int32 [R] or null: rows of ended captions
};
struct SmallGemm {
    const void* W;                    // [N, K] operand type, torch Linear layout
    const void* A;                    // SMALL_PRO_GLOBAL
    int R, N, K, S;                   // S: K slices of the sum plan (captioner.hip::decode_splitk)
    int pro, epi, nchain;             // nchain: 4 = gemm_rows_kernel's sums, 1 = the register-staged tiles' (one chain)
    SmallLN ln;
    SmallSA sa;
    float* out_part;
    const float* bias; int act;       // act: 0 none, 1 exact-erf GELU, 2 ReLU
    void* out; int ldc;
};
struct SmallCross {
    const void* W; const float* bias; // query projection [D, D], bias [D]
    int R, D, H, S;                   // S: K slices of the query projection's sum plan
    SmallLN ln;                       // the row's LayerNorm (x_out written by the head-0 workgroup of a row)
    const void* kbase; const void* vbase; size_t kv_row0;      // the layer's k / v block and the launch's first row in it
    int rows_per_kv, kv_ld, n_keys, kv_kind;
    const int* skip;
    void* out;                        // context [R][D] in the operand type
};

int launch_small_gemm(int dtype, const SmallGemm& p, hipStream_t s);
int launch_small_cross(int dtype, const SmallCross& p, hipStream_t s);
int cap_g8_clamped_decode_small(unsigned long long* total, int reset);
