// C[M,N] = A[M,K] . W[N,K]^T (+bias, +epilogue) on gfx950 MFMA.  See gemm.hip.
#pragma once
#include "common.h"

enum GemmEpi {
    EPI_STORE = 0,     // C[row*ldc+col] = f(acc + bias) (+ resid)            out: T or f32
    EPI_PATCH = 1,     // ViT patch-embed: row (b,p) -> token row b*(P+1)+1+p, + pos[(1+p)*N+col]   out: f32
    EPI_CROSSKV = 2,   // decoder cross-attention K/V for all layers -> [layer][kv][image][head][token][64]  out: T
    EPI_QKVCACHE = 3,  // decoder self-attention: q -> qbuf, k/v -> cache[kv][row][head][t][64]  out: T
    EPI_PARTIAL = 4,   // split-K: slice z writes raw fp32 partial sums to C[z][M][ldc]; bias/residual/LayerNorm are
                       // applied by the consumer (launch_reduce_layernorm), in a fixed order -> deterministic
};

struct GemmParams {
    const void* A; int lda;        // activations, element type T, row-major [M,K]
    const void* W; int ldw;        // weights, element type T, row-major [N,K] (torch Linear layout)
    void* C; int ldc;              // output
    const float* bias;             // [N] fp32 or nullptr
    const float* resid; int ldr;   // fp32 [M,N] residual (may alias C) or nullptr
    int M, N, K;
    int gelu;                      // activation after bias: 0 none, 1 exact-erf GELU, 2 ReLU
    int out_f32;                   // 1: C is fp32, 0: C is T
    int epi;
    int splitk;                    // EPI_PARTIAL only: number of K slices (K % (slab*splitk) == 0)
    // EPI_PATCH: p0 = patches per image, aux = position table [(P+1), N] fp32
    // EPI_CROSSKV: p0 = tokens per image, p1 = heads, p2 = images   (N = layers*2*heads*64)
    // EPI_QKVCACHE: p0 = row capacity of the cache, p1 = heads, p2 = max positions, p3 = position t; C2 = cache base
    int p0, p1, p2, p3;
    const float* aux;
    void* C2;
    int kv16;                      // EPI_CROSSKV with an fp32 output: 1 = write the K/V rows as KV16 blocks (common.h: int16 + one
                                   // scale per head row; gemm_pp.hip is the only producer) instead of fp32; C = base of block (layer 0, k)
    int tile0, tile1;              // set by launch_big2 only: the range of 256 x 256 tiles one launch covers (0, 0 = all)
    const int* m_live;             // device int32 or null (tile 6, the decode "rows" kernel): row tiles that start at or beyond
                                   // *m_live return at once - the compacted greedy decode loop (ops.h, RowMap); M keeps
                                   // the slab stride / capacity meaning
};

// dtype: CAP_DT_F32 / CAP_DT_BF16.  tile: 0 = auto, 1 = 128x128, 2 = 64x64, 3 = 256x256 persistent LDS-DMA kernel
// (bf16: second generation), 4 = 256x256 register-staged (takes a residual operand), 5 = first-generation LDS-DMA kernel,
// 6 = decode "rows" kernel (64x64 tile, the block's K range split over its four waves with private LDS-DMA pipelines; bf16 /
// G8 with the plain-store or split-K epilogue - its sums are ordered differently from the other tiles': callers use it for
// a given (N, K) at EVERY row count or not at all),
// bf16 and G8: tile 3 = gemm_pp.hip (20 = the same, explicitly).  In a -DCAP_EXPERIMENTS build (python -m
// embodied_captioning_amd.build --experiments) also the kernels it replaced and the instrumented builds: bf16 10/11/12 =
// gemm_big2_kernel with the compiler / iglp_opt(0) / iglp_opt(1) schedule, 14/15 = gemm_big3_kernel (half-slab four-stage
// structure) without / with iglp_opt(1), 9 / 13 = instrumented generations one / two; G8 10 / 13 = gemm_big2_kernel<g8_t> and its
// instrumented build; 21 = instrumented gemm_pp_kernel (cycle stamps to GemmParams::aux; tools/gemm_cycles.py,
// tools/bench_gemm_pp.py) - the default library does not contain them
int launch_gemm(int dtype, const GemmParams& p, int tile, hipStream_t stream);

// 256x256 persistent kernel with the two wave groups half a stage apart (gemm_pp.hip; G8 and bf16 operands).  Returns -2 when the
// shape / epilogue is not one it takes (nothing launched, no error set).  launch_gemm reaches it as tile 20 (21: cycle stamps
// to p.aux in a -DCAP_EXPERIMENTS build).
int launch_gemm_pp(int dtype, const GemmParams& p, bool prof, hipStream_t stream);
