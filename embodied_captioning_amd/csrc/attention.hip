// Attention kernels of the captioner path (head_dim = 64 everywhere on this path).
//
//  vit_attention_mfma  bf16, whole K / V^T of one (image, head) resident in LDS (N <= 288 tokens), QK^T and PV on
//                      v_mfma_f32_32x32x16_bf16.  S^T = K.Q^T is computed so that a lane owns one query column:
//                      the softmax row-reduction is in-lane (+ one cross-half shuffle), and the S^T accumulator
//                      registers are fed straight back as the B operand of O^T = V^T.P^T (no LDS round trip).
//  vit_attention_flash bf16, 289..608 tokens (CoCa at 336x336): same LDS-resident K/V, keys walked in chunks with an
//                      online softmax because the score tile no longer fits in registers.
//  vit_attention_scalar  any dtype / any N: one thread per query, K/V tiles broadcast from LDS, online softmax in
//                      fp32.  Strict-fp32 mode uses it; tests use it to cross-check the MFMA kernel.
//  decode_attention    one new query per (row, head) against cached keys (self-attention cache with per-position
//                      ancestor indirection for beams, or the beam-shared cross-attention cache). HBM-bound:
//                      8 lanes x 16 B cover one 128-byte key row, 8 keys per wave-instruction.
#include <algorithm>

#include <type_traits>

#include "ops.h"
#include "decode_attn.h"

namespace {

__device__ __forceinline__ int swz_off(int row, int chunk) { return row * 128 + ((chunk ^ ((row >> 1) & 7)) << 4); }

constexpr float LOG2E = 1.4426950408889634f;
typedef short s16x4 __attribute__((ext_vector_type(4)));
#define CAP_GPTR(p) ((const __attribute__((address_space(1))) void*)(p))
#define CAP_LPTR(p) ((__attribute__((address_space(3))) void*)(p))

// ------------------------------------------------------------------------------------------------
// K and V of one (image, head) sit in LDS as [key][64] bf16 rows of 128 B with the same 16-byte-chunk XOR swizzle as the
// GEMM tiles, filled by LDS-DMA (swizzle on the source address).  Q.K^T reads K rows with ds_read_b128; P.V needs V
// "by column" (4 consecutive keys for one d per lane): ds_read_b64_tr_b16 delivers exactly that from the row-major
// image, so V is never transposed in memory.
template <int KB>
__global__ __launch_bounds__(256, 2) void vit_attention_mfma(const bf16_t* __restrict__ qkv, bf16_t* __restrict__ ctx,
                                                             int N, int H) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int NP = KB * 32;
    char* Ks = smem;                                   // [NP] rows of 128 B
    char* Vs = smem + NP * 128;                        // [NP] rows of 128 B
    const int b = blockIdx.x / H, h = blockIdx.x % H;
    const int D = H * 64, ld = 3 * D;
    const int tid = threadIdx.x, lane = tid & 63, r32 = lane & 31, hh = lane >> 5;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const bf16_t* base = qkv + (size_t)b * N * ld + h * 64;

    // 1 KiB pieces of 8 rows; piece p of K and of V go to wave p % 4
    for (int p = wave; p < NP / 8; p += 4) {
        const int row = p * 8 + (lane >> 3);
        const int gch = (lane & 7) ^ ((row >> 1) & 7);
        const bf16_t* src = base + (size_t)min(row, N - 1) * ld + gch * 8;
        __builtin_amdgcn_global_load_lds(CAP_GPTR(src + D), CAP_LPTR(Ks + p * 1024), 16, 0, 0);
        __builtin_amdgcn_global_load_lds(CAP_GPTR(src + 2 * D), CAP_LPTR(Vs + p * 1024), 16, 0, 0);
    }
    // this wave's query tiles (qt = wave, wave+4, ...): take their Q fragments now, so the loads fly with the K/V DMA
    constexpr int QT = (KB + 3) / 4;
    bf16x8 qfa[QT][4];
#pragma unroll
    for (int t = 0; t < QT; ++t) {
        const int qc = min((wave + 4 * t) * 32 + r32, N - 1);
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) qfa[t][ks] = *(const bf16x8*)(base + (size_t)qc * ld + ks * 16 + hh * 8);
    }
    __syncthreads();   // vmcnt(0) + barrier: every piece has landed

    // transposed-read addressing for the P.V A operand: 16-lane group g = lane>>4 handles d columns (g&1)*16 .. +15 and
    // key sub-block 4*(g>>1); inside a group lane 4q+p supplies row (key) q, columns 4p..4p+3
    const int tq = (lane >> 2) & 3, tp = lane & 3, tg = lane >> 4;
    const int nqt = (N + 31) / 32;
#pragma unroll
    for (int t = 0; t < QT; ++t) {
        const int qt = wave + 4 * t;
        if (qt >= nqt) break;
        const int q = qt * 32 + r32;
        const bf16x8 (&qf)[4] = qfa[t];

        f32x16 s[KB];
#pragma unroll
        for (int kb = 0; kb < KB; ++kb) {
#pragma unroll
            for (int e = 0; e < 16; ++e) s[kb][e] = 0.f;
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) {
                bf16x8 a = *(const bf16x8*)(Ks + swz_off(kb * 32 + r32, ks * 2 + hh));
                s[kb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, qf[ks], s[kb], 0, 0, 0);
            }
        }
        // softmax over keys for query column r32: key(kb, e) = kb*32 + (e&3) + 8*(e>>2) + 4*hh
        float m = -INFINITY;
#pragma unroll
        for (int kb = 0; kb < KB; ++kb)
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                if (kb == KB - 1) {      // only the last key block can hold padding keys
                    const int key = kb * 32 + (e & 3) + 8 * (e >> 2) + 4 * hh;
                    if (key >= N) s[kb][e] = -INFINITY;
                }
                m = fmaxf(m, s[kb][e]);
            }
        m = fmaxf(m, __shfl_xor(m, 32, 64));
        float l = 0.f;
        const float c1 = 0.125f * LOG2E;
#pragma unroll
        for (int kb = 0; kb < KB; ++kb)
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const float p = __builtin_amdgcn_exp2f((s[kb][e] - m) * c1);   // raw v_exp_f32: inputs are <= 0
                s[kb][e] = p;
                l += p;
            }
        l += __shfl_xor(l, 32, 64);

        f32x16 o[2];
#pragma unroll
        for (int db = 0; db < 2; ++db)
#pragma unroll
            for (int e = 0; e < 16; ++e) o[db][e] = 0.f;
#pragma unroll
        for (int kb = 0; kb < KB; ++kb)
#pragma unroll
            for (int s2 = 0; s2 < 2; ++s2) {
                // B operand: element j of lane half hh is P^T[key = kb*32 + 16*s2 + 8*(j>>2) + 4*hh + (j&3)][q]
                bf16x8 pb;
#pragma unroll
                for (int j = 0; j < 8; ++j) pb[j] = (bf16_t)s[kb][8 * s2 + j];
#pragma unroll
                for (int db = 0; db < 2; ++db) {
                    // A operand: lane (d = db*32 + r32, hh) needs V[key0 .. key0+3][d] and V[key0+8 .. +11][d],
                    // key0 = kb*32 + 16*s2 + 4*hh.  In its 16-lane group this lane ADDRESSES row key0+tq,
                    // columns dcol..dcol+3 and RECEIVES column (lane&15) of the 4 rows.
                    const int dcol = db * 32 + (tg & 1) * 16 + tp * 4;
                    const int key0 = kb * 32 + 16 * s2 + 4 * (tg >> 1) + tq;
                    const char* a0 = Vs + swz_off(key0, dcol >> 3) + (dcol & 7) * 2;
                    const char* a1 = Vs + swz_off(key0 + 8, dcol >> 3) + (dcol & 7) * 2;
                    const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)a0);
                    const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)a1);
                    const bf16x8 a = __builtin_bit_cast(bf16x8, __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7));
                    o[db] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, pb, o[db], 0, 0, 0);
                }
            }
        if (q < N) {
            const float inv = 1.0f / l;
            bf16_t* op = ctx + ((size_t)b * N + q) * D + h * 64;
#pragma unroll
            for (int db = 0; db < 2; ++db)
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    bf16x4 w;
#pragma unroll
                    for (int i = 0; i < 4; ++i) w[i] = (bf16_t)(o[db][4 * g + i] * inv);
                    *(bf16x4*)(op + db * 32 + 8 * g + 4 * hh) = w;
                }
        }
    }
}

// Long-sequence variant (CoCa ViT-L/14 at 336x336: 577 tokens, KB = 19): K and V of the (image, head) still fit in LDS
// (2 x 608 x 128 B = 152 KiB, one workgroup per CU), but the 32 x 608 score tile of a query block does not fit in
// registers, so the keys are walked in chunks of KC blocks with an online softmax (running max m, running sum l,
// O rescaled by exp2((m_old - m_new) c)) - same MFMA dataflow per chunk as vit_attention_mfma.
template <int KB, int KC>
__global__ __launch_bounds__(256, 1) void vit_attention_flash(const bf16_t* __restrict__ qkv, bf16_t* __restrict__ ctx,
                                                              int N, int H) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int NP = KB * 32;
    char* Ks = smem;
    char* Vs = smem + NP * 128;
    const int b = blockIdx.x / H, h = blockIdx.x % H;
    const int D = H * 64, ld = 3 * D;
    const int tid = threadIdx.x, lane = tid & 63, r32 = lane & 31, hh = lane >> 5;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const bf16_t* base = qkv + (size_t)b * N * ld + h * 64;

    for (int p = wave; p < NP / 8; p += 4) {
        const int row = p * 8 + (lane >> 3);
        const int gch = (lane & 7) ^ ((row >> 1) & 7);
        const bf16_t* src = base + (size_t)min(row, N - 1) * ld + gch * 8;
        __builtin_amdgcn_global_load_lds(CAP_GPTR(src + D), CAP_LPTR(Ks + p * 1024), 16, 0, 0);
        __builtin_amdgcn_global_load_lds(CAP_GPTR(src + 2 * D), CAP_LPTR(Vs + p * 1024), 16, 0, 0);
    }
    const int nqt = (N + 31) / 32;
    auto load_q = [&](int qt, bf16x8 (&qf)[4]) {
        const int qc = min(qt * 32 + r32, N - 1);
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) qf[ks] = *(const bf16x8*)(base + (size_t)qc * ld + ks * 16 + hh * 8);
    };
    bf16x8 qf[4], qn[4];
    load_q(min(wave, nqt - 1), qf);
    __syncthreads();   // vmcnt(0) + barrier: every piece has landed

    const int tq = (lane >> 2) & 3, tp = lane & 3, tg = lane >> 4;
    const float c1 = 0.125f * LOG2E;
    for (int qt = wave; qt < nqt; qt += 4) {
        load_q(min(qt + 4, nqt - 1), qn);                 // next tile's Q flies under this tile's MFMAs
        const int q = qt * 32 + r32;
        float m = -INFINITY, l = 0.f;
        f32x16 o[2];
#pragma unroll
        for (int db = 0; db < 2; ++db)
#pragma unroll
            for (int e = 0; e < 16; ++e) o[db][e] = 0.f;
#pragma unroll
        for (int c0 = 0; c0 < KB; c0 += KC) {
            f32x16 s[KC];
            float cm = -INFINITY;
#pragma unroll
            for (int kc = 0; kc < KC; ++kc) {
                const int kb = c0 + kc;
                if (kb < KB) {
#pragma unroll
                    for (int e = 0; e < 16; ++e) s[kc][e] = 0.f;
#pragma unroll
                    for (int ks = 0; ks < 4; ++ks) {
                        bf16x8 a = *(const bf16x8*)(Ks + swz_off(kb * 32 + r32, ks * 2 + hh));
                        s[kc] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, qf[ks], s[kc], 0, 0, 0);
                    }
#pragma unroll
                    for (int e = 0; e < 16; ++e) {
                        if (kb == KB - 1) {      // only the last key block can hold padding keys
                            const int key = kb * 32 + (e & 3) + 8 * (e >> 2) + 4 * hh;
                            if (key >= N) s[kc][e] = -INFINITY;
                        }
                        cm = fmaxf(cm, s[kc][e]);
                    }
                }
            }
            cm = fmaxf(cm, __shfl_xor(cm, 32, 64));
            const float mn = fmaxf(m, cm);
            const float alpha = __builtin_amdgcn_exp2f((m - mn) * c1);     // first chunk: exp2(-inf) = 0, l and o are 0
            l *= alpha;
#pragma unroll
            for (int db = 0; db < 2; ++db)
#pragma unroll
                for (int e = 0; e < 16; ++e) o[db][e] *= alpha;
            m = mn;
#pragma unroll
            for (int kc = 0; kc < KC; ++kc) {
                const int kb = c0 + kc;
                if (kb < KB) {
#pragma unroll
                    for (int e = 0; e < 16; ++e) {
                        const float pv = __builtin_amdgcn_exp2f((s[kc][e] - mn) * c1);
                        s[kc][e] = pv;
                        l += pv;
                    }
#pragma unroll
                    for (int s2 = 0; s2 < 2; ++s2) {
                        bf16x8 pb;
#pragma unroll
                        for (int j = 0; j < 8; ++j) pb[j] = (bf16_t)s[kc][8 * s2 + j];
#pragma unroll
                        for (int db = 0; db < 2; ++db) {
                            const int dcol = db * 32 + (tg & 1) * 16 + tp * 4;
                            const int key0 = kb * 32 + 16 * s2 + 4 * (tg >> 1) + tq;
                            const char* a0 = Vs + swz_off(key0, dcol >> 3) + (dcol & 7) * 2;
                            const char* a1 = Vs + swz_off(key0 + 8, dcol >> 3) + (dcol & 7) * 2;
                            const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)a0);
                            const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)a1);
                            const bf16x8 a = __builtin_bit_cast(bf16x8, __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7));
                            o[db] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, pb, o[db], 0, 0, 0);
                        }
                    }
                }
            }
        }
        l += __shfl_xor(l, 32, 64);
        if (q < N) {
            const float inv = 1.0f / l;
            bf16_t* op = ctx + ((size_t)b * N + q) * D + h * 64;
#pragma unroll
            for (int db = 0; db < 2; ++db)
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    bf16x4 w;
#pragma unroll
                    for (int i = 0; i < 4; ++i) w[i] = (bf16_t)(o[db][4 * g + i] * inv);
                    *(bf16x4*)(op + db * 32 + 8 * g + 4 * hh) = w;
                }
        }
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) qf[ks] = qn[ks];
    }
}

// Heads wider than 64 (BLIP-2's ViT-g/14: 88): the head is handled as TWO 64-wide halves, each with its own K and V
// image in LDS (the swizzled 128-byte-row layout of the kernels above; columns beyond head_dim are zero - the images are
// cleared first and the LDS-DMA lanes of missing columns are masked off).  S = Q_lo.K_lo^T + Q_hi.K_hi^T is one longer MFMA
// chain, O has four 32-wide column blocks.  257 tokens: 4 x 288 x 128 B = 144 KiB, one workgroup per CU; keys in chunks of
// KC blocks with the online softmax of vit_attention_flash.
template <int KB, int KC>
__global__ __launch_bounds__(256, 1) void vit_attention_flash_wide(const bf16_t* __restrict__ qkv, bf16_t* __restrict__ ctx,
                                                                   int N, int H, int hd, int causal) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int NP = KB * 32, IMG = NP * 128;
    char* Ks = smem;                                   // [2][NP] rows of 128 B
    char* Vs = smem + 2 * IMG;
    const int b = blockIdx.x / H, h = blockIdx.x % H;
    const int D = H * hd, ld = 3 * D;
    const int tid = threadIdx.x, lane = tid & 63, r32 = lane & 31, hh = lane >> 5;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const bf16_t* base = qkv + (size_t)b * N * ld + h * hd;

    if (hd < 128) {                                    // clear the upper-half images: their missing columns must read as zero
        for (int i = tid; i < IMG / 16; i += 256) {
            *(f32x4*)(Ks + IMG + i * 16) = f32x4(0.f);
            *(f32x4*)(Vs + IMG + i * 16) = f32x4(0.f);
        }
        __syncthreads();
    }
    for (int p = wave; p < NP / 8; p += 4) {
        const int row = p * 8 + (lane >> 3);
        const int gch = (lane & 7) ^ ((row >> 1) & 7);
        const bf16_t* src = base + (size_t)min(row, N - 1) * ld + gch * 8;
#pragma unroll
        for (int hf = 0; hf < 2; ++hf) {
            if (hf * 64 + gch * 8 < hd) {              // whole 16-byte chunks: head_dim is a multiple of 8
                __builtin_amdgcn_global_load_lds(CAP_GPTR(src + hf * 64 + D), CAP_LPTR(Ks + hf * IMG + p * 1024), 16, 0, 0);
                __builtin_amdgcn_global_load_lds(CAP_GPTR(src + hf * 64 + 2 * D), CAP_LPTR(Vs + hf * IMG + p * 1024), 16, 0, 0);
            }
        }
    }
    const int nqt = (N + 31) / 32;
    auto load_q = [&](int qt, bf16x8 (&qf)[8]) {
        const int qc = min(qt * 32 + r32, N - 1);
#pragma unroll
        for (int ks = 0; ks < 8; ++ks) {
            const int col = ks * 16 + hh * 8;
            if (col < hd) qf[ks] = *(const bf16x8*)(base + (size_t)qc * ld + col);
            else
#pragma unroll
                for (int i = 0; i < 8; ++i) qf[ks][i] = (bf16_t)0.f;
        }
    };
    bf16x8 qf[8];
    __syncthreads();   // vmcnt(0) + barrier: every piece has landed

    const int tq = (lane >> 2) & 3, tp = lane & 3, tg = lane >> 4;
    const float c1 = LOG2E / sqrtf((float)hd);
    for (int qt = wave; qt < nqt; qt += 4) {
        load_q(qt, qf);
        const int q = qt * 32 + r32;
        float m = -INFINITY, l = 0.f;
        f32x16 o[4];
#pragma unroll
        for (int db = 0; db < 4; ++db)
#pragma unroll
            for (int e = 0; e < 16; ++e) o[db][e] = 0.f;
#pragma unroll
        for (int c0 = 0; c0 < KB; c0 += KC) {
            f32x16 s[KC];
            float cm = -INFINITY;
#pragma unroll
            for (int kc = 0; kc < KC; ++kc) {
                const int kb = c0 + kc;
                if (kb < KB) {
#pragma unroll
                    for (int e = 0; e < 16; ++e) s[kc][e] = 0.f;
#pragma unroll
                    for (int ks = 0; ks < 8; ++ks) {
                        bf16x8 a = *(const bf16x8*)(Ks + (ks >> 2) * IMG + swz_off(kb * 32 + r32, (ks & 3) * 2 + hh));
                        s[kc] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, qf[ks], s[kc], 0, 0, 0);
                    }
#pragma unroll
                    for (int e = 0; e < 16; ++e) {
                        if (kb == KB - 1 || causal) {      // causal (decoder prefill): a query sees the keys up to itself
                            const int key = kb * 32 + (e & 3) + 8 * (e >> 2) + 4 * hh;
                            if (key >= N || (causal && key > q)) s[kc][e] = -INFINITY;
                        }
                        cm = fmaxf(cm, s[kc][e]);
                    }
                }
            }
            cm = fmaxf(cm, __shfl_xor(cm, 32, 64));
            const float mn = fmaxf(m, cm);
            const float alpha = __builtin_amdgcn_exp2f((m - mn) * c1);
            l *= alpha;
#pragma unroll
            for (int db = 0; db < 4; ++db)
#pragma unroll
                for (int e = 0; e < 16; ++e) o[db][e] *= alpha;
            m = mn;
#pragma unroll
            for (int kc = 0; kc < KC; ++kc) {
                const int kb = c0 + kc;
                if (kb < KB) {
#pragma unroll
                    for (int e = 0; e < 16; ++e) {
                        const float pv = __builtin_amdgcn_exp2f((s[kc][e] - mn) * c1);
                        s[kc][e] = pv;
                        l += pv;
                    }
#pragma unroll
                    for (int s2 = 0; s2 < 2; ++s2) {
                        bf16x8 pb;
#pragma unroll
                        for (int j = 0; j < 8; ++j) pb[j] = (bf16_t)s[kc][8 * s2 + j];
#pragma unroll
                        for (int db = 0; db < 4; ++db) {
                            const int dcol = (db & 1) * 32 + (tg & 1) * 16 + tp * 4;          // column inside the 64-wide half db >> 1
                            const int key0 = kb * 32 + 16 * s2 + 4 * (tg >> 1) + tq;
                            const char* vimg = Vs + (db >> 1) * IMG;
                            const char* a0 = vimg + swz_off(key0, dcol >> 3) + (dcol & 7) * 2;
                            const char* a1 = vimg + swz_off(key0 + 8, dcol >> 3) + (dcol & 7) * 2;
                            const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)a0);
                            const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)a1);
                            const bf16x8 a = __builtin_bit_cast(bf16x8, __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7));
                            o[db] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, pb, o[db], 0, 0, 0);
                        }
                    }
                }
            }
        }
        l += __shfl_xor(l, 32, 64);
        if (q < N) {
            const float inv = 1.0f / l;
            bf16_t* op = ctx + ((size_t)b * N + q) * D + h * hd;
#pragma unroll
            for (int db = 0; db < 4; ++db)
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const int d = db * 32 + 8 * g + 4 * hh;
                    if (d < hd) {                      // head_dim is a multiple of 8 and d of 4: d + 3 < head_dim whenever d < head_dim
                        bf16x4 w;
#pragma unroll
                        for (int i = 0; i < 4; ++i) w[i] = (bf16_t)(o[db][4 * g + i] * inv);
                        *(bf16x4*)(op + d) = w;
                    }
                }
        }
    }
}

// ------------------------------------------------------------------------------------------------
constexpr int SC_KT = 128;   // keys per LDS tile
template <typename T, typename TO = T>
__global__ __launch_bounds__(256) void vit_attention_scalar(const T* __restrict__ qkv, TO* __restrict__ ctx, int N, int H) {
    __shared__ float Ks[SC_KT][65];
    __shared__ float Vs[SC_KT][65];
    const int b = blockIdx.x / H, h = blockIdx.x % H;
    const int D = H * 64, ld = 3 * D;
    const int tid = threadIdx.x;
    const int q = blockIdx.y * 256 + tid;
    const T* base = qkv + (size_t)b * N * ld + h * 64;
    float qv[64], o[64];
    const int qc = min(q, N - 1);
#pragma unroll
    for (int d = 0; d < 64; ++d) { qv[d] = to_f32(base[(size_t)qc * ld + d]) * 0.125f; o[d] = 0.f; }
    float m = -INFINITY, l = 0.f;
    for (int k0 = 0; k0 < N; k0 += SC_KT) {
        const int nk = min(SC_KT, N - k0);
        __syncthreads();
        for (int c = tid; c < nk * 64; c += 256) {
            const int j = c >> 6, d = c & 63;
            Ks[j][d] = to_f32(base[(size_t)(k0 + j) * ld + D + d]);
            Vs[j][d] = to_f32(base[(size_t)(k0 + j) * ld + 2 * D + d]);
        }
        __syncthreads();
        for (int j = 0; j < nk; ++j) {
            float sc = 0.f;
#pragma unroll
            for (int d = 0; d < 64; ++d) sc = fmaf(qv[d], Ks[j][d], sc);
            const float mn = fmaxf(m, sc);
            const float c = expf(m - mn), p = expf(sc - mn);
            l = l * c + p;
#pragma unroll
            for (int d = 0; d < 64; ++d) o[d] = fmaf(p, Vs[j][d], o[d] * c);
            m = mn;
        }
    }
    if (q < N) {
        const float inv = 1.0f / l;
        TO* op = ctx + ((size_t)b * N + q) * D;
#pragma unroll
        for (int d = 0; d < 64; d += 4)
            store4(op, h * 64 + d, make_float4(o[d] * inv, o[d + 1] * inv, o[d + 2] * inv, o[d + 3] * inv));
    }
}

// ------------------------------------------------------------------------------------------------
// fp32 ViT attention on the matrix pipe (strict-fp32 and split-fp16 modes: qkv arrives as fp32).  v_mfma_f32_32x32x2_f32
// multiplies exact fp32 products (157 TFLOP/s peak, 1/16 of the bf16 rate - still ~10x the one-thread-per-query kernel
// above, which it replaces for 1 / 7 / 9 key blocks).  Same dataflow as vit_attention_mfma: K and V of the (image, head)
// sit in LDS, S^T = K.Q^T so a lane owns one query column and the S^T accumulators are the B operand of O^T = V^T.P^T.
//   * LDS rows of 64 fp32 with a pitch of 68 floats: lane (r32, hh) reads the 16-byte chunk 2 ks + hh of key row r32
//     (its 4 values feed 4 MFMAs as the k-pair (hh = 0, hh = 1); Q uses the same k permutation, so the sum is unchanged)
//     - with pitch 68 the 16 lanes of a ds_read_b128 phase hit 16 distinct 4-bank groups;
//   * P.V: MFMA #e of a key block takes keys {kappa_e, kappa_e + 4} = exactly what accumulator register e of the two lane
//     halves holds, so P never leaves its registers; V^T[d][key] is a ds_read_b32 of 32 consecutive floats per half.
// One workgroup per (image, head), one wave per SIMD (the MFMA chains are long: 32 dependent MFMAs per score tile).
template <int KB, typename TO>
__global__ __launch_bounds__(256, 1) void vit_attention_f32_mfma(const float* __restrict__ qkv, TO* __restrict__ ctx, int N, int H) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int NP = KB * 32, PITCH = 68;
    float* Ks = (float*)smem;
    float* Vs = Ks + NP * PITCH;
    const int b = blockIdx.x / H, h = blockIdx.x % H;
    const int D = H * 64, ld = 3 * D;
    const int tid = threadIdx.x, lane = tid & 63, r32 = lane & 31, hh = lane >> 5;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const float* base = qkv + (size_t)b * N * ld + h * 64;

    for (int i = tid; i < NP * 16; i += 256) {
        const int row = i >> 4, ch = i & 15;
        const float* src = base + (size_t)min(row, N - 1) * ld + ch * 4;
        *(f32x4*)(Ks + row * PITCH + ch * 4) = *(const f32x4*)(src + D);
        *(f32x4*)(Vs + row * PITCH + ch * 4) = *(const f32x4*)(src + 2 * D);
    }
    __syncthreads();

    const int nqt = (N + 31) / 32;
    const float c1 = 0.125f * LOG2E;
    for (int qt = wave; qt < nqt; qt += 4) {
        const int q = qt * 32 + r32, qc = min(q, N - 1);
        f32x4 qf[8];
#pragma unroll
        for (int ks = 0; ks < 8; ++ks) qf[ks] = *(const f32x4*)(base + (size_t)qc * ld + (2 * ks + hh) * 4);

        f32x16 s[KB];
#pragma unroll
        for (int kb = 0; kb < KB; ++kb) {
#pragma unroll
            for (int e = 0; e < 16; ++e) s[kb][e] = 0.f;
#pragma unroll
            for (int ks = 0; ks < 8; ++ks) {
                const f32x4 a = *(const f32x4*)(Ks + (kb * 32 + r32) * PITCH + (2 * ks + hh) * 4);
#pragma unroll
                for (int j = 0; j < 4; ++j) s[kb] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[j], qf[ks][j], s[kb], 0, 0, 0);
            }
            __builtin_amdgcn_sched_barrier(0);     // keep the next block's K reads from being hoisted over this chain (spills)
        }
        // softmax over keys for query column r32: key(kb, e) = kb*32 + (e&3) + 8*(e>>2) + 4*hh
        float m = -INFINITY;
#pragma unroll
        for (int kb = 0; kb < KB; ++kb)
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                if (kb == KB - 1) {
                    const int key = kb * 32 + (e & 3) + 8 * (e >> 2) + 4 * hh;
                    if (key >= N) s[kb][e] = -INFINITY;
                }
                m = fmaxf(m, s[kb][e]);
            }
        m = fmaxf(m, __shfl_xor(m, 32, 64));
        float l = 0.f;
#pragma unroll
        for (int kb = 0; kb < KB; ++kb)
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const float p = __builtin_amdgcn_exp2f((s[kb][e] - m) * c1);
                s[kb][e] = p;
                l += p;
            }
        l += __shfl_xor(l, 32, 64);

        f32x16 o[2];
#pragma unroll
        for (int db = 0; db < 2; ++db)
#pragma unroll
            for (int e = 0; e < 16; ++e) o[db][e] = 0.f;
#pragma unroll
        for (int kb = 0; kb < KB; ++kb)
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const int key = kb * 32 + (e & 3) + 8 * (e >> 2) + 4 * hh;
#pragma unroll
                for (int db = 0; db < 2; ++db)
                    o[db] = __builtin_amdgcn_mfma_f32_32x32x2f32(Vs[key * PITCH + db * 32 + r32], s[kb][e], o[db], 0, 0, 0);
                if ((e & 3) == 3) __builtin_amdgcn_sched_barrier(0);
            }
        if (q < N) {
            const float inv = 1.0f / l;
            TO* op = ctx + ((size_t)b * N + q) * D;
#pragma unroll
            for (int db = 0; db < 2; ++db)
#pragma unroll
                for (int g = 0; g < 4; ++g)
                    store4(op, h * 64 + db * 32 + 8 * g + 4 * hh,
                           make_float4(o[db][4 * g] * inv, o[db][4 * g + 1] * inv, o[db][4 * g + 2] * inv, o[db][4 * g + 3] * inv));
        }
    }
}


// ------------------------------------------------------------------------------------------------
// The same exact-product fp32 MFMA attention for heads of ANY width up to 96 (a multiple of 8: BLIP-2's ViT-g/14 has 88) and
// any token count: K and V of such a head do not fit in LDS whole (257 keys x 88 dims x 4 B x 2 = 181 KB), so the keys are
// walked in chunks of KC blocks of 32 with an online softmax across chunks (running maximum, denominator and rescaled context -
// the arithmetic of vit_attention_split's chunked mode).  A workgroup = one (image, head, group of 4 query tiles), one 32-query
// tile per wave whose state (m, l, O) lives in registers; the K / V chunk is shared by the 4 waves (KC = 3: 2 x 96 rows x 100
// floats = 77 KB, two workgroups per CU).  V columns beyond the head width are zero in LDS (the third 32-wide block of O is
// partly padding and is not stored).  fp32 q | k | v in (what the qkv GEMM writes in fp32 mode, and in split mode when the head is
// not 64 wide), context out as fp32 or G8.  Replaces the VALU kernel generic_attention_kernel for the ViT tower: BLIP-2 OPT-2.7b
// geometry, 32 frames: 45 ms of ViT-g attention per generate before.
template <int KC, int NDB, typename TO>
__global__ __launch_bounds__(256, 2) void vit_attention_f32_mfma_wide(const float* __restrict__ qkv, TO* __restrict__ ctx, int N, int H,
                                                                      int HD, int QG) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int NPC = KC * 32, PITCH = NDB * 32 + 4;    // floats per K / V row in LDS (conflict-free ds_read_b128 as with 68)
    float* Ks = (float*)smem;
    float* Vs = Ks + NPC * PITCH;
    const int qg = blockIdx.x % QG, bh = blockIdx.x / QG, b = bh / H, h = bh % H;
    const int D = H * HD, ld = 3 * D, nks = HD >> 3, nch = HD >> 2;
    const int tid = threadIdx.x, lane = tid & 63, r32 = lane & 31, hh = lane >> 5;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const float* base = qkv + (size_t)b * N * ld + h * HD;
    const int nqt = (N + 31) / 32, nkb = nqt;
    const float c1 = LOG2E / sqrtf((float)HD);

    const int qt = qg * 4 + wave;                         // this wave's query tile (past the end: it still helps to fill the chunk)
    const bool live = qt < nqt;
    const int q = qt * 32 + r32, qc = min(q, N - 1);
    f32x4 qf[12];                                         // up to 96 dims: k-step ks covers dims 8 ks .. 8 ks + 7 (4 per lane half)
#pragma unroll
    for (int ks = 0; ks < 12; ++ks)
        qf[ks] = ks < nks ? *(const f32x4*)(base + (size_t)qc * ld + (2 * ks + hh) * 4) : f32x4(0.f);
    float m = -INFINITY, l = 0.f;
    f32x16 o[NDB];
#pragma unroll
    for (int db = 0; db < NDB; ++db)
#pragma unroll
        for (int e = 0; e < 16; ++e) o[db][e] = 0.f;

    for (int c0 = 0; c0 < nkb; c0 += KC) {
        if (c0 > 0) __syncthreads();                      // every wave is done with the previous chunk
        for (int i = tid; i < NPC * (PITCH / 4); i += 256) {
            const int row = i / (PITCH / 4), ch = i - row * (PITCH / 4);
            f32x4 kv = 0.f, vv = 0.f;
            if (ch < nch) {
                const float* src = base + (size_t)min(c0 * 32 + row, N - 1) * ld + ch * 4;
                kv = *(const f32x4*)(src + D);
                vv = *(const f32x4*)(src + 2 * D);
            }
            *(f32x4*)(Ks + row * PITCH + ch * 4) = kv;
            *(f32x4*)(Vs + row * PITCH + ch * 4) = vv;    // zero beyond the head width: the padding columns of O stay 0
        }
        __syncthreads();
        if (!live) continue;
        f32x16 s[KC];
#pragma unroll
        for (int kc = 0; kc < KC; ++kc) {
#pragma unroll
            for (int e = 0; e < 16; ++e) s[kc][e] = 0.f;
            if (c0 + kc < nkb) {
#pragma unroll
                for (int ks = 0; ks < 12; ++ks) {
                    if (ks < nks) {
                        const f32x4 a = *(const f32x4*)(Ks + (kc * 32 + r32) * PITCH + (2 * ks + hh) * 4);
#pragma unroll
                        for (int j = 0; j < 4; ++j) s[kc] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[j], qf[ks][j], s[kc], 0, 0, 0);
                    }
                }
            }
            __builtin_amdgcn_sched_barrier(0);
        }
        // scores of this chunk for query column r32: key(kc, e) = (c0 + kc) * 32 + (e & 3) + 8 (e >> 2) + 4 hh
        float cm = -INFINITY;
#pragma unroll
        for (int kc = 0; kc < KC; ++kc)
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const int key = (c0 + kc) * 32 + (e & 3) + 8 * (e >> 2) + 4 * hh;
                if (key >= N) s[kc][e] = -INFINITY;       // padding keys and key blocks past the end
                cm = fmaxf(cm, s[kc][e]);
            }
        cm = fmaxf(cm, __shfl_xor(cm, 32, 64));
        const float mn = fmaxf(m, cm);
        const float alpha = __builtin_amdgcn_exp2f((m - mn) * c1);      // first chunk: exp2(-inf) = 0 and l, o are 0
        l *= alpha;
#pragma unroll
        for (int db = 0; db < NDB; ++db)
#pragma unroll
            for (int e = 0; e < 16; ++e) o[db][e] *= alpha;
        m = mn;
#pragma unroll
        for (int kc = 0; kc < KC; ++kc)
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const float pe = __builtin_amdgcn_exp2f((s[kc][e] - mn) * c1);
                s[kc][e] = pe;
                l += pe;
            }
#pragma unroll
        for (int kc = 0; kc < KC; ++kc) {
            if (c0 + kc >= nkb) continue;                 // (its probabilities are all zero)
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const int key = kc * 32 + (e & 3) + 8 * (e >> 2) + 4 * hh;
#pragma unroll
                for (int db = 0; db < NDB; ++db)
                    o[db] = __builtin_amdgcn_mfma_f32_32x32x2f32(Vs[key * PITCH + db * 32 + r32], s[kc][e], o[db], 0, 0, 0);
                if ((e & 3) == 3) __builtin_amdgcn_sched_barrier(0);
            }
        }
    }
    l += __shfl_xor(l, 32, 64);
    if (live && q < N) {
        const float inv = 1.0f / l;
        TO* op = ctx + ((size_t)b * N + q) * D;
#pragma unroll
        for (int db = 0; db < NDB; ++db)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int d0 = db * 32 + 8 * g + 4 * hh;       // 4 consecutive dims; HD % 4 == 0: inside or outside together
                if (d0 < HD)
                    store4(op, h * HD + d0, make_float4(o[db][4 * g] * inv, o[db][4 * g + 1] * inv, o[db][4 * g + 2] * inv, o[db][4 * g + 3] * inv));
            }
    }
}

// ------------------------------------------------------------------------------------------------
// Split-fp16 ViT attention (CAP_F32_SPLIT): q|k|v arrive as G8 (the qkv GEMM writes its output that way), both matrix
// products run on the fp16 MFMA pipe as hi.lo + lo.hi + hi.hi (common.h), softmax in fp32, context out as G8.  Dataflow of
// vit_attention_mfma: S^T = K.Q^T (a lane owns a query column), O^T = V^T.P^T with P taken from the S^T accumulators.
//   * a head's 64 dims of one token are 256 contiguous bytes of the G8 row = 16 chunks [H0 L0 H1 L1 .. H7 L7]; K and V sit in
//     LDS as rows of 256 B filled by LDS-DMA, chunk c of row r stored at slot c ^ (r & 15) (swizzle applied on the DMA's
//     source address) so the 16 lanes of a ds_read_b128 phase - 16 consecutive keys, same chunk - hit 16 different slots;
//   * lane (r32, hh) of a 32x32x16 MFMA supplies 8 consecutive d = group 2 ks + hh: hi chunk 2 (2 ks + hh), lo chunk + 1;
//   * P.V: V^T fragments through ds_read_b64_tr_b16 from the hi and the lo chunks; P (fp32, in (0, 1]) is split in registers.
// Any token count: a workgroup = one (image, head, group of 8 query tiles), EIGHT waves = one 32-query tile each (two per
// SIMD, so one wave's softmax / P-split VALU work runs under the other's MFMAs); the keys are walked in chunks of KC key
// blocks (KC = 7: 224 keys, 112 KiB of LDS) with an online softmax across chunks - 197 tokens are one chunk and one group
// (the arithmetic of a single-pass softmax), 577 tokens (384-pixel checkpoints) three chunks x three groups.
// ONE: the caller guarantees a single chunk (nkb <= KC): no rescaling state, the single-pass arithmetic.  EXACT (with ONE):
// nkb == KC, every key block exists and only the last one has padding keys - no per-block guards in the MFMA chains (197
// tokens: 218 us per launch at 256 x 12 heads, 255 us with the guards).
template <int KC, bool ONE, bool EXACT = false>
__global__ __launch_bounds__(512, 1) void vit_attention_split(const g8_t* __restrict__ qkv, g8_t* __restrict__ ctx, int N, int H, int QG) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int NP = KC * 32;
    char* Ks = smem;                                   // [NP] rows of 256 B
    char* Vs = smem + NP * 256;
    const int qg = blockIdx.x % QG, bh = blockIdx.x / QG, b = bh / H, h = bh % H;
    const int D = H * 64, ld = 3 * D;
    const int tid = threadIdx.x, lane = tid & 63, r32 = lane & 31, hh = lane >> 5;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const char* base = (const char*)(qkv + (size_t)b * N * ld + h * 64);      // byte address of (token 0, q dims of head h)
    const size_t rowb = (size_t)ld * 4;
    auto koff = [](int row, int chunk) { return row * 256 + ((chunk ^ (row & 15)) << 4); };
    // V is read by ds_read_b64_tr_b16, served in two groups of 32 lanes: 4 consecutive keys x the 4 same-parity chunks of half a
    // row x 8 bytes.  Key r & 15 folds those 16 pieces onto 8 slots (2-way conflict on every V read); keys {0, 1, 8, 9} by r & 3
    // send the four rows' chunk sets to {0 2 4 6}, {1 3 5 7}, {8 ..}, {9 ..} (+ 8 for the other half): 16 slots, all 64 banks.
    auto voff = [](int row, int chunk) { return row * 256 + ((chunk ^ ((row & 1) | ((row & 2) << 2))) << 4); };
    const int tq = (lane >> 2) & 3, tp = lane & 3, tg = lane >> 4;
    const int nqt = (N + 31) / 32, nkb = nqt;
    const float c1 = 0.125f * LOG2E;

    const int qt = qg * 8 + wave;                      // this wave's query tile (may be past the end: it still helps with the DMA)
    const bool live = qt < nqt;
    const int q = qt * 32 + r32, qc = min(q, N - 1);
    f16x8 qh[4], ql[4];
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
        const char* qp = base + (size_t)qc * rowb + (2 * ks + hh) * 32;
        qh[ks] = *(const f16x8*)qp;
        ql[ks] = *(const f16x8*)(qp + 16);
    }
    float m = -INFINITY, l = 0.f;
    f32x16 o[2];
#pragma unroll
    for (int db = 0; db < 2; ++db)
#pragma unroll
        for (int e = 0; e < 16; ++e) o[db][e] = 0.f;

    for (int c0 = 0; c0 < (ONE ? 1 : nkb); c0 += KC) {
        if (c0 > 0) __syncthreads();                   // every wave is done with the previous chunk's K / V
        // 1 KiB pieces of 4 rows; piece p of K and of V go to wave p % 8
        for (int p = wave; p < NP / 4; p += 8) {
            const int row = p * 4 + (lane >> 4);
            const int c = (lane & 15) ^ (row & 15);
            const int cv = (lane & 15) ^ ((row & 1) | ((row & 2) << 2));
            const char* src = base + (size_t)min(c0 * 32 + row, N - 1) * rowb;
            __builtin_amdgcn_global_load_lds(CAP_GPTR(src + (size_t)D * 4 + c * 16), CAP_LPTR(Ks + p * 1024), 16, 0, 0);
            __builtin_amdgcn_global_load_lds(CAP_GPTR(src + (size_t)2 * D * 4 + cv * 16), CAP_LPTR(Vs + p * 1024), 16, 0, 0);
        }
        __syncthreads();                               // vmcnt(0) + barrier: every piece has landed
        if (!live) continue;
        f32x16 s[KC];
        float cm = -INFINITY;
#pragma unroll
        for (int kc = 0; kc < KC; ++kc) {
#pragma unroll
            for (int e = 0; e < 16; ++e) s[kc][e] = 0.f;
            if (EXACT || c0 + kc < nkb) {
#pragma unroll
                for (int ks = 0; ks < 4; ++ks) {
                    const f16x8 kh = *(const f16x8*)(Ks + koff(kc * 32 + r32, 2 * (2 * ks + hh)));
                    const f16x8 kl = *(const f16x8*)(Ks + koff(kc * 32 + r32, 2 * (2 * ks + hh) + 1));
                    s[kc] = __builtin_amdgcn_mfma_f32_32x32x16_f16(kh, ql[ks], s[kc], 0, 0, 0);
                    s[kc] = __builtin_amdgcn_mfma_f32_32x32x16_f16(kl, qh[ks], s[kc], 0, 0, 0);
                    s[kc] = __builtin_amdgcn_mfma_f32_32x32x16_f16(kh, qh[ks], s[kc], 0, 0, 0);
                }
            }
            __builtin_amdgcn_sched_barrier(0);      // keep later K reads from piling up over this chain (spills at 256 VGPRs)
        }
        // scores of this chunk for query column r32: key(kc, e) = (c0 + kc)*32 + (e&3) + 8*(e>>2) + 4*hh
#pragma unroll
        for (int kc = 0; kc < KC; ++kc)
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                if (!EXACT || kc == KC - 1) {
                    const int key = (c0 + kc) * 32 + (e & 3) + 8 * (e >> 2) + 4 * hh;
                    if (key >= N) s[kc][e] = -INFINITY;     // padding keys and key blocks past the end
                }
                cm = fmaxf(cm, s[kc][e]);
            }
        cm = fmaxf(cm, __shfl_xor(cm, 32, 64));
        float mn = cm;
        if constexpr (!ONE) {
            mn = fmaxf(m, cm);
            const float alpha = __builtin_amdgcn_exp2f((m - mn) * c1);      // first chunk: exp2(-inf) = 0 and l, o are 0
            l *= alpha;
#pragma unroll
            for (int db = 0; db < 2; ++db)
#pragma unroll
                for (int e = 0; e < 16; ++e) o[db][e] *= alpha;
            m = mn;
        }
#pragma unroll
        for (int kc = 0; kc < KC; ++kc)
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const float p = __builtin_amdgcn_exp2f((s[kc][e] - mn) * c1);
                s[kc][e] = p;
                l += p;
            }
#pragma unroll
        for (int kc = 0; kc < KC; ++kc) {
            if (!EXACT && c0 + kc >= nkb) continue;     // (its probabilities are all zero)
#pragma unroll
            for (int s2 = 0; s2 < 2; ++s2) {
                // B operand: element j of lane half hh is P^T[key = kb*32 + 16*s2 + 8*(j>>2) + 4*hh + (j&3)][q]
                f16x8 ph, pl;
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    const float pv = s[kc][8 * s2 + j];
                    const f16_t hv = (f16_t)pv;
                    ph[j] = hv;
                    pl[j] = (f16_t)(pv - (float)hv);
                }
#pragma unroll
                for (int db = 0; db < 2; ++db) {
                    // A operand (see vit_attention_mfma): this lane ADDRESSES row key0 (+8), dims dcol..dcol+3 and RECEIVES
                    // column (lane & 15) of the 4 rows of its 16-lane group; once from the hi chunk, once from the lo chunk
                    const int dcol = db * 32 + (tg & 1) * 16 + tp * 4;
                    const int key0 = kc * 32 + 16 * s2 + 4 * (tg >> 1) + tq;
                    const int ch = 2 * (dcol >> 3), sub = (dcol & 7) * 2;
                    const s16x4 h0 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(Vs + voff(key0, ch) + sub));
                    const s16x4 h1 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(Vs + voff(key0 + 8, ch) + sub));
                    const s16x4 l0 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(Vs + voff(key0, ch + 1) + sub));
                    const s16x4 l1 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(Vs + voff(key0 + 8, ch + 1) + sub));
                    const f16x8 vh = __builtin_bit_cast(f16x8, __builtin_shufflevector(h0, h1, 0, 1, 2, 3, 4, 5, 6, 7));
                    const f16x8 vl = __builtin_bit_cast(f16x8, __builtin_shufflevector(l0, l1, 0, 1, 2, 3, 4, 5, 6, 7));
                    o[db] = __builtin_amdgcn_mfma_f32_32x32x16_f16(vh, pl, o[db], 0, 0, 0);
                    o[db] = __builtin_amdgcn_mfma_f32_32x32x16_f16(vl, ph, o[db], 0, 0, 0);
                    o[db] = __builtin_amdgcn_mfma_f32_32x32x16_f16(vh, ph, o[db], 0, 0, 0);
                }
                __builtin_amdgcn_sched_barrier(0);
            }
        }
    }
    l += __shfl_xor(l, 32, 64);
    if (live && q < N) {
        const float inv = 1.0f / l;
        g8_t* op = ctx + ((size_t)b * N + q) * D;
#pragma unroll
        for (int db = 0; db < 2; ++db)
#pragma unroll
            for (int g = 0; g < 4; ++g)
                store4(op, h * 64 + db * 32 + 8 * g + 4 * hh,
                       make_float4(o[db][4 * g] * inv, o[db][4 * g + 1] * inv, o[db][4 * g + 2] * inv, o[db][4 * g + 3] * inv));
    }
}

// ------------------------------------------------------------------------------------------------
// The single-pass case of vit_attention_split (ONE && EXACT: 193..224 tokens = KC full key blocks, ViT-B/16 at 224: 197) as a
// PERSISTENT kernel: one workgroup per CU walks the (image, head) units u = blockIdx.x, + gridDim.x, ... and the loads of the next
// unit run under the arithmetic of the current one.  K is needed by the score products only and V by the context products only,
// so the two 56 KiB buffers are refilled at different times:
//   barrier A (K_u, q_u have landed; every wave is done with V_{u-1}):  issue the DMA of V_u          -> S^T = K.Q^T
//   barrier B (V_u has landed; every wave is done with K_u):            issue the DMA of K_{u+1}, load q_{u+1} -> softmax, O^T = V^T.P^T, store
// In vit_attention_split every workgroup starts with its 112 KiB of DMA and a barrier in front of the first MFMA (~3 us of an
// ~18 us unit at one workgroup per CU).  The per-unit arithmetic is that kernel's, statement for statement: the context has the
// same bits (tests/test_split_gpu.py::test_vit_attention_persistent_matches_per_unit_kernel).  Measured, interleaved in one process
// (tools/bench_vit_attention.py, 256 x 12 units): 180 us per launch against 214 for the per-unit kernel; 1024 x 12: 711 / 799.
// Tried on this structure and not kept: the probabilities of key half-block t + 1 (exp, denominator, hi / lo split) formed in the
// scheduling region of half-block t's context products, i.e. one wave's softmax VALU work under its own MFMAs - the same bits; first
// with the compiler's addresses (every one of the lane's ~170 LDS fragment addresses its own register, hoisted out of the unit loop:
// 276-444 bytes of scratch per lane, 260 us per launch), then with the explicit lane bases below (239 registers, no scratch): 184 us
// against 180 for this form - level.  And, to run the two waves of a SIMD out of phase: a workgroup per (image, head, HALF of the query
// tiles) - four waves, one 56 KiB buffer that holds K and then V (the V DMA under the softmax), two workgroups per CU (162
// registers) - same bits, 190 us.  The three forms land within 15 % of each other because the launch moves 620 MB (q | k | v in,
// context out) - 3.4 TB/s at 180 us: it is more than half way to the HBM rate, not an MFMA- or VALU-bound kernel any more.
template <int I, int N, typename F> __device__ __forceinline__ void att_static_for(F&& f) {
    if constexpr (I < N) {
        f(std::integral_constant<int, I>{});
        att_static_for<I + 1, N>(f);
    }
}
template <int OFF> __device__ __forceinline__ s16x4 ds_read_tr16(unsigned addr) {
    s16x4 d;
    asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(d) : "v"(addr), "n"(OFF));
    return d;
}

template <int KC>
__global__ __launch_bounds__(512, 1) void vit_attention_split_pw(const g8_t* __restrict__ qkv, g8_t* __restrict__ ctx, int N, int H, int n_units) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int NP = KC * 32;
    char* Ks = smem;                                   // [NP] rows of 256 B
    char* Vs = smem + NP * 256;
    const int D = H * 64, ld = 3 * D;
    const int tid = threadIdx.x, lane = tid & 63, r32 = lane & 31, hh = lane >> 5;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const size_t rowb = (size_t)ld * 4;
    const int tq = (lane >> 2) & 3, tp = lane & 3, tg = lane >> 4;
    const float c1 = 0.125f * LOG2E;
    const bool live = wave < KC;                       // query tile `wave` of the unit (nqt == KC); the eighth wave only moves data
    const int q = wave * 32 + r32, qc = min(q, N - 1);

    auto unit_base = [&](int u) { return (const char*)(qkv + (size_t)(u / H) * N * ld + (u % H) * 64); };
    auto issue_k = [&](const char* base) {
        for (int p = wave; p < NP / 4; p += 8) {
            const int row = p * 4 + (lane >> 4);
            const int c = (lane & 15) ^ (row & 15);
            const char* src = base + (size_t)min(row, N - 1) * rowb;
            __builtin_amdgcn_global_load_lds(CAP_GPTR(src + (size_t)D * 4 + c * 16), CAP_LPTR(Ks + p * 1024), 16, 0, 0);
        }
    };
    auto issue_v = [&](const char* base) {
        for (int p = wave; p < NP / 4; p += 8) {
            const int row = p * 4 + (lane >> 4);
            const int cv = (lane & 15) ^ ((row & 1) | ((row & 2) << 2));
            const char* src = base + (size_t)min(row, N - 1) * rowb;
            __builtin_amdgcn_global_load_lds(CAP_GPTR(src + (size_t)2 * D * 4 + cv * 16), CAP_LPTR(Vs + p * 1024), 16, 0, 0);
        }
    };
    f16x8 qh[4], ql[4];
    auto load_q = [&](const char* base) {
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
            const char* qp = base + (size_t)qc * rowb + (2 * ks + hh) * 32;
            qh[ks] = *(const f16x8*)qp;
            ql[ks] = *(const f16x8*)(qp + 16);
        }
    };

    // LDS fragment addresses as a lane base + a compile-time offset.  K: block kc, k-step ks, half hl is row kc * 32 + r32, chunk
    // 2 (2 ks + hh) + hl, swizzled by the row's low four bits = r32's: 8 bases, + kc * 8192.  V (ds_read_b64_tr_b16): row key0 = kc * 32 +
    // 16 s2 + 4 (tg >> 1) + tq (+ 8), chunk ch = 2 (dcol >> 3) + hl swizzled by the row's low two bits = tq's: 4 bases, + (kc * 32 + 16 s2
    // [+ 8]) * 256.  Written out because the compiler does not see through the XOR: left to it, every one of the ~170 addresses is its
    // own register, hoisted out of the unit loop.
    unsigned kb[4][2], vb[2][2];
#pragma unroll
    for (int ks = 0; ks < 4; ++ks)
#pragma unroll
        for (int hl = 0; hl < 2; ++hl) kb[ks][hl] = (unsigned)(r32 * 256 + (((2 * (2 * ks + hh) + hl) ^ (r32 & 15)) << 4));
    {
        const int keyv = (tq & 1) | ((tq & 2) << 2);
#pragma unroll
        for (int db = 0; db < 2; ++db)
#pragma unroll
            for (int hl = 0; hl < 2; ++hl) {
                const int dcol = db * 32 + (tg & 1) * 16 + tp * 4;
                vb[db][hl] = (unsigned)((4 * (tg >> 1) + tq) * 256 + (((2 * (dcol >> 3) + hl) ^ keyv) << 4) + (dcol & 7) * 2);
            }
    }

    int u = blockIdx.x;
    if (u >= n_units) return;
    const char* base = unit_base(u);
    issue_k(base);
    load_q(base);
    for (;;) {
        __syncthreads();                               // A
        issue_v(base);
        f32x16 s[KC];
        float cm = -INFINITY;
        if (live) {
#pragma unroll
            for (int kc = 0; kc < KC; ++kc) {
#pragma unroll
                for (int e = 0; e < 16; ++e) s[kc][e] = 0.f;
#pragma unroll
                for (int ks = 0; ks < 4; ++ks) {
                    const f16x8 kh = *(const f16x8*)(Ks + kb[ks][0] + kc * 8192);
                    const f16x8 kl = *(const f16x8*)(Ks + kb[ks][1] + kc * 8192);
                    s[kc] = __builtin_amdgcn_mfma_f32_32x32x16_f16(kh, ql[ks], s[kc], 0, 0, 0);
                    s[kc] = __builtin_amdgcn_mfma_f32_32x32x16_f16(kl, qh[ks], s[kc], 0, 0, 0);
                    s[kc] = __builtin_amdgcn_mfma_f32_32x32x16_f16(kh, qh[ks], s[kc], 0, 0, 0);
                }
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        __syncthreads();                               // B
        const int un = u + (int)gridDim.x;
        const char* nbase = base;
        if (un < n_units) {
            nbase = unit_base(un);
            issue_k(nbase);
            load_q(nbase);
        }
        if (live) {
            // scores for query column r32: key(kc, e) = kc*32 + (e&3) + 8*(e>>2) + 4*hh
#pragma unroll
            for (int kc = 0; kc < KC; ++kc)
#pragma unroll
                for (int e = 0; e < 16; ++e) {
                    if (kc == KC - 1) {
                        const int key = kc * 32 + (e & 3) + 8 * (e >> 2) + 4 * hh;
                        if (key >= N) s[kc][e] = -INFINITY;     // padding keys of the last block
                    }
                    cm = fmaxf(cm, s[kc][e]);
                }
            cm = fmaxf(cm, __shfl_xor(cm, 32, 64));
            const float mn = cm;
            float l = 0.f;
            f32x16 o[2];
#pragma unroll
            for (int db = 0; db < 2; ++db)
#pragma unroll
                for (int e = 0; e < 16; ++e) o[db][e] = 0.f;
#pragma unroll
            for (int kc = 0; kc < KC; ++kc)
#pragma unroll
                for (int e = 0; e < 16; ++e) {
                    const float pe = __builtin_amdgcn_exp2f((s[kc][e] - mn) * c1);
                    s[kc][e] = pe;
                    l += pe;
                }
            // The V fragment reads are asm statements: the compiler cannot tell an LDS read from Vs from the LDS-DMA into Ks that is
            // in flight (the next unit's K, issued at barrier B) and waits vmcnt(0) in front of the first ds_read_b64_tr_b16 - which
            // puts the whole prefetch, q rows included, in front of the context products instead of under them.  V itself was
            // confirmed (vmcnt(0)) before barrier B.
            const unsigned vaddr[2][2] = {{(unsigned)(size_t)CAP_LPTR(Vs) + vb[0][0], (unsigned)(size_t)CAP_LPTR(Vs) + vb[0][1]},
                                          {(unsigned)(size_t)CAP_LPTR(Vs) + vb[1][0], (unsigned)(size_t)CAP_LPTR(Vs) + vb[1][1]}};
            att_static_for<0, 2 * KC>([&](auto tc) {
                constexpr int t = decltype(tc)::value, kc = t >> 1, s2 = t & 1, ro = (kc * 32 + 16 * s2) * 256;
                // B operand: element j of lane half hh is P^T[key = kc*32 + 16*s2 + 8*(j>>2) + 4*hh + (j&3)][q]
                f16x8 ph, pl;
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    const float pv = s[kc][8 * s2 + j];
                    const f16_t hv = (f16_t)pv;
                    ph[j] = hv;
                    pl[j] = (f16_t)(pv - (float)hv);
                }
                // A operand (see vit_attention_mfma): this lane ADDRESSES row key0 (+8), dims dcol..dcol+3 and RECEIVES column
                // (lane & 15) of the 4 rows of its 16-lane group; once from the hi chunk, once from the lo chunk
                s16x4 h0[2], h1[2], l0[2], l1[2];
#pragma unroll
                for (int db = 0; db < 2; ++db) {
                    h0[db] = ds_read_tr16<ro>(vaddr[db][0]);
                    h1[db] = ds_read_tr16<ro + 2048>(vaddr[db][0]);
                    l0[db] = ds_read_tr16<ro>(vaddr[db][1]);
                    l1[db] = ds_read_tr16<ro + 2048>(vaddr[db][1]);
                }
                asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(h0[0]), "+v"(h1[0]), "+v"(l0[0]), "+v"(l1[0]), "+v"(h0[1]), "+v"(h1[1]), "+v"(l0[1]), "+v"(l1[1]));
#pragma unroll
                for (int db = 0; db < 2; ++db) {
                    const f16x8 vh = __builtin_bit_cast(f16x8, __builtin_shufflevector(h0[db], h1[db], 0, 1, 2, 3, 4, 5, 6, 7));
                    const f16x8 vl = __builtin_bit_cast(f16x8, __builtin_shufflevector(l0[db], l1[db], 0, 1, 2, 3, 4, 5, 6, 7));
                    o[db] = __builtin_amdgcn_mfma_f32_32x32x16_f16(vh, pl, o[db], 0, 0, 0);
                    o[db] = __builtin_amdgcn_mfma_f32_32x32x16_f16(vl, ph, o[db], 0, 0, 0);
                    o[db] = __builtin_amdgcn_mfma_f32_32x32x16_f16(vh, ph, o[db], 0, 0, 0);
                }
                __builtin_amdgcn_sched_barrier(0);
            });
            l += __shfl_xor(l, 32, 64);
            if (q < N) {
                const float inv = 1.0f / l;
                g8_t* op = ctx + ((size_t)(u / H) * N + q) * D;
                const int h = u % H;
#pragma unroll
                for (int db = 0; db < 2; ++db)
#pragma unroll
                    for (int g = 0; g < 4; ++g)
                        store4(op, h * 64 + db * 32 + 8 * g + 4 * hh,
                               make_float4(o[db][4 * g] * inv, o[db][4 * g + 1] * inv, o[db][4 * g + 2] * inv, o[db][4 * g + 3] * inv));
            }
        }
        if (un >= n_units) break;
        u = un;
        base = nbase;
    }
}

// ------------------------------------------------------------------------------------------------

template <typename T>
__global__ __launch_bounds__(256) void decode_attention_kernel(const T* __restrict__ q, const T* __restrict__ kbase,
                                                               const T* __restrict__ vbase, const int* __restrict__ anc,
                                                               int anc_ld, int rows_per_kv, int kv_ld, int n_keys,
                                                               T* __restrict__ out, int H) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float* sc = (float*)smem;                         // [n_keys] scores -> probabilities
    float* red = sc + ((n_keys + 3) & ~3);            // [8] block reductions
    float* part = red + 8;                            // [4][64] per-wave partial outputs
    const int row = blockIdx.x, h = blockIdx.y, Dh = H * 64;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, ksub = lane >> 3, dch = lane & 7;
    float qv[8];
    load8<T>(q + (size_t)row * Dh + h * 64 + dch * 8, qv);
#pragma unroll
    for (int i = 0; i < 8; ++i) qv[i] *= 0.125f;
    const int ngroups = (n_keys + 7) / 8;
    for (int g = wave; g < ngroups; g += 4) {
        const int key = g * 8 + ksub;
        float s = 0.f;
        if (key < n_keys) {
            const int src = anc ? anc[(size_t)row * anc_ld + key] : row / rows_per_kv;
            float kv[8];
            load8<T>(kbase + (((size_t)src * H + h) * kv_ld + key) * 64 + dch * 8, kv);
#pragma unroll
            for (int i = 0; i < 8; ++i) s = fmaf(qv[i], kv[i], s);
        }
        s += __shfl_xor(s, 1, 64); s += __shfl_xor(s, 2, 64); s += __shfl_xor(s, 4, 64);
        if (dch == 0 && key < n_keys) sc[key] = s;
    }
    __syncthreads();
    float m = -INFINITY;
    for (int j = tid; j < n_keys; j += 256) m = fmaxf(m, sc[j]);
    m = wave_max(m);
    if (lane == 0) red[wave] = m;
    __syncthreads();
    m = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
    float l = 0.f;
    for (int j = tid; j < n_keys; j += 256) { float p = expf(sc[j] - m); sc[j] = p; l += p; }
    l = wave_sum(l);
    if (lane == 0) red[4 + wave] = l;
    __syncthreads();
    l = red[4] + red[5] + red[6] + red[7];
    float o[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) o[i] = 0.f;
    for (int g = wave; g < ngroups; g += 4) {
        const int key = g * 8 + ksub;
        if (key < n_keys) {
            const int src = anc ? anc[(size_t)row * anc_ld + key] : row / rows_per_kv;
            float vv[8];
            load8<T>(vbase + (((size_t)src * H + h) * kv_ld + key) * 64 + dch * 8, vv);
            const float p = sc[key];
#pragma unroll
            for (int i = 0; i < 8; ++i) o[i] = fmaf(p, vv[i], o[i]);
        }
    }
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        o[i] += __shfl_xor(o[i], 8, 64); o[i] += __shfl_xor(o[i], 16, 64); o[i] += __shfl_xor(o[i], 32, 64);
    }
    if (ksub == 0) {
#pragma unroll
        for (int i = 0; i < 8; ++i) part[wave * 64 + dch * 8 + i] = o[i];
    }
    __syncthreads();
    if (tid < 64) {
        const float v = (part[tid] + part[64 + tid] + part[128 + tid] + part[192 + tid]) / l;
        out[(size_t)row * Dh + h * 64 + tid] = from_f32<T>(v);
    }
}




// ---- decode attention, short history (n_keys <= 8*NI): one wave per (row, head): decode_attn.h's wave unit.
template <typename T, int NI, typename TO = T>
__global__ __launch_bounds__(256, 3) void decode_attention_wave_kernel(const T* __restrict__ q, T* __restrict__ kbase,
                                                                    T* __restrict__ vbase,
                                                                    const int* __restrict__ anc, int anc_ld,
                                                                    int rows_per_kv, int kv_ld, int n_keys,
                                                                    TO* __restrict__ out, int R, int H, QSource qs,
                                                                    const int* __restrict__ skip, RowMap map) {
    const int unit = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (unit >= R * H) return;
    const int crow = unit / H, h = unit - crow * H;      // row of q / the partial sums / out; the caches belong to `row`
    if (map.n && crow >= *map.n) return;
    const int row = map.live ? map.live[crow] : crow;
    if (skip && skip[row]) return;          // caption already ended: its logits are never looked at again
    decode_attention_wave_unit<T, NI, TO>(q, kbase, vbase, anc, anc_ld, rows_per_kv, kv_ld, n_keys, out + (size_t)crow * H * 64, R, H,
                                          qs, row, h, threadIdx.x & 63, true, nullptr, crow);
}


// ---- decode attention, long history (cross-attention over the image tokens): one wave per (row, head), the history
// walked in chunks (decode_attn.h's online unit).  With ~4 waves per SIMD that keeps >100 KB in flight per CU, which is what
// streaming the beam-shared K/V cache at HBM rate needs.
template <typename T, int G, bool DB, bool NT = false, typename TO = T, typename TKV = T>      // TKV: element type of the K/V cache
__global__ __launch_bounds__(256, 3) void decode_attention_online_kernel(const T* __restrict__ q, const T* __restrict__ kbase,
                                                                      const T* __restrict__ vbase,
                                                                      const int* __restrict__ anc, int anc_ld,
                                                                      int rows_per_kv, int kv_ld, int n_keys,
                                                                      TO* __restrict__ out, int R, int H, QSource qs,
                                                                      const int* __restrict__ skip, size_t ri0 = 0, RowMap map = RowMap()) {
    // ri0: row index of the first K/V row the launch may touch, for caches addressed by row index (KV16: kbase / vbase are the
    // bases of the layer's whole k / v block).
    // G = key groups (of 8 keys) per chunk, chosen by the launcher so the chunks are balanced (197 keys -> 4 x 56).
    const int unit = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (unit >= R * H) return;
    const int crow = unit / H, h = unit - crow * H;      // row of q / the partial sums / out; the K/V block belongs to `row`
    if (map.n && crow >= *map.n) return;
    const int row = map.live ? map.live[crow] : crow;
    // a caption that has ended keeps its row, but nothing reads the row's logits any more (the selection kernels emit pad):
    // its 100-200 KB of K/V per layer are not streamed.  On the bench workload the mean caption is ~11 of 19 steps long.
    if (skip && skip[row]) return;
    decode_attention_online_unit<T, G, DB, NT, TO, TKV>(q, kbase, vbase, anc, anc_ld, kv_ld, n_keys, out + (size_t)crow * H * 64, R, H, qs,
                                                        row, h, threadIdx.x & 63, ri0,
                                                        ri0 + ((size_t)(row / rows_per_kv) * H + h) * kv_ld, nullptr, nullptr, crow);
}


// ---- decode attention over a K/V block SHARED by the NB beams of an image (cross-attention under beam search, no ancestry):
// one wave per (image, head) streams the block once and serves all NB query rows from the same registers - with one wave per
// (row, head) the NB rows each pull the block through L2 (CoCa 336, 5 beams: 500 MB of L2 traffic per launch, 81 us).  Chunking
// (G) and the per-row operation order are those of decode_attention_online_kernel: a row's result has the same bits.
template <typename T, int G, int NB, typename TO = T, typename TKV = T>
__global__ __launch_bounds__(256, 2) void decode_attention_shared_kernel(const T* __restrict__ q, const T* __restrict__ kbase,
                                                                      const T* __restrict__ vbase, int kv_ld, int n_keys,
                                                                      TO* __restrict__ out, int n_img, int H, QSource qs,
                                                                      const int* __restrict__ skip, size_t ri0 = 0) {
#pragma clang fp contract(off)      // the online unit's arithmetic exactly as written there (decode_attn.h): same bits per row
    constexpr int CH = 8 * G;
    const int unit = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (unit >= n_img * H) return;
    const int img = unit / H, h = unit - img * H, Dh = H * 64, R = n_img * NB;
    const int lane = threadIdx.x & 63, ksub = lane >> 3, dch = lane & 7;
    bool live[NB], any = false;
#pragma unroll
    for (int b = 0; b < NB; ++b) { live[b] = !(skip && skip[img * NB + b]); any |= live[b]; }
    if (!any) return;
    float m[NB], l[NB], o[NB][8], qv[NB][8];
#pragma unroll
    for (int b = 0; b < NB; ++b) {
        m[b] = -INFINITY; l[b] = 0.f;
#pragma unroll
        for (int e = 0; e < 8; ++e) o[b][e] = 0.f;
    }
    Raw8<TKV> kr[G], vr[G];
    auto issue = [&](int k0) {
#pragma unroll
        for (int g = 0; g < G; ++g) {
            const int key = k0 + g * 8 + ksub;
            if (key < n_keys) {
                const size_t ri = ri0 + ((size_t)img * H + h) * kv_ld + key;
                kr[g].load_row(kbase, ri, dch); vr[g].load_row(vbase, ri, dch);
            } else {
                kr[g].zero(); vr[g].zero();
            }
        }
    };
    issue(0);
#pragma unroll
    for (int b = 0; b < NB; ++b) {
        const int row = img * NB + b;
        if (qs.part) {
            Part8 pq;
            pq.issue(qs, R, row, qs.col0 + h * 64 + dch * 8);
            pq.finish<T>(qs, qv[b]);
        } else {
            load8<T>(q + (size_t)row * Dh + h * 64 + dch * 8, qv[b]);
        }
#pragma unroll
        for (int e = 0; e < 8; ++e) qv[b][e] *= 0.125f;
    }
    for (int k0 = 0;;) {
#pragma unroll
        for (int b = 0; b < NB; ++b) {
            float sc[G], cm = -INFINITY;
#pragma unroll
            for (int g = 0; g < G; ++g) {
                const float sv = decode_key_score<TKV>(qv[b], kr[g]);
                sc[g] = (k0 + g * 8 + ksub < n_keys) ? sv : -INFINITY;
                cm = fmaxf(cm, sc[g]);
            }
            cm = fmaxf(cm, __shfl_xor(cm, 8, 64)); cm = fmaxf(cm, __shfl_xor(cm, 16, 64)); cm = fmaxf(cm, __shfl_xor(cm, 32, 64));
            const float mn = fmaxf(m[b], cm);
            const float c = expf(m[b] - mn);
            l[b] *= c;
#pragma unroll
            for (int e = 0; e < 8; ++e) o[b][e] *= c;
#pragma unroll
            for (int g = 0; g < G; ++g) {
                const float pj = expf(sc[g] - mn);
                l[b] += pj;
                const float pw = Raw8<TKV>::scaled ? pj * vr[g].scale() : pj;
#pragma unroll
                for (int e = 0; e < 8; ++e) o[b][e] = fmaf(pw, vr[g].get(e), o[b][e]);
            }
            m[b] = mn;
        }
        k0 += CH;
        if (k0 >= n_keys) break;
        issue(k0);
    }
#pragma unroll
    for (int b = 0; b < NB; ++b) {
        float lb = l[b];
        lb += __shfl_xor(lb, 8, 64); lb += __shfl_xor(lb, 16, 64); lb += __shfl_xor(lb, 32, 64);
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            o[b][e] += __shfl_xor(o[b][e], 8, 64); o[b][e] += __shfl_xor(o[b][e], 16, 64); o[b][e] += __shfl_xor(o[b][e], 32, 64);
        }
        if (ksub == 0 && live[b]) {
            const float inv = 1.0f / lb;
            TO* op = out + (size_t)(img * NB + b) * Dh;
            store4(op, h * 64 + dch * 8, make_float4(o[b][0] * inv, o[b][1] * inv, o[b][2] * inv, o[b][3] * inv));
            store4(op, h * 64 + dch * 8 + 4, make_float4(o[b][4] * inv, o[b][5] * inv, o[b][6] * inv, o[b][7] * inv));
        }
    }
}

// ---- attentional pooler: Q learned queries (already layer-normed and projected on the host, identical for every
// image) attend over the N image tokens.  One thread per query, K/V tiles broadcast from LDS, online softmax; head_dim is
// a template parameter (CoCa ViT-L/14: 768 / 8 heads = 96).  ~1 % of the CoCa encoder's flops.
template <typename T, int HD, typename TO = T>
__global__ __launch_bounds__(64) void pool_attention_kernel(const float* __restrict__ qp, const T* __restrict__ kv,
                                                            TO* __restrict__ out, int N, int Q, int E, int heads) {
    constexpr int KT = 32;
    __shared__ float Ks[KT][HD + 1];
    __shared__ float Vs[KT][HD + 1];
    const int b = blockIdx.x / heads, h = blockIdx.x % heads;
    const int tid = threadIdx.x, q = blockIdx.y * 64 + tid, qc = min(q, Q - 1);
    const float scale = rsqrtf((float)HD);
    float qv[HD], o[HD];
#pragma unroll
    for (int d = 0; d < HD; ++d) { qv[d] = qp[(size_t)qc * E + h * HD + d] * scale; o[d] = 0.f; }
    float m = -INFINITY, l = 0.f;
    const T* base = kv + (size_t)b * N * 2 * E + h * HD;
    for (int k0 = 0; k0 < N; k0 += KT) {
        const int nk = min(KT, N - k0);
        __syncthreads();
        for (int c = tid; c < nk * HD; c += 64) {
            const int j = c / HD, d = c - j * HD;
            Ks[j][d] = to_f32(base[(size_t)(k0 + j) * 2 * E + d]);
            Vs[j][d] = to_f32(base[(size_t)(k0 + j) * 2 * E + E + d]);
        }
        __syncthreads();
        for (int j = 0; j < nk; ++j) {
            float sc = 0.f;
#pragma unroll
            for (int d = 0; d < HD; ++d) sc = fmaf(qv[d], Ks[j][d], sc);
            const float mn = fmaxf(m, sc);
            const float c = expf(m - mn), pj = expf(sc - mn);
            l = l * c + pj;
#pragma unroll
            for (int d = 0; d < HD; ++d) o[d] = fmaf(pj, Vs[j][d], o[d] * c);
            m = mn;
        }
    }
    if (q < Q) {
        const float inv = 1.0f / l;
        TO* orow = out + ((size_t)b * Q + q) * E;
#pragma unroll
        for (int d = 0; d < HD; ++d) store1(orow, h * HD + d, o[d] * inv);
    }
}

// ---- sentence-encoder self-attention (bidirectional, key-padding mask from the sentence length): a caption is <= a few
// dozen tokens, so one wave owns one (sentence, head): K and V rows sit in LDS as fp32, lane = query, online softmax over
// the valid keys.  <0.5 % of the encoder's flops; head_dim is a template parameter (MiniLM: 384 / 12 = 32).
template <typename T, int HD>
__global__ __launch_bounds__(64) void text_attention_kernel(const T* __restrict__ qkv, const int* __restrict__ lens,
                                                            T* __restrict__ ctx, int L, int H) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float* Ks = (float*)smem;
    float* Vs = Ks + (size_t)L * HD;
    const int b = blockIdx.x / H, h = blockIdx.x % H, lane = threadIdx.x;
    const int D = H * HD, ld = 3 * D;
    const int n = min(max(lens[b], 1), L);
    const T* base = qkv + (size_t)b * L * ld + h * HD;
    for (int i = lane; i < n * HD; i += 64) {
        const int j = i / HD, d = i - j * HD;
        Ks[i] = to_f32(base[(size_t)j * ld + D + d]);
        Vs[i] = to_f32(base[(size_t)j * ld + 2 * D + d]);
    }
    __syncthreads();
    const float scale = HD == 32 ? 0.17677669529663687f : 0.125f;
    for (int q0 = 0; q0 < L; q0 += 64) {
        const int q = q0 + lane;
        if (q >= L) break;
        float qv[HD], o[HD];
#pragma unroll
        for (int d = 0; d < HD; ++d) { qv[d] = to_f32(base[(size_t)q * ld + d]) * scale; o[d] = 0.f; }
        float m = -INFINITY, l = 0.f;
        for (int j = 0; j < n; ++j) {
            float sc = 0.f;
#pragma unroll
            for (int d = 0; d < HD; ++d) sc = fmaf(qv[d], Ks[j * HD + d], sc);
            const float mn = fmaxf(m, sc);
            const float c = expf(m - mn), pj = expf(sc - mn);
            l = l * c + pj;
#pragma unroll
            for (int d = 0; d < HD; ++d) o[d] = fmaf(pj, Vs[j * HD + d], o[d] * c);
            m = mn;
        }
        const float inv = 1.0f / l;
        T* op = ctx + ((size_t)b * L + q) * D + h * HD;
#pragma unroll
        for (int d = 0; d < HD; ++d) op[d] = from_f32<T>(o[d] * inv);
    }
}

// ---- generic attention for the BLIP-2 path (head_dim 88 in ViT-g, 80 in OPT-2.7b, 64 in the Q-Former; query / key / value
// live in differently strided buffers: fused qkv rows, a per-layer K/V cache, image tokens).  One wave per (batch, head, block
// of 64 queries): lane = query, its q and output rows in registers (HDP = head_dim padded to 32/64/96/128), keys staged 32 at
// a time into LDS as fp32, online softmax.  causal_off >= 0: query i sees keys j <= i + causal_off.  A correctness-first
// kernel (VALU dot products); the hot ViT-B/L paths keep their MFMA kernels.
template <typename T, int HDP, typename TO = T>
__global__ __launch_bounds__(64) void generic_attention_kernel(const T* __restrict__ q, long ldq, long qbs,
                                                               const T* __restrict__ k, long ldk, long kbs,
                                                               const T* __restrict__ v, long ldv, long vbs,
                                                               TO* __restrict__ out, long ldo, long obs, int Lq, int Lk,
                                                               int H, int hd, int causal_off, float scale) {
    __shared__ float Ks[32 * HDP], Vs[32 * HDP];
    const int nqb = (Lq + 63) / 64;
    const int qb = blockIdx.x % nqb, bh = blockIdx.x / nqb, h = bh % H, b = bh / H;
    const int lane = threadIdx.x, qi = qb * 64 + lane;
    const bool live = qi < Lq;
    float qv[HDP], o[HDP];
#pragma unroll
    for (int d = 0; d < HDP; ++d) {
        qv[d] = (live && d < hd) ? to_f32(q[(size_t)b * qbs + (size_t)qi * ldq + h * hd + d]) * scale : 0.f;
        o[d] = 0.f;
    }
    float m = -INFINITY, l = 0.f;
    const int kmax = causal_off >= 0 ? min(Lk, qb * 64 + 63 + causal_off + 1) : Lk;      // last key any query of this block sees
    for (int k0 = 0; k0 < kmax; k0 += 32) {
        __syncthreads();
        for (int i = lane; i < 32 * HDP; i += 64) {
            const int j = i / HDP, d = i - j * HDP;
            const bool ok = k0 + j < Lk && d < hd;
            Ks[i] = ok ? to_f32(k[(size_t)b * kbs + (size_t)(k0 + j) * ldk + h * hd + d]) : 0.f;
            Vs[i] = ok ? to_f32(v[(size_t)b * vbs + (size_t)(k0 + j) * ldv + h * hd + d]) : 0.f;
        }
        __syncthreads();
        const int jn = min(32, Lk - k0);
        for (int j = 0; j < jn; ++j) {
            if (causal_off >= 0 && k0 + j > qi + causal_off) break;
            float sc = 0.f;
#pragma unroll
            for (int d = 0; d < HDP; ++d) sc = fmaf(qv[d], Ks[j * HDP + d], sc);
            const float mn = fmaxf(m, sc);
            const float c = expf(m - mn), pj = expf(sc - mn);
            l = l * c + pj;
#pragma unroll
            for (int d = 0; d < HDP; ++d) o[d] = fmaf(pj, Vs[j * HDP + d], o[d] * c);
            m = mn;
        }
    }
    if (live) {
        const float inv = 1.0f / l;
        TO* orow = out + (size_t)b * obs + (size_t)qi * ldo;          // row base: a multiple of 8 elements for a G8 output
#pragma unroll
        for (int d = 0; d < HDP; ++d)
            if (d < hd) store1(orow, h * hd + d, o[d] * inv);
    }
}

// The same attention for ONE block of up to 64 queries without a mask (the Q-Former: 32 queries against themselves and against the
// 257 image tokens), keys split over the four waves of a 256-thread workgroup: generic_attention_kernel walks all keys with one
// wave per query block - 260 us per launch at 257 keys whatever the batch (a launch has B x heads such workgroups and each is one
// wave's serial loop; BLIP-2, one crop: 1.6 ms of Q-Former cross-attention per caption).  Each wave runs the online softmax over its
// quarter of the keys (staged wave-privately), then the four states (m, l, o) are merged: o = sum_w o_w e^(m_w - M) / sum_w l_w e^(m_w - M).
template <typename T, int HDP, typename TO = T>
__global__ __launch_bounds__(256) void generic_attention_kp_kernel(const T* __restrict__ q, long ldq, long qbs, const T* __restrict__ k, long ldk,
                                                                   long kbs, const T* __restrict__ v, long ldv, long vbs, TO* __restrict__ out,
                                                                   long ldo, long obs, int Lq, int Lk, int H, int hd, float scale) {
    extern __shared__ __attribute__((aligned(16))) float kp_sm[];
    const int bh = blockIdx.x, h = bh % H, b = bh / H;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, qi = lane;
    const bool live = qi < Lq;
    float* Ks = kp_sm + wave * (2 * 32 * HDP);
    float* Vs = Ks + 32 * HDP;
    float qv[HDP], o[HDP];
#pragma unroll
    for (int d = 0; d < HDP; ++d) {
        qv[d] = (live && d < hd) ? to_f32(q[(size_t)b * qbs + (size_t)qi * ldq + h * hd + d]) * scale : 0.f;
        o[d] = 0.f;
    }
    float m = -INFINITY, l = 0.f;
    const int per = (Lk + 3) / 4, j0 = wave * per, j1 = min(Lk, j0 + per);
    for (int c = 0; c < per; c += 32) {                  // the same trip count in every wave: the barriers are workgroup barriers
        const int k0 = j0 + c;
        __syncthreads();
        for (int i = lane; i < 32 * HDP; i += 64) {
            const int j = i / HDP, d = i - j * HDP;
            const bool ok = k0 + j < j1 && d < hd;
            Ks[i] = ok ? to_f32(k[(size_t)b * kbs + (size_t)(k0 + j) * ldk + h * hd + d]) : 0.f;
            Vs[i] = ok ? to_f32(v[(size_t)b * vbs + (size_t)(k0 + j) * ldv + h * hd + d]) : 0.f;
        }
        __syncthreads();
        const int jn = min(32, j1 - k0);
        for (int j = 0; j < jn; ++j) {
            float sc = 0.f;
#pragma unroll
            for (int d = 0; d < HDP; ++d) sc = fmaf(qv[d], Ks[j * HDP + d], sc);
            const float mn = fmaxf(m, sc);
            const float cc = expf(m - mn), pj = expf(sc - mn);
            l = l * cc + pj;
#pragma unroll
            for (int d = 0; d < HDP; ++d) o[d] = fmaf(pj, Vs[j * HDP + d], o[d] * cc);
            m = mn;
        }
    }
    __syncthreads();                                      // staging is done: the space holds the four states now
    float* st = kp_sm + wave * ((HDP + 2) * 64);          // [HDP + 2][64 lanes]
    st[lane] = m;
    st[64 + lane] = l;
#pragma unroll
    for (int d = 0; d < HDP; ++d) st[(2 + d) * 64 + lane] = o[d];
    __syncthreads();
    if (!live) return;
    float mw[4], M = -INFINITY;
#pragma unroll
    for (int w = 0; w < 4; ++w) { mw[w] = kp_sm[w * ((HDP + 2) * 64) + lane]; M = fmaxf(M, mw[w]); }
    float f[4], L = 0.f;
#pragma unroll
    for (int w = 0; w < 4; ++w) {
        f[w] = mw[w] == -INFINITY ? 0.f : expf(mw[w] - M);               // a wave without keys contributes nothing
        L += kp_sm[w * ((HDP + 2) * 64) + 64 + lane] * f[w];
    }
    const float inv = 1.0f / L;
    TO* orow = out + (size_t)b * obs + (size_t)qi * ldo;
    for (int d = wave * (HDP / 4); d < (wave + 1) * (HDP / 4); ++d) {    // each wave finishes a quarter of the head's dimensions
        if (d >= hd) break;
        float acc = 0.f;
#pragma unroll
        for (int w = 0; w < 4; ++w) acc += kp_sm[w * ((HDP + 2) * 64) + (2 + d) * 64 + lane] * f[w];
        store1(orow, h * hd + d, acc * inv);
    }
}

// One query per (batch, head) against a short history (OPT decode step): lanes = keys for the scores, lanes = head
// dimensions for the weighted sum; q and the probabilities pass through LDS.
template <typename T, typename TO = T>
__global__ __launch_bounds__(64) void generic_decode_attention_kernel(const T* __restrict__ q, long qbs, const T* __restrict__ k,
                                                                      long ldk, long kbs, const T* __restrict__ v, long ldv,
                                                                      long vbs, TO* __restrict__ out, long obs, int Lk, int H,
                                                                      int hd, float scale) {
    __shared__ float qs[128], ps[1024];
    const int b = blockIdx.x / H, h = blockIdx.x % H, lane = threadIdx.x;
    for (int d = lane; d < hd; d += 64) qs[d] = to_f32(q[(size_t)b * qbs + h * hd + d]) * scale;
    __syncthreads();
    float m = -INFINITY;
    for (int j = lane; j < Lk; j += 64) {
        const T* kr = k + (size_t)b * kbs + (size_t)j * ldk + h * hd;
        float sc = 0.f;
        for (int d = 0; d < hd; d += 8) {
            float kk[8];
            load8<T>(kr + d, kk);
#pragma unroll
            for (int e = 0; e < 8; ++e) sc = fmaf(qs[d + e], kk[e], sc);
        }
        ps[j] = sc;
        m = fmaxf(m, sc);
    }
    m = wave_max(m);
    float l = 0.f;
    for (int j = lane; j < Lk; j += 64) {
        const float pj = expf(ps[j] - m);
        ps[j] = pj;
        l += pj;
    }
    l = wave_sum(l);
    __syncthreads();
    const float inv = 1.0f / l;
    for (int d = lane; d < hd; d += 64) {
        float o = 0.f;
        for (int j = 0; j < Lk; ++j) o = fmaf(ps[j], to_f32(v[(size_t)b * vbs + (size_t)j * ldv + h * hd + d]), o);
        store1(out + (size_t)b * obs, h * hd + d, o * inv);
    }
}

// Cached decode step of a pre-LN decoder (OPT): one query per (batch, head) taken from the fused q|k|v row; the step's new
// key/value are appended to the caches [B][Lmax][T] at position `past` here (no separate append launch) and used from the
// row directly.  One wave per (batch, head): lanes over keys for the scores, then (key group of 4) x (8-dim chunk) for
// P.V with a 4-way sum through LDS.  head_dim a multiple of 8, <= 128; past + 1 <= 1024.
// HD8 = head_dim / 8 when it is known at compile time (10: OPT-2.7b, 8, 16), 0 = any: with a constant trip count a lane's
// 16-byte loads of a key row are all issued before the first FMA (a runtime loop serialises load -> fma, 10 round trips).
template <typename T, int HD8, typename TO = T>
__global__ __launch_bounds__(64) void opt_decode_attention_kernel(const T* __restrict__ qkv, T* __restrict__ kc, T* __restrict__ vc,
                                                                  TO* __restrict__ out, int Tw, int H, int hd, int Lmax, int past,
                                                                  float scale) {
    __shared__ float qs[128], ps[1024], os[4][128];
    const int b = blockIdx.x / H, h = blockIdx.x % H, lane = threadIdx.x;
    const T* row = qkv + (size_t)b * 3 * Tw + h * hd;
    const T* kb = kc + (size_t)b * Lmax * Tw + h * hd;
    const T* vb = vc + (size_t)b * Lmax * Tw + h * hd;
    const int Lk = past + 1;
    const int n8 = HD8 ? HD8 : hd >> 3;
    for (int d = lane; d < hd; d += 64) {
        qs[d] = to_f32(row[d]) * scale;
        kc[((size_t)b * Lmax + past) * Tw + h * hd + d] = row[Tw + d];
        vc[((size_t)b * Lmax + past) * Tw + h * hd + d] = row[2 * Tw + d];
    }
    __syncthreads();
    float m = -INFINITY;
    for (int j = lane; j < Lk; j += 64) {
        const T* kr = j == past ? row + Tw : kb + (size_t)j * Tw;
        float sc = 0.f;
        if constexpr (HD8 > 0) {
            float kk[HD8][8];
#pragma unroll
            for (int c = 0; c < HD8; ++c) load8<T>(kr + c * 8, kk[c]);
#pragma unroll
            for (int c = 0; c < HD8; ++c)
#pragma unroll
                for (int e = 0; e < 8; ++e) sc = fmaf(qs[c * 8 + e], kk[c][e], sc);
        } else {
            for (int c = 0; c < n8; ++c) {
                float kk[8];
                load8<T>(kr + c * 8, kk);
#pragma unroll
                for (int e = 0; e < 8; ++e) sc = fmaf(qs[c * 8 + e], kk[e], sc);
            }
        }
        ps[j] = sc;
        m = fmaxf(m, sc);
    }
    m = wave_max(m);
    float l = 0.f;
    for (int j = lane; j < Lk; j += 64) {
        const float pj = expf(ps[j] - m);
        ps[j] = pj;
        l += pj;
    }
    l = wave_sum(l);
    __syncthreads();
    const int jg = lane >> 4, dc = (lane & 15) * 8;
    float o[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) o[e] = 0.f;
    if (dc < hd) {
        int j = jg;
        for (; j + 12 < Lk; j += 16) {            // four keys of this lane's group per trip, loads first
            float vv[4][8];
#pragma unroll
            for (int u = 0; u < 4; ++u) load8<T>((j + 4 * u == past ? row + 2 * Tw : vb + (size_t)(j + 4 * u) * Tw) + dc, vv[u]);
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const float pj = ps[j + 4 * u];
#pragma unroll
                for (int e = 0; e < 8; ++e) o[e] = fmaf(pj, vv[u][e], o[e]);
            }
        }
        for (; j < Lk; j += 4) {
            float vv[8];
            load8<T>((j == past ? row + 2 * Tw : vb + (size_t)j * Tw) + dc, vv);
            const float pj = ps[j];
#pragma unroll
            for (int e = 0; e < 8; ++e) o[e] = fmaf(pj, vv[e], o[e]);
        }
    }
#pragma unroll
    for (int e = 0; e < 8; ++e) os[jg][dc + e] = o[e];
    __syncthreads();
    const float inv = 1.0f / l;
    for (int d = lane; d < hd; d += 64) store1(out + (size_t)b * Tw, h * hd + d, (os[0][d] + os[1][d] + os[2][d] + os[3][d]) * inv);
}

// OPT decoder inputs.  Prefill: row (b, j) of x[B, P, T] = (j < nq ? projected query (b, j) : token table[bos]) + position
// table[j + 2] (HF OPTLearnedPositionalEmbedding offset).  Decode: x[b] = token table[seq[b][cur]] + position table[cur + 2].
__global__ void opt_prefill_inputs_kernel(const float* __restrict__ proj, const float* __restrict__ tok, const float* __restrict__ pos,
                                          float* __restrict__ x, int B, int nq, int T, int bos) {
    const size_t n = (size_t)B * (nq + 1) * T;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        const int d = i % T;
        const size_t r = i / T;
        const int j = r % (nq + 1), b = r / (nq + 1);
        const float e = j < nq ? proj[((size_t)b * nq + j) * T + d] : tok[(size_t)bos * T + d];
        x[i] = e + pos[(size_t)(j + 2) * T + d];
    }
}
__global__ void opt_token_inputs_kernel(const int* __restrict__ seq, int seq_ld, int cur, const float* __restrict__ tok,
                                        const float* __restrict__ pos, float* __restrict__ x, int B, int T) {
    const size_t n = (size_t)B * T;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        const int d = i % T, b = i / T;
        x[i] = tok[(size_t)seq[(size_t)b * seq_ld + cur] * T + d] + pos[(size_t)(cur + 2) * T + d];
    }
}
// k|v columns of fused qkv rows [B * L, 3T] -> caches [B][Lmax][T] at positions pos0 .. pos0 + L - 1
template <typename T>
__global__ void kv_append_kernel(const T* __restrict__ qkv, T* __restrict__ kc, T* __restrict__ vc, int B, int L, int Tw,
                                 int Lmax, int pos0) {
    const size_t n = (size_t)B * L * Tw;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        const int d = i % Tw;
        const size_t r = i / Tw;
        const int j = r % L, b = r / L;
        const size_t dst = ((size_t)b * Lmax + pos0 + j) * Tw + d;
        kc[dst] = qkv[r * 3 * Tw + Tw + d];
        vc[dst] = qkv[r * 3 * Tw + 2 * Tw + d];
    }
}
// rows of a [n, D] table repeated for every batch item: dst_f / dst_t [B * n, D]
template <typename T>
__global__ void rows_broadcast_kernel(const float* __restrict__ src, float* __restrict__ dst_f, T* __restrict__ dst_t, int B, int n, int D) {
    const size_t tot = (size_t)B * n * D;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < tot; i += (size_t)gridDim.x * blockDim.x) {
        const float x = src[i % ((size_t)n * D)];
        dst_f[i] = x;
        store1(dst_t + (i - i % D), (int)(i % D), x);
    }
}

template <int KB>
int launch_mfma_kb(const void* qkv, void* ctx, int B, int N, int H, hipStream_t s) {
    const int lds = 2 * KB * 32 * 128;
    auto kern = vit_attention_mfma<KB>;
    if (cap_kernel_setup((const void*)kern, lds, nullptr) != 0) return -1;
    hipLaunchKernelGGL(kern, dim3(B * H), dim3(256), lds, s, (const bf16_t*)qkv, (bf16_t*)ctx, N, H);
    CAP_HIP_CHECK(hipGetLastError());
    return 0;
}

template <int KB, int KC>
int launch_flash_kb(const void* qkv, void* ctx, int B, int N, int H, hipStream_t s) {
    const int lds = 2 * KB * 32 * 128;
    auto kern = vit_attention_flash<KB, KC>;
    if (cap_kernel_setup((const void*)kern, lds, nullptr) != 0) return -1;
    hipLaunchKernelGGL(kern, dim3(B * H), dim3(256), lds, s, (const bf16_t*)qkv, (bf16_t*)ctx, N, H);
    CAP_HIP_CHECK(hipGetLastError());
    return 0;
}

template <int KB, int KC>
int launch_flash_wide(const void* qkv, void* ctx, int B, int N, int H, int hd, int causal, hipStream_t s) {
    const int lds = 4 * KB * 32 * 128;
    auto kern = vit_attention_flash_wide<KB, KC>;
    if (cap_kernel_setup((const void*)kern, lds, nullptr) != 0) return -1;
    hipLaunchKernelGGL(kern, dim3(B * H), dim3(256), lds, s, (const bf16_t*)qkv, (bf16_t*)ctx, N, H, hd, causal);
    CAP_HIP_CHECK(hipGetLastError());
    return 0;
}

template <int KC, bool ONE, bool EXACT = false>
int launch_split_kc(const void* qkv, void* ctx, int B, int N, int H, hipStream_t s) {
    const int lds = 2 * KC * 32 * 256;
    auto kern = vit_attention_split<KC, ONE, EXACT>;
    if (cap_kernel_setup((const void*)kern, lds, nullptr) != 0) return -1;
    const int QG = ((N + 31) / 32 + 7) / 8;               // groups of 8 query tiles
    hipLaunchKernelGGL(kern, dim3(B * H * QG), dim3(512), lds, s, (const g8_t*)qkv, (g8_t*)ctx, N, H, QG);
    CAP_HIP_CHECK(hipGetLastError());
    return 0;
}

template <int KC>
int launch_split_pw(const void* qkv, void* ctx, int B, int N, int H, hipStream_t s) {
    const int lds = 2 * KC * 32 * 256;
    auto kern = vit_attention_split_pw<KC>;
    int n_cu = 0;
    if (cap_kernel_setup((const void*)kern, lds, &n_cu) != 0) return -1;
    const int units = B * H;
    hipLaunchKernelGGL(kern, dim3(units < n_cu ? units : n_cu), dim3(512), lds, s, (const g8_t*)qkv, (g8_t*)ctx, N, H, units);
    CAP_HIP_CHECK(hipGetLastError());
    return 0;
}

template <int KB, typename TO>
int launch_f32_mfma_kb(const void* qkv, void* ctx, int B, int N, int H, hipStream_t s) {
    const int lds = 2 * KB * 32 * 68 * 4;
    auto kern = vit_attention_f32_mfma<KB, TO>;
    if (cap_kernel_setup((const void*)kern, lds, nullptr) != 0) return -1;
    hipLaunchKernelGGL(kern, dim3(B * H), dim3(256), lds, s, (const float*)qkv, (TO*)ctx, N, H);
    CAP_HIP_CHECK(hipGetLastError());
    return 0;
}

}  // namespace

template <typename TO>
int launch_f32_mfma_wide(const void* qkv, void* ctx, int B, int N, int H, int hd, hipStream_t s) {
    const int nqt = (N + 31) / 32, QG = (nqt + 3) / 4;
    const int ndb = (hd + 31) / 32;
    const int lds = 2 * 3 * 32 * (ndb * 32 + 4) * 4;
#define CAP_WIDE(NDBB)                                                                                                \
    do {                                                                                                               \
        auto kern = vit_attention_f32_mfma_wide<3, NDBB, TO>;                                                          \
        if (cap_kernel_setup((const void*)kern, lds, nullptr) != 0) return -1;                                          \
        hipLaunchKernelGGL(kern, dim3(B * H * QG), dim3(256), lds, s, (const float*)qkv, (TO*)ctx, N, H, hd, QG);          \
    } while (0)
    if (ndb == 1) CAP_WIDE(1); else if (ndb == 2) CAP_WIDE(2); else CAP_WIDE(3);
#undef CAP_WIDE
    CAP_HIP_CHECK(hipGetLastError());
    return 0;
}

bool vit_attention_takes_g8(int N) { return N >= 1; }     // any token count: the keys are walked in chunks

int launch_vit_attention(int dtype, const void* qkv, void* ctx, int B, int N, int H, int impl, hipStream_t s, int head_dim, int causal,
                         int out_dtype) {
    if (out_dtype < 0) out_dtype = dtype;
    if (dtype == CAP_DT_G8) {              // G8 q|k|v (the split mode's qkv GEMM output): the split-fp16 MFMA kernel
        if (out_dtype != CAP_DT_G8 || head_dim != 64 || causal || !vit_attention_takes_g8(N)) {
            cap_set_error("vit_attention: G8 q|k|v need head_dim 64 and no mask (N=%d)", N);
            return -1;
        }
        if (N <= 64) return launch_split_kc<2, true>(qkv, ctx, B, N, H, s);       // fixture-sized inputs
        // 197 tokens: one pass, 7 key blocks.  More units than CUs: the persistent form (loads of the next unit under the current
        // one's arithmetic); impl 5 = the per-unit kernel (tests, A/B)
        if (N > 192 && N <= 224 && impl != 5 && B * H > 256) return launch_split_pw<7>(qkv, ctx, B, N, H, s);
        if (N > 192 && N <= 224) return launch_split_kc<7, true, true>(qkv, ctx, B, N, H, s);
        if (N <= 224) return launch_split_kc<7, true>(qkv, ctx, B, N, H, s);
        return launch_split_kc<4, false>(qkv, ctx, B, N, H, s);                  // chunks of 128 keys, online softmax
    }
    if (out_dtype != dtype && !(dtype == CAP_DT_F32 && out_dtype == CAP_DT_G8)) {
        cap_set_error("vit_attention: output type %d for input type %d is not supported here", out_dtype, dtype);
        return -1;
    }
    if (head_dim > 64 && head_dim <= 128 && head_dim % 8 == 0 && dtype == CAP_DT_BF16 && impl != 1) {
        if ((N + 31) / 32 == 9) return launch_flash_wide<9, 5>(qkv, ctx, B, N, H, head_dim, causal, s);   // ViT-g/14 of BLIP-2: 88-wide heads, 257 tokens
        if (N <= 64) return launch_flash_wide<2, 2>(qkv, ctx, B, N, H, head_dim, causal, s);               // OPT prefill: 80-wide heads, 33 positions, causal
    }
    if (dtype == CAP_DT_F32 && !causal && impl != 1 && head_dim != 64 && head_dim % 8 == 0 && head_dim <= 96) {
        // exact-product fp32 MFMA attention for wide heads (BLIP-2's ViT-g: 88), keys in chunks; fp32 or G8 context
        return out_dtype == CAP_DT_G8 ? launch_f32_mfma_wide<g8_t>(qkv, ctx, B, N, H, head_dim, s)
                                      : launch_f32_mfma_wide<float>(qkv, ctx, B, N, H, head_dim, s);
    }
    if (head_dim != 64 || causal) {   // any other width / length, or a causal mask at width 64: generic kernel over the fused qkv rows
        const long D = (long)H * head_dim;
        const char* base = (const char*)qkv;
        const size_t e = dtype == CAP_DT_BF16 ? 2 : 4;
        return launch_generic_attention(dtype, base, 3 * D, (long)N * 3 * D, base + D * e, 3 * D, (long)N * 3 * D, base + 2 * D * e,
                                        3 * D, (long)N * 3 * D, ctx, D, (long)N * D, B, N, N, H, head_dim, causal ? 0 : -1, s, out_dtype);
    }
    const int kb = (N + 31) / 32;
    const bool mfma_ok = dtype == CAP_DT_BF16 && (kb == 1 || kb == 7 || kb == 9 || kb == 19);
    if (impl == 2 && !mfma_ok) {
        cap_set_error("vit_attention: MFMA path needs bf16 and 1, 7, 9 or 19 key blocks (N=%d)", N);
        return -1;
    }
    if (dtype == CAP_DT_F32 && impl != 1 && (kb == 1 || kb == 7 || kb == 9)) {      // fp32 MFMA kernel (impl 1 forces the scalar one)
        const bool g8 = out_dtype == CAP_DT_G8;
        if (kb == 1) return g8 ? launch_f32_mfma_kb<1, g8_t>(qkv, ctx, B, N, H, s) : launch_f32_mfma_kb<1, float>(qkv, ctx, B, N, H, s);
        if (kb == 7) return g8 ? launch_f32_mfma_kb<7, g8_t>(qkv, ctx, B, N, H, s) : launch_f32_mfma_kb<7, float>(qkv, ctx, B, N, H, s);
        return g8 ? launch_f32_mfma_kb<9, g8_t>(qkv, ctx, B, N, H, s) : launch_f32_mfma_kb<9, float>(qkv, ctx, B, N, H, s);
    }
    if (impl == 0) impl = mfma_ok ? 2 : 1;
    if (impl == 2) {
        if (kb == 1) return launch_mfma_kb<1>(qkv, ctx, B, N, H, s);
        if (kb == 7) return launch_mfma_kb<7>(qkv, ctx, B, N, H, s);
        if (kb == 19) return launch_flash_kb<19, 5>(qkv, ctx, B, N, H, s);
        return launch_mfma_kb<9>(qkv, ctx, B, N, H, s);
    }
    dim3 grid(B * H, (N + 255) / 256);
    if (dtype == CAP_DT_BF16)
        hipLaunchKernelGGL(vit_attention_scalar<bf16_t>, grid, dim3(256), 0, s, (const bf16_t*)qkv, (bf16_t*)ctx, N, H);
    else if (out_dtype == CAP_DT_G8)
        hipLaunchKernelGGL((vit_attention_scalar<float, g8_t>), grid, dim3(256), 0, s, (const float*)qkv, (g8_t*)ctx, N, H);
    else
        hipLaunchKernelGGL(vit_attention_scalar<float>, grid, dim3(256), 0, s, (const float*)qkv, (float*)ctx, N, H);
    CAP_HIP_CHECK(hipGetLastError());
    return 0;
}

int launch_decode_attention(int dtype, const void* q, const void* kbase, const void* vbase, const int* anc,
                            int anc_ld, int rows_per_kv, int kv_ld, int n_keys, void* out, int R, int H, int impl,
                            hipStream_t s, const float* q_part, int q_S, const float* q_bias, int q_ld, int q_col0,
                            int append_kv, int out_dtype, const int* skip_rows, int kv16, size_t kv_row0, RowMap map) {
    // kv16: kbase / vbase are the bases of KV16 blocks (common.h), the launch's first row has index kv_row0 in them
    if (out_dtype < 0) out_dtype = dtype;
    if (kv16 && !(dtype == CAP_DT_F32 && impl == 0 && !anc && !append_kv && n_keys > 32)) {
        cap_set_error("decode_attention: a KV16 cache is the split / fp32 modes' cross-attention cache (no ancestry, > 32 keys)");
        return -1;
    }
    if (out_dtype != dtype && !(dtype == CAP_DT_F32 && out_dtype == CAP_DT_G8 && impl == 0)) {
        cap_set_error("decode_attention: output type %d for input type %d is not supported here", out_dtype, dtype);
        return -1;
    }
    QSource qs;
    qs.part = q_part; qs.bias = q_bias; qs.S = q_S; qs.part_ld = q_ld; qs.col0 = q_col0; qs.append_kv = append_kv;
    if (q_part && (impl != 0 || q_S < 1 || !q_bias || (q_ld & 3) || (q_col0 & 3))) {
        cap_set_error("decode_attention: fused split-K query needs impl 0, bias and 16-byte aligned columns");
        return -1;
    }
    if (n_keys <= 0 || n_keys > 8192) { cap_set_error("decode_attention: bad key count %d", n_keys); return -1; }
    if ((map.live || map.n) && (impl != 0 || rows_per_kv != 1 || anc)) {
        cap_set_error("decode_attention: a row map is taken by the greedy kernels only (impl 0, one row per K/V block, no ancestry)");
        return -1;
    }
#define CAP_DA_WAVE_O(TT, NI, TOO)                                                                                     \
    hipLaunchKernelGGL((decode_attention_wave_kernel<TT, NI, TOO>), dim3((R * H + 3) / 4), dim3(256), 0, s, (const TT*)q,  \
                       (TT*)kbase, (TT*)vbase, anc, anc_ld, rows_per_kv, kv_ld, n_keys, (TOO*)out, R, H, qs, skip_rows, map)
#define CAP_DA_WAVE(TT, NI) CAP_DA_WAVE_O(TT, NI, TT)
    // bf16: chunks of 40 keys, double-buffered (168 VGPRs -> 3 waves/SIMD, all of a 256-row launch resident at once;
    // 197 image tokens = 5 chunks).  fp32: chunks of 56 keys, single buffer (same register budget).
    // greedy (one row per K/V block): the stream is read exactly once per launch, non-temporal loads keep it from displacing
    // the decode GEMMs' weights in L2 (-6 % on the kernel: 34.3 -> 32.3 us at 256 rows x 197 keys).  With beams the K rows
    // of an image share the block through L2 and the hint costs +17 %, so it is not used there.
#define CAP_DA_ONLINE_NT(TT, DBB, NTT)                                                                                 \
    hipLaunchKernelGGL((decode_attention_online_kernel<TT, (DBB ? 5 : 7), DBB, NTT>), dim3((R * H + 3) / 4), dim3(256), 0, \
                       s, (const TT*)q, (const TT*)kbase, (const TT*)vbase, anc, anc_ld, rows_per_kv, kv_ld, n_keys,      \
                       (TT*)out, R, H, qs, skip_rows, (size_t)0, map)
#define CAP_DA_ONLINE(TT, DBB)                                                                                         \
    do {                                                                                                               \
        if (DBB && rows_per_kv == 1 && !anc) CAP_DA_ONLINE_NT(TT, DBB, DBB); else CAP_DA_ONLINE_NT(TT, DBB, false);    \
    } while (0)
    const int ng8 = (n_keys + 7) / 8;            // groups of 8 keys
    if (append_kv && ng8 > 4 && q_part) {
        cap_set_error("decode_attention: fused k/v append supports up to 32 positions (got %d)", n_keys);
        return -1;
    }
    // beams of an image over its shared block (cross-attention, no ancestry): one wave per (image, head) for all beams
    if (impl == 0 && !anc && ng8 > 4 && rows_per_kv >= 2 && rows_per_kv <= 5 && R % rows_per_kv == 0 && !append_kv) {
        const int n_img = R / rows_per_kv;
#define CAP_DA_SHARED(TT, GG, NBB, TOO, TKK)                                                                            \
    hipLaunchKernelGGL((decode_attention_shared_kernel<TT, GG, NBB, TOO, TKK>), dim3((n_img * H + 3) / 4), dim3(256), 0, s, \
                       (const TT*)q, (const TT*)kbase, (const TT*)vbase, kv_ld, n_keys, (TOO*)out, n_img, H, qs, skip_rows, kv_row0)
#define CAP_DA_SHARED_NB(TT, GG, TOO, TKK)                                                                              \
    switch (rows_per_kv) {                                                                                             \
        case 2: CAP_DA_SHARED(TT, GG, 2, TOO, TKK); break;                                                             \
        case 3: CAP_DA_SHARED(TT, GG, 3, TOO, TKK); break;                                                             \
        case 4: CAP_DA_SHARED(TT, GG, 4, TOO, TKK); break;                                                             \
        default: CAP_DA_SHARED(TT, GG, 5, TOO, TKK); break;                                                            \
    }
        if (dtype == CAP_DT_BF16) CAP_DA_SHARED_NB(bf16_t, 5, bf16_t, bf16_t)
        else if (out_dtype == CAP_DT_G8 && kv16) CAP_DA_SHARED_NB(float, 5, g8_t, kv16_t)       // (the online kernel's chunking: same bits per row)
        else if (out_dtype == CAP_DT_G8) CAP_DA_SHARED_NB(float, 7, g8_t, float)
        else if (kv16) CAP_DA_SHARED_NB(float, 5, float, kv16_t)
        else CAP_DA_SHARED_NB(float, 7, float, float)
#undef CAP_DA_SHARED_NB
#undef CAP_DA_SHARED
        CAP_HIP_CHECK(hipGetLastError());
        return 0;
    }
    if (impl == 0) {
        if (dtype == CAP_DT_BF16) {
            if (ng8 <= 1) CAP_DA_WAVE(bf16_t, 1); else if (ng8 <= 2) CAP_DA_WAVE(bf16_t, 2);
            else if (ng8 <= 4) CAP_DA_WAVE(bf16_t, 4); else CAP_DA_ONLINE(bf16_t, true);
        } else if (out_dtype == CAP_DT_G8) {      // split mode: fp32 caches, the context row is the next GEMM's G8 operand
            if (ng8 <= 1) CAP_DA_WAVE_O(float, 1, g8_t); else if (ng8 <= 2) CAP_DA_WAVE_O(float, 2, g8_t);
            else if (ng8 <= 4) CAP_DA_WAVE_O(float, 4, g8_t);
            else if (kv16)
                // chunks of 40 keys, one register buffer: 31.5 us per launch at 256 rows x 197 keys against 32.9 with chunks of 56
                // and 32.7 / 31.7 / 68.7 double-buffered with 32 / 24 / 40 keys per chunk (tools/bench_cross_attention.py): with
                // ~4 us of launch, 160 MB in the rest is the HBM rate
                hipLaunchKernelGGL((decode_attention_online_kernel<float, 5, false, false, g8_t, kv16_t>), dim3((R * H + 3) / 4), dim3(256), 0,
                                   s, (const float*)q, (const float*)kbase, (const float*)vbase, anc, anc_ld, rows_per_kv, kv_ld,
                                   n_keys, (g8_t*)out, R, H, qs, skip_rows, kv_row0, map);
            else
                hipLaunchKernelGGL((decode_attention_online_kernel<float, 7, false, false, g8_t>), dim3((R * H + 3) / 4), dim3(256), 0,
                                   s, (const float*)q, (const float*)kbase, (const float*)vbase, anc, anc_ld, rows_per_kv, kv_ld,
                                   n_keys, (g8_t*)out, R, H, qs, skip_rows, (size_t)0, map);
        } else {
            if (ng8 <= 1) CAP_DA_WAVE(float, 1); else if (ng8 <= 2) CAP_DA_WAVE(float, 2);
            else if (ng8 <= 4) CAP_DA_WAVE(float, 4);
            else if (kv16)
                hipLaunchKernelGGL((decode_attention_online_kernel<float, 5, false, false, float, kv16_t>), dim3((R * H + 3) / 4), dim3(256), 0,
                                   s, (const float*)q, (const float*)kbase, (const float*)vbase, anc, anc_ld, rows_per_kv, kv_ld,
                                   n_keys, (float*)out, R, H, qs, skip_rows, kv_row0, map);
            else CAP_DA_ONLINE(float, false);
        }
        CAP_HIP_CHECK(hipGetLastError());
        return 0;
    }
#undef CAP_DA_WAVE
#undef CAP_DA_WAVE_O
#undef CAP_DA_ONLINE
#undef CAP_DA_ONLINE_NT
    // impl 1: the simple two-pass kernel (kept as an independent implementation for the tests)
    const int lds = (((n_keys + 3) & ~3) + 8 + 256) * 4;
    dim3 grid(R, H);
    if (dtype == CAP_DT_BF16)
        hipLaunchKernelGGL(decode_attention_kernel<bf16_t>, grid, dim3(256), lds, s, (const bf16_t*)q,
                           (const bf16_t*)kbase, (const bf16_t*)vbase, anc, anc_ld, rows_per_kv, kv_ld, n_keys,
                           (bf16_t*)out, H);
    else
        hipLaunchKernelGGL(decode_attention_kernel<float>, grid, dim3(256), lds, s, (const float*)q,
                           (const float*)kbase, (const float*)vbase, anc, anc_ld, rows_per_kv, kv_ld, n_keys,
                           (float*)out, H);
    CAP_HIP_CHECK(hipGetLastError());
    return 0;
}

// fp32 rows [n_rows, 64] -> one KV16 block (common.h): what the cross-K/V GEMM's epilogue writes, as a kernel of its own (tests).
// One wave per 4 rows: 16 lanes per row, 4 values per lane; the row maximum goes through two DPP-free shuffles.
__global__ void pack_kv16_kernel(const float* __restrict__ src, char* __restrict__ dst, size_t n_rows) {
    const size_t r = ((size_t)blockIdx.x * blockDim.x + threadIdx.x) >> 4;
    const int d = (int)(threadIdx.x & 15) * 4;
    if (r >= n_rows) return;                             // (whole 16-lane groups leave together)
    const float4 v = *(const float4*)(src + r * 64 + d);
    float am = fmaxf(fmaxf(fabsf(v.x), fabsf(v.y)), fmaxf(fabsf(v.z), fabsf(v.w)));
    am = fmaxf(am, __shfl_xor(am, 1, 64)); am = fmaxf(am, __shfl_xor(am, 2, 64));
    am = fmaxf(am, __shfl_xor(am, 4, 64)); am = fmaxf(am, __shfl_xor(am, 8, 64));
    float sc, inv;
    kv16_scales(am, sc, inv);
    uint2 w;
    w.x = kv16_pack2(v.x, v.y, inv);
    w.y = kv16_pack2(v.z, v.w, inv);
    *(uint2*)(dst + kv16_row_off(r) + d * 2) = w;
    if (d == 0) *(float*)(dst + kv16_scale_off(r)) = sc;
}
int launch_pack_kv16(const float* src, void* dst, size_t n_rows, hipStream_t s) {
    const size_t threads = n_rows * 16;
    hipLaunchKernelGGL(pack_kv16_kernel, dim3((unsigned)((threads + 255) / 256 ? (threads + 255) / 256 : 1)), dim3(256), 0, s, src, (char*)dst, n_rows);
    CAP_HIP_CHECK(hipGetLastError());
    return 0;
}

int launch_generic_attention(int dtype, const void* q, long ldq, long qbs, const void* k, long ldk, long kbs, const void* v,
                             long ldv, long vbs, void* out, long ldo, long obs, int B, int Lq, int Lk, int H, int hd,
                             int causal_off, hipStream_t s, int out_dtype) {
    if (hd < 8 || hd > 128 || B < 1 || Lq < 1 || Lk < 1) { cap_set_error("generic_attention: head_dim %d / shape unsupported", hd); return -1; }
    if (out_dtype < 0) out_dtype = dtype;
    const bool g8o = out_dtype == CAP_DT_G8;               // split mode: fp32 q / k / v in, the context is the next GEMM's G8 operand
    if (out_dtype != dtype && !(dtype == CAP_DT_F32 && g8o && ldo % 8 == 0 && obs % 8 == 0)) {
        cap_set_error("generic_attention: output type %d for input type %d is not supported here", out_dtype, dtype);
        return -1;
    }
    const float scale = 1.0f / sqrtf((float)hd);
    if (Lq == 1 && Lk <= 1024 && hd % 8 == 0) {       // decode step: one query per (batch, head)
        if (dtype == CAP_DT_BF16)
            hipLaunchKernelGGL(generic_decode_attention_kernel<bf16_t>, dim3(B * H), dim3(64), 0, s, (const bf16_t*)q, qbs, (const bf16_t*)k,
                               ldk, kbs, (const bf16_t*)v, ldv, vbs, (bf16_t*)out, obs, Lk, H, hd, scale);
        else if (g8o)
            hipLaunchKernelGGL((generic_decode_attention_kernel<float, g8_t>), dim3(B * H), dim3(64), 0, s, (const float*)q, qbs, (const float*)k,
                               ldk, kbs, (const float*)v, ldv, vbs, (g8_t*)out, obs, Lk, H, hd, scale);
        else
            hipLaunchKernelGGL(generic_decode_attention_kernel<float>, dim3(B * H), dim3(64), 0, s, (const float*)q, qbs, (const float*)k,
                               ldk, kbs, (const float*)v, ldv, vbs, (float*)out, obs, Lk, H, hd, scale);
        CAP_HIP_CHECK(hipGetLastError());
        return 0;
    }
    const int hdp = hd <= 32 ? 32 : hd <= 64 ? 64 : hd <= 96 ? 96 : 128;
    if (causal_off < 0 && Lq <= 64 && Lk >= 16 && hdp <= 64) {        // one unmasked query block: keys over four waves (the Q-Former)
#define CAP_GK(TT, HDP, TO)                                                                                            \
    do {                                                                                                               \
        constexpr int lds = ((4 * 2 * 32 * HDP) > (4 * (HDP + 2) * 64) ? (4 * 2 * 32 * HDP) : (4 * (HDP + 2) * 64)) * 4;   \
        auto kk = generic_attention_kp_kernel<TT, HDP, TO>;                                                             \
        if (cap_kernel_setup((const void*)kk, lds, nullptr) != 0) return -1;                                            \
        hipLaunchKernelGGL(kk, dim3(B * H), dim3(256), lds, s, (const TT*)q, ldq, qbs, (const TT*)k, ldk, kbs, (const TT*)v, ldv, vbs, \
                           (TO*)out, ldo, obs, Lq, Lk, H, hd, scale);                                                   \
    } while (0)
#define CAP_GK_T(TT, TO) do { if (hdp == 32) CAP_GK(TT, 32, TO); else CAP_GK(TT, 64, TO); } while (0)
        if (dtype == CAP_DT_BF16) CAP_GK_T(bf16_t, bf16_t); else if (g8o) CAP_GK_T(float, g8_t); else CAP_GK_T(float, float);
#undef CAP_GK_T
#undef CAP_GK
        CAP_HIP_CHECK(hipGetLastError());
        return 0;
    }
    const dim3 grid(B * H * ((Lq + 63) / 64));
#define CAP_GA(TT, HDP, TO)                                                                                            \
    hipLaunchKernelGGL((generic_attention_kernel<TT, HDP, TO>), grid, dim3(64), 0, s, (const TT*)q, ldq, qbs, (const TT*)k, ldk, \
                       kbs, (const TT*)v, ldv, vbs, (TO*)out, ldo, obs, Lq, Lk, H, hd, causal_off, scale)
#define CAP_GA_T(TT, TO)                                                                                               \
    do { if (hdp == 32) CAP_GA(TT, 32, TO); else if (hdp == 64) CAP_GA(TT, 64, TO); else if (hdp == 96) CAP_GA(TT, 96, TO); else CAP_GA(TT, 128, TO); } while (0)
    if (dtype == CAP_DT_BF16) CAP_GA_T(bf16_t, bf16_t); else if (g8o) CAP_GA_T(float, g8_t); else CAP_GA_T(float, float);
#undef CAP_GA_T
#undef CAP_GA
    CAP_HIP_CHECK(hipGetLastError());
    return 0;
}

int launch_opt_decode_attention(int dtype, const void* qkv, void* kc, void* vc, void* out, int B, int T, int H, int Lmax,
                                int past, hipStream_t s, int out_dtype) {
    if (out_dtype < 0) out_dtype = dtype;
    if (out_dtype != dtype && !(dtype == CAP_DT_F32 && out_dtype == CAP_DT_G8 && T % 8 == 0)) {
        cap_set_error("opt_decode_attention: output type %d for input type %d is not supported here", out_dtype, dtype);
        return -1;
    }
    const int hd = H > 0 ? T / H : 0;
    if (B < 1 || H < 1 || T % H != 0 || hd % 8 != 0 || hd > 128 || past < 0 || past + 1 > 1024 || past >= Lmax) {
        cap_set_error("opt_decode_attention: unsupported shape T=%d H=%d past=%d Lmax=%d", T, H, past, Lmax);
        return -1;
    }
    const float scale = 1.0f / sqrtf((float)hd);
#define CAP_ODA(TT, H8, TO)                                                                                             \
    hipLaunchKernelGGL((opt_decode_attention_kernel<TT, H8, TO>), dim3(B * H), dim3(64), 0, s, (const TT*)qkv, (TT*)kc, (TT*)vc,  \
                       (TO*)out, T, H, hd, Lmax, past, scale)
    if (dtype == CAP_DT_BF16) {
        if (hd == 80) CAP_ODA(bf16_t, 10, bf16_t); else if (hd == 64) CAP_ODA(bf16_t, 8, bf16_t); else if (hd == 128) CAP_ODA(bf16_t, 16, bf16_t); else CAP_ODA(bf16_t, 0, bf16_t);
    } else if (out_dtype == CAP_DT_G8) {
        if (hd == 80) CAP_ODA(float, 10, g8_t); else if (hd == 64) CAP_ODA(float, 8, g8_t); else if (hd == 128) CAP_ODA(float, 16, g8_t); else CAP_ODA(float, 0, g8_t);
    } else {
        if (hd == 80) CAP_ODA(float, 10, float); else if (hd == 64) CAP_ODA(float, 8, float); else if (hd == 128) CAP_ODA(float, 16, float); else CAP_ODA(float, 0, float);
    }
#undef CAP_ODA
    CAP_HIP_CHECK(hipGetLastError());
    return 0;
}
int launch_opt_prefill_inputs(const float* proj, const float* tok, const float* pos, float* x, int B, int nq, int T, int bos,
                              hipStream_t s) {
    hipLaunchKernelGGL(opt_prefill_inputs_kernel, dim3(256), dim3(256), 0, s, proj, tok, pos, x, B, nq, T, bos);
    CAP_HIP_CHECK(hipGetLastError());
    return 0;
}
int launch_opt_token_inputs(const int* seq, int seq_ld, int cur, const float* tok, const float* pos, float* x, int B, int T,
                            hipStream_t s) {
    hipLaunchKernelGGL(opt_token_inputs_kernel, dim3(64), dim3(256), 0, s, seq, seq_ld, cur, tok, pos, x, B, T);
    CAP_HIP_CHECK(hipGetLastError());
    return 0;
}
int launch_kv_append(int dtype, const void* qkv, void* kc, void* vc, int B, int L, int T, int Lmax, int pos0, hipStream_t s) {
    if (dtype == CAP_DT_BF16)
        hipLaunchKernelGGL(kv_append_kernel<bf16_t>, dim3(256), dim3(256), 0, s, (const bf16_t*)qkv, (bf16_t*)kc, (bf16_t*)vc, B, L, T, Lmax, pos0);
    else
        hipLaunchKernelGGL(kv_append_kernel<float>, dim3(256), dim3(256), 0, s, (const float*)qkv, (float*)kc, (float*)vc, B, L, T, Lmax, pos0);
    CAP_HIP_CHECK(hipGetLastError());
    return 0;
}
int launch_rows_broadcast(int dtype, const float* src, float* dst_f, void* dst_t, int B, int n, int D, hipStream_t s) {
    if (dtype == CAP_DT_BF16)
        hipLaunchKernelGGL(rows_broadcast_kernel<bf16_t>, dim3(256), dim3(256), 0, s, src, dst_f, (bf16_t*)dst_t, B, n, D);
    else if (dtype == CAP_DT_G8)
        hipLaunchKernelGGL(rows_broadcast_kernel<g8_t>, dim3(256), dim3(256), 0, s, src, dst_f, (g8_t*)dst_t, B, n, D);
    else
        hipLaunchKernelGGL(rows_broadcast_kernel<float>, dim3(256), dim3(256), 0, s, src, dst_f, (float*)dst_t, B, n, D);
    CAP_HIP_CHECK(hipGetLastError());
    return 0;
}

int launch_text_attention(int dtype, const void* qkv, const int* lens, void* ctx, int B, int L, int H, int head_dim,
                          hipStream_t s) {
    if ((head_dim != 32 && head_dim != 64) || L < 1 || L > 512) {
        cap_set_error("text_attention: head_dim %d / length %d unsupported (32 or 64, 1..512)", head_dim, L);
        return -1;
    }
    const int lds = 2 * L * head_dim * 4;
#define CAP_TA(TT, HDD)                                                                                                \
    hipLaunchKernelGGL((text_attention_kernel<TT, HDD>), dim3(B * H), dim3(64), lds, s, (const TT*)qkv, lens, (TT*)ctx, L, H)
    if (dtype == CAP_DT_BF16) { if (head_dim == 32) CAP_TA(bf16_t, 32); else CAP_TA(bf16_t, 64); }
    else { if (head_dim == 32) CAP_TA(float, 32); else CAP_TA(float, 64); }
#undef CAP_TA
    CAP_HIP_CHECK(hipGetLastError());
    return 0;
}

int launch_pool_attention(int dtype, const float* qp, const void* kv, void* out, int B, int N, int Q, int E, int heads,
                          hipStream_t s, int out_dtype) {
    if (out_dtype < 0) out_dtype = dtype;
    if (out_dtype != dtype && !(dtype == CAP_DT_F32 && out_dtype == CAP_DT_G8 && E % 8 == 0)) {
        cap_set_error("pool_attention: output type %d for input type %d is not supported here", out_dtype, dtype);
        return -1;
    }
    const int hd = E / heads;
    if (hd * heads != E || (hd != 64 && hd != 96)) {
        cap_set_error("pool_attention: head_dim %d not supported (64 or 96)", hd);
        return -1;
    }
    dim3 grid(B * heads, (Q + 63) / 64);
#define CAP_POOL(TT, HDV, TO)                                                                                 \
    hipLaunchKernelGGL((pool_attention_kernel<TT, HDV, TO>), grid, dim3(64), 0, s, qp, (const TT*)kv, (TO*)out, N, Q, E, heads)
    if (dtype == CAP_DT_BF16) { if (hd == 64) CAP_POOL(bf16_t, 64, bf16_t); else CAP_POOL(bf16_t, 96, bf16_t); }
    else if (out_dtype == CAP_DT_G8) { if (hd == 64) CAP_POOL(float, 64, g8_t); else CAP_POOL(float, 96, g8_t); }
    else { if (hd == 64) CAP_POOL(float, 64, float); else CAP_POOL(float, 96, float); }
#undef CAP_POOL
    CAP_HIP_CHECK(hipGetLastError());
    return 0;
}

CAP_DEFINE_G8_CLAMP_READER(cap_g8_clamped_attention)
