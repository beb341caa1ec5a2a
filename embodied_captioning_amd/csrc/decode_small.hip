// Small-batch decode path (rows = images x beams <= 16): the reference's real call pattern is ONE crop per call
// (captioner/models/coca/coca.py:27-33, blip2/blip2.py:24-29, agents/goal_exploration/goal_exploration.py:95-105) and
// BASELINE config 1 is 8.  At that size a decoder layer-step is a chain of dependent launches that each stream a few MB of
// weights: what it costs is launches x (dispatch + first bytes + drain), so the 11 launches of the batch path
// (captioner.hip::run_decoder_step: 6 GEMMs, 2 attentions, 3 split-K consumers) become 6 here:
//
//   qkv   = GEMM      [prologue: split-K consumer + LayerNorm of the previous layer's FFN]        -> q|k|v partial sums
//   so    = GEMM      [prologue: self-attention of the slice's heads, k/v appended to the cache]  -> split-K slabs
//   cross = per (row, head): LayerNorm of the row, its 64 query columns, cross-attention          -> context row (operand type)
//   co    = GEMM                                                                                  -> split-K slabs
//   f1    = GEMM      [prologue: split-K consumer + LayerNorm]                bias + GELU         -> operand type
//   f2    = GEMM                                                                                  -> split-K slabs
//
// Every consumer runs inside the NEXT kernel's prologue, redundantly in each workgroup that needs its result (a few KB per
// row out of L2), and the fp32 LayerNorm output that the batch path keeps in `dx` (the post-LN residual) is written by one
// designated workgroup into a ping-pong pair, so no workgroup reads a row another one is replacing.
//
// Same bits as the batch path, by construction and by test (tests/test_small_decode_gpu.py): a GEMM forms exactly the partial
// sums of gemm_rows_kernel - K slices S from the same plan (captioner.hip::decode_splitk), inside a slice slab j goes to
// chain j % 4, a chain is the MFMA sequence [w_lo.a_hi, w_hi.a_lo, w_hi.a_hi] per slab in slab order, the four chains are
// summed in chain order, slices in slice order, then bias, then residual - or, for the transform / vocabulary GEMMs, the one
// chain per output of the register-staged tile; the LayerNorm is ln.h's wave-per-row form (the statistics order every decode
// LayerNorm kernel shares), the attentions are decode_attn.h's unit functions.  What changes is WHERE a sum is formed: a
// 16-column block of W per wave straight from global memory into MFMA operand registers (a lane's 8 k values of a G8 row are 32
// contiguous bytes), no LDS staging of W, up to 12 slabs in flight per wave.
#include "gemm_tile.h"
#include "ln.h"
#include "decode_attn.h"
#include "ops.h"
#include "decode_small.h"
#include "decode_frag.h"

#ifdef CAP_EXPERIMENTS
// cycle stamps of workgroup 0 of the LAST launch (tools/bench_small_decode.py --stamps): [kernel][2 i] = s_memtime (shader clock),
// [kernel][2 i + 1] = the 100 MHz wall clock; kernel 0 = cross, 1 = GEMM
__device__ unsigned long long g_small_stamps[2][32];
#define SMALL_STAMP(k, i, cond)                                                                          \
    do {                                                                                                 \
        if (blockIdx.x == 0 && (threadIdx.x & 63) == 0 && (cond)) {                                      \
            g_small_stamps[k][2 * (i)] = __builtin_readcyclecounter();                                   \
            g_small_stamps[k][2 * (i) + 1] = wall_clock64();                                             \
        }                                                                                                \
    } while (0)
#else
#define SMALL_STAMP(k, i, cond) do { } while (0)
#endif

namespace {

constexpr int PRO_GLOBAL = SMALL_PRO_GLOBAL, PRO_LN = SMALL_PRO_LN, PRO_SELFATTN = SMALL_PRO_SELFATTN;
constexpr int SEPI_PARTIAL = SMALL_EPI_PARTIAL, SEPI_ACT_T = SMALL_EPI_ACT_T, SEPI_ACT_F32 = SMALL_EPI_ACT_F32;

// ------------------------------------------------------------------------------------------------------------------
// GEMM for R <= 16 rows: C[R, N] = A[R, K] . W[N, K]^T.
//   NCHAIN == 4: the sums of gemm_rows_kernel.  Workgroup = (16-column block tn, K slice kz); its four GEMM waves are the four
//                chains of the slice (chain w: slabs w, w + 4, ...), summed in chain order through LDS.
//   NCHAIN == 1: the sums of the register-staged tiles (gemm_kernel): one chain per output over the slice's slabs in order.
//                Workgroup = 64 columns, one 16-column block per wave.
// PRO: where the A operand comes from (see the file header); EPI: split-K slab / bias + activation -> operand type / fp32.
// NW: waves per workgroup (the self-attention prologue uses 8: its units are spread over all of them).
template <typename T, int PRO, int EPI, int NCHAIN, int NW, int RPW>
__global__ __launch_bounds__(NW * 64, 1) void dec_small_gemm_kernel(SmallGemm p) {
    p.W = glob(p.W); p.A = glob(p.A); p.out_part = glob(p.out_part); p.bias = glob(p.bias); p.out = glob(p.out);
    p.ln = glob_ln(p.ln);
    p.sa.qkv_part = glob(p.sa.qkv_part); p.sa.qkv_bias = glob(p.sa.qkv_bias); p.sa.kc = glob(p.sa.kc); p.sa.vc = glob(p.sa.vc);
    p.sa.anc = glob(p.sa.anc); p.sa.skip = glob(p.sa.skip);
    constexpr int SLAB = is_g8<T> ? 32 : 64;
    constexpr int ESZ = is_g8<T> ? 4 : 2;
    constexpr int CT = NCHAIN == 4 ? 16 : 64;
    // slabs per register batch; two batches in flight.  (The self-attention prologue's 8-wave workgroup has 256 registers per
    // lane and its units need ~190: a K slice of a few heads is 1-2 slabs per chain anyway.)
    constexpr int NF = PRO == PRO_SELFATTN ? 2 : (NCHAIN == 1 && RPW < 4 ? 12 : 6);
    using TA = typename AttT<T>::type;
    extern __shared__ __attribute__((aligned(16))) char smem[];

    const int ntn = (p.N + CT - 1) / CT;
    const int tn = blockIdx.x % ntn, kz = blockIdx.x / ntn;
    const int Ks = p.K / p.S, nkb = Ks / SLAB;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int r16 = lane & 15, kg = lane >> 4;
    const bool gw = wave < 4;                               // a GEMM wave
    const int n0 = NCHAIN == 4 ? tn * 16 : tn * 64 + wave * 16;
    const int first = NCHAIN == 4 ? wave : 0, stride = NCHAIN == 4 ? 4 : 1;
    const int n = !gw ? 0 : (NCHAIN == 4 ? (nkb - wave + 3) >> 2 : nkb);            // slabs of this wave
    const bool live = gw && n0 < p.N;

    // LDS image of the A rows: the K slice's columns (self-attention prologue: its heads) or the whole LayerNorm row
    const int img_k = PRO == PRO_LN ? p.K : Ks, img_slab0 = PRO == PRO_LN ? kz * nkb : 0;
    const int pitch = img_k * ESZ + 16;
    char* img = smem;
    float* red = (float*)(smem + (PRO == PRO_GLOBAL ? 0 : 16 * pitch));              // [4][16][16] chain sums

    const char* wrow = (const char*)p.W + ((size_t)min(n0 + r16, p.N - 1) * p.K + (size_t)kz * Ks) * ESZ;
    const char* arow = PRO == PRO_GLOBAL ? (const char*)p.A + ((size_t)min(r16, p.R - 1) * p.K + (size_t)kz * Ks) * ESZ : nullptr;
    const bool arow_live = r16 < p.R;

    Frag wq[2][NF], aq[2][NF];
    auto load_batch = [&](int b, Frag (&w)[NF], Frag (&a)[NF]) {
#pragma unroll
        for (int i = 0; i < NF; ++i) {
            const int j = b * NF + i;
            if (j < n) {
                const size_t off = (size_t)(first + j * stride) * 128;
                w[i] = load_frag<T>(wrow + off, kg);
                if constexpr (PRO == PRO_GLOBAL) a[i] = load_frag<T>(arow + off, kg);
            }
        }
    };
    SMALL_STAMP(1, 0, wave == 0);
    if (live) {
        load_batch(0, wq[0], aq[0]);
        if (n > NF) load_batch(1, wq[1], aq[1]);
    }
    SMALL_STAMP(1, 1, wave == 0);

    // ---- prologue: the A rows of this slice as an LDS image
    if constexpr (PRO == PRO_LN) {
        if (p.K <= 768) ln_rows_prologue<T, RPW, 3>(p.ln, p.R, p.K, img, pitch, wave, NW, lane, blockIdx.x == 0);
        else ln_rows_prologue<T, RPW, 4>(p.ln, p.R, p.K, img, pitch, wave, NW, lane, blockIdx.x == 0);
        __syncthreads();
    } else if constexpr (PRO == PRO_SELFATTN) {
        const int hps = Ks / 64, h0 = kz * hps, units = hps * p.R;
        QSource qs;
        qs.part = p.sa.qkv_part; qs.bias = p.sa.qkv_bias; qs.S = p.sa.qkv_S; qs.part_ld = 3 * p.sa.H * 64; qs.col0 = 0; qs.append_kv = 1;
        const int ng8 = (p.sa.n_keys + 7) / 8;
        // (rows of ended captions are computed like the others: 16 rows at most, and no branch between the units of a wave)
        TA* kc = (TA*)p.sa.kc; TA* vc = (TA*)p.sa.vc;
        const bool wr = tn == 0;
        for (int u = wave; u < units; u += NW) {
            const int row = u / hps, h = h0 + u % hps;
            T* out_row = (T*)(img + (size_t)row * pitch) - h0 * 64;
            if (ng8 <= 1)
                decode_attention_wave_unit<TA, 1, T>(nullptr, kc, vc, p.sa.anc, p.sa.anc_ld, 1, p.sa.kv_ld, p.sa.n_keys, out_row, p.R, p.sa.H, qs, row, h, lane, wr);
            else if (ng8 <= 2)
                decode_attention_wave_unit<TA, 2, T>(nullptr, kc, vc, p.sa.anc, p.sa.anc_ld, 1, p.sa.kv_ld, p.sa.n_keys, out_row, p.R, p.sa.H, qs, row, h, lane, wr);
            else
                decode_attention_wave_unit<TA, 4, T>(nullptr, kc, vc, p.sa.anc, p.sa.anc_ld, 1, p.sa.kv_ld, p.sa.n_keys, out_row, p.R, p.sa.H, qs, row, h, lane, wr);
        }
        __syncthreads();
    }

    SMALL_STAMP(1, 2, wave == 0);
    // ---- the chains
    f32x4 acc = 0.f;
    if (live) {
        auto run_batch = [&](int b, const Frag (&w)[NF], const Frag (&a)[NF]) {
#pragma unroll
            for (int i = 0; i < NF; ++i) {
                const int j = b * NF + i;
                if (j < n) {
                    Frag af;
                    if constexpr (PRO == PRO_GLOBAL) af = arow_live ? a[i] : zero_frag();
                    else af = arow_live ? load_frag<T>(img + (size_t)r16 * pitch + (size_t)(img_slab0 + first + j * stride) * 128, kg) : zero_frag();
                    mma_slab<T>(acc, w[i], af);
                }
            }
        };
        for (int b = 0; b * NF < n; b += 2) {
            run_batch(b, wq[0], aq[0]);
            if ((b + 2) * NF < n) load_batch(b + 2, wq[0], aq[0]);
            if ((b + 1) * NF < n) {
                run_batch(b + 1, wq[1], aq[1]);
                if ((b + 3) * NF < n) load_batch(b + 3, wq[1], aq[1]);
            }
        }
    }

    SMALL_STAMP(1, 3, wave == 0 && acc[0] == acc[0]);
    // ---- epilogue.  acc[e] = C[row r16][n0 + 4 kg + e]
    f32x4 v = acc;
    int row = r16, col = n0 + 4 * kg;
    bool store = live;
    if constexpr (NCHAIN == 4) {
        if (gw) *(f32x4*)(red + (wave * 16 + r16) * 16 + 4 * kg) = acc;
        __syncthreads();
        store = tid < 64;
        if (store) {
            row = tid >> 2;
            const int c4 = tid & 3;
            col = tn * 16 + c4 * 4;
            const float* src = red + row * 16 + c4 * 4;
            v = *(const f32x4*)src;
            v += *(const f32x4*)(src + 256);
            v += *(const f32x4*)(src + 512);
            v += *(const f32x4*)(src + 768);
        }
    }
    SMALL_STAMP(1, 4, wave == 0);
    if (!store || row >= p.R || col >= p.N) return;
    if constexpr (is_g8<T>) v *= (1.0f / G8_WSCALE);
    if constexpr (EPI == SEPI_PARTIAL) {
        *(f32x4*)(p.out_part + ((size_t)kz * p.R + row) * p.N + col) = v;
    } else {
        if (p.bias) v += *(const f32x4*)(p.bias + col);
        if (p.act == 1) {
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] = gelu_for<T>(v[e]);
        } else if (p.act == 2) {
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] = fmaxf(v[e], 0.f);
        }
        if constexpr (EPI == SEPI_ACT_F32) *(f32x4*)((float*)p.out + (size_t)row * p.ldc + col) = v;
        else store4((T*)p.out + (size_t)row * p.ldc, col, make_float4(v[0], v[1], v[2], v[3]));
    }
}

// ------------------------------------------------------------------------------------------------------------------
// Cross-attention block of one (row, head): LayerNorm of the row (split-K consumer of the self-attention output projection),
// the head's 64 query columns (the sums of the batch path's cq GEMM: S slices x 4 chains), attention over the image's K/V.
// Eight waves.  What the kernel waits for is memory round trips (2-3 us each for data the previous kernel wrote), so everything
// is requested at once: wave 7 puts the row's slabs / residual / gamma / beta in flight first, every wave its share of the
// head's query weights (wave = K slice x half of the 64 columns), waves 0-6 the LDS-DMA copy of the (row, head) K/V block -
// 197 rows: 50 KB of KV16 groups, K and V.  Then LayerNorm (wave 7) -> query chains (all waves) -> scores of the 25 key groups
// (all waves, decode_attn.h) -> one wave walks the softmax / P.V with the arithmetic and chunking of
// decode_attention_online_kernel, reading V from LDS: same bits as the batch path.
// KV_LDS false: blocks that do not fit (577-token checkpoints) or short histories: global memory as in the batch path.
template <typename T, typename TKV, int G, bool KV_LDS, int NV>
__global__ __launch_bounds__(512, 1) void dec_small_cross_kernel(SmallCross p) {
    p.W = glob(p.W); p.bias = glob(p.bias); p.kbase = glob(p.kbase); p.vbase = glob(p.vbase); p.skip = glob(p.skip); p.out = glob(p.out);
    p.ln = glob_ln(p.ln);
    constexpr int SLAB = is_g8<T> ? 32 : 64;
    constexpr int ESZ = is_g8<T> ? 4 : 2;
    constexpr int NF = 6;
    using TA = typename AttT<T>::type;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int D = p.D, H = p.H;
    const int row = blockIdx.x / H, h = blockIdx.x - row * H;
    if (p.skip && p.skip[row]) return;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int r16 = lane & 15, kg = lane >> 4;
    SMALL_STAMP(0, 0, wave == 0);

    char* xrow = smem;                                       // the LayerNorm row as a GEMM operand: D * ESZ bytes
    float* qpart = (float*)(smem + ((D * ESZ + 255) & ~255));   // [4][64] slice sums
    float* qfin = qpart + 256;                               // [64]
    float* scl = qfin + 64;                                  // [n_keys rounded up to 64] scores
    char* kimg = (char*)(scl + ((p.n_keys + 63) & ~63));
    // ---- K/V block of (image of this row, head): row indices [rb, rb + n_keys) of the layer's k / v block
    const size_t rb = p.kv_row0 + ((size_t)(row / p.rows_per_kv) * H + h) * p.kv_ld;
    size_t src_off, n16, ri_lds;
    if constexpr (std::is_same<TKV, kv16_t>::value) {
        const size_t g0 = rb >> 5, g1 = (rb + p.n_keys - 1) >> 5;
        src_off = g0 * KV16_GROUP_BYTES; n16 = (g1 - g0 + 1) * (KV16_GROUP_BYTES / 16); ri_lds = rb & 31;
    } else {
        src_off = rb * 64 * sizeof(TKV); n16 = (size_t)p.n_keys * 64 * sizeof(TKV) / 16; ri_lds = 0;
    }
    char* vimg = kimg + n16 * 16;

    // ---- wave 7: the LayerNorm row's loads go first
    const SmallLN& ln = p.ln;
    LnRow lv[NV];
    LnCols lk[NV];
    if (wave == 7) {
        ln_load_row<NV>(ln, p.R, D, row, lane, lv);
        ln_load_cols<NV>(ln, D, lane, lk);
        SMALL_STAMP(0, 10, true);
    }
    // ---- W of this head's query columns: wave = (K slice sl, column half ch): the slice's slabs x 2 column blocks
    const int S = p.S, Ks = D / S, nkb = Ks / SLAB;
    const int sl = wave & 3, ch = wave >> 2;
    const bool gw = sl < S;
    const char* wbase = (const char*)p.W + ((size_t)(h * 64 + ch * 32 + r16) * D + (size_t)sl * Ks) * ESZ;
    Frag wq[NF][2];
    if (gw) {
#pragma unroll
        for (int j = 0; j < NF; ++j)
            if (j < nkb) {
#pragma unroll
                for (int cb = 0; cb < 2; ++cb) wq[j][cb] = load_frag<T>(wbase + (size_t)cb * 16 * D * ESZ + (size_t)j * 128, kg);
            }
    }
    SMALL_STAMP(0, 11, wave == 0);
    SMALL_STAMP(0, 12, wave == 7);
    if constexpr (KV_LDS) {
        if (wave < 7) {
            const char* ks = (const char*)p.kbase + src_off;
            const char* vs = (const char*)p.vbase + src_off;
            for (size_t i = (size_t)wave * 64; i < n16; i += 7 * 64) {      // 1 KiB per wave-instruction, lane-linear in LDS
                if (i + lane < n16) {
                    __builtin_amdgcn_global_load_lds(CAP_GPTR(ks + (i + lane) * 16), CAP_LPTR(kimg + i * 16), 16, 0, 0);
                    __builtin_amdgcn_global_load_lds(CAP_GPTR(vs + (i + lane) * 16), CAP_LPTR(vimg + i * 16), 16, 0, 0);
                }
            }
        }
    }
    SMALL_STAMP(0, 1, wave == 0);
    // ---- LayerNorm of the row (wave 7), operand row to LDS; the head-0 workgroup writes the fp32 row
    if (wave == 7) {
        SMALL_STAMP(0, 7, lv[0].pz[0][0] == lv[0].pz[0][0]);
        ln_finish_row<T, NV>(ln, p.R, D, row, lane, lv, lk, (T*)xrow, h == 0 && ln.x_out ? ln.x_out + (size_t)row * D : nullptr);
        SMALL_STAMP(0, 8, true);
    }
    __syncthreads();
    SMALL_STAMP(0, 2, wave == 0);
    SMALL_STAMP(0, 13, wave == 0 && wq[0][0].x[0][0] == wq[0][0].x[0][0]);
    // ---- the query columns: chain c of slice sl = slabs c, c + 4, ...; the row sits in MFMA column 0
    if (gw) {
        f32x4 acc[4][2];
#pragma unroll
        for (int c = 0; c < 4; ++c)
#pragma unroll
            for (int cb = 0; cb < 2; ++cb) acc[c][cb] = 0.f;
#pragma unroll
        for (int j = 0; j < NF; ++j)
            if (j < nkb) {
                const Frag af = r16 == 0 ? load_frag<T>(xrow + ((size_t)sl * nkb + j) * 128, kg) : zero_frag();
#pragma unroll
                for (int cb = 0; cb < 2; ++cb) mma_slab<T>(acc[j & 3][cb], wq[j][cb], af);
            }
        for (int j = NF; j < nkb; ++j) {                      // longer slices than the register batch: one slab at a time
            const Frag af = r16 == 0 ? load_frag<T>(xrow + ((size_t)sl * nkb + j) * 128, kg) : zero_frag();
#pragma unroll
            for (int cb = 0; cb < 2; ++cb) {
                const Frag w = load_frag<T>(wbase + (size_t)cb * 16 * D * ESZ + (size_t)j * 128, kg);
#pragma unroll
                for (int c = 0; c < 4; ++c)
                    if ((j & 3) == c) { f32x4 t = acc[c][cb]; mma_slab<T>(t, w, af); acc[c][cb] = t; }
            }
        }
        if (r16 == 0) {
#pragma unroll
            for (int cb = 0; cb < 2; ++cb) {
                f32x4 v = acc[0][cb];
                v += acc[1][cb]; v += acc[2][cb]; v += acc[3][cb];
                if constexpr (is_g8<T>) v *= (1.0f / G8_WSCALE);
                *(f32x4*)(qpart + sl * 64 + ch * 32 + cb * 16 + 4 * kg) = v;
            }
        }
    }
    SMALL_STAMP(0, 3, wave == 0);
    if constexpr (KV_LDS) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    SMALL_STAMP(0, 4, wave == 0);
    // ---- q = slices in order + bias, through the attention's value type (Part8::finish)
    if (tid < 64) {
        float s = qpart[tid];
#pragma unroll
        for (int z = 1; z < 4; ++z) {
            const float w = z < S ? 1.f : 0.f;
            s += qpart[(z < S ? z : 0) * 64 + tid] * w;
        }
        s += p.bias[h * 64 + tid];
        qfin[tid] = to_f32(from_f32<TA>(s));
    }
    __syncthreads();
    SMALL_STAMP(0, 5, wave == 0);
    QSource qs;
    qs.part = nullptr; qs.bias = nullptr; qs.S = 0; qs.part_ld = 0; qs.col0 = 0; qs.append_kv = 0;
    T* out_row = (T*)p.out + (size_t)row * H * 64;
    if constexpr (!std::is_same<TKV, kv16_t>::value) {
        if (p.n_keys <= 32) {      // short histories (fixture-sized image towers): the batch path's one-round-trip wave kernel
            if (wave != 0) return;
            TA* kb = (TA*)p.kbase + p.kv_row0 * 64;
            TA* vb = (TA*)p.vbase + p.kv_row0 * 64;
            const int ng8 = (p.n_keys + 7) / 8;
            if (ng8 <= 1) decode_attention_wave_unit<TA, 1, T>(nullptr, kb, vb, nullptr, 0, p.rows_per_kv, p.kv_ld, p.n_keys, out_row, p.R, H, qs, row, h, lane, false, qfin);
            else if (ng8 <= 2) decode_attention_wave_unit<TA, 2, T>(nullptr, kb, vb, nullptr, 0, p.rows_per_kv, p.kv_ld, p.n_keys, out_row, p.R, H, qs, row, h, lane, false, qfin);
            else decode_attention_wave_unit<TA, 4, T>(nullptr, kb, vb, nullptr, 0, p.rows_per_kv, p.kv_ld, p.n_keys, out_row, p.R, H, qs, row, h, lane, false, qfin);
            return;
        }
    }
    const void* kb = KV_LDS ? (const void*)kimg : p.kbase;
    const void* vb = KV_LDS ? (const void*)vimg : p.vbase;
    const size_t rib = KV_LDS ? ri_lds : rb;
    decode_attention_scores<TKV>(kb, rib, p.n_keys, qfin, scl, lane, wave, 8);
    __syncthreads();
    SMALL_STAMP(0, 9, wave == 0);
    if (wave != 0) return;
    decode_attention_online_unit<TA, G, false, false, T, TKV, true>(nullptr, kb, vb, nullptr, 0, p.kv_ld, p.n_keys, out_row, p.R, H, qs, row,
                                                                    h, lane, 0, rib, qfin, scl);
    SMALL_STAMP(0, 6, true);
}

template <typename T, int PRO, int EPI, int NCHAIN, int NW, int RPW>
int launch_small_cfg(const SmallGemm& p, hipStream_t s) {
    constexpr int ESZ = is_g8<T> ? 4 : 2;
    constexpr int CT = NCHAIN == 4 ? 16 : 64;
    const int img_k = PRO == PRO_LN ? p.K : p.K / p.S;
    const int lds = (PRO == PRO_GLOBAL ? 0 : 16 * (img_k * ESZ + 16)) + 4 * 16 * 16 * 4;
    auto kern = dec_small_gemm_kernel<T, PRO, EPI, NCHAIN, NW, RPW>;
    if (cap_kernel_setup((const void*)kern, lds, nullptr) != 0) return -1;
    const int grid = ((p.N + CT - 1) / CT) * p.S;
    hipLaunchKernelGGL(kern, dim3(grid), dim3(NW * 64), lds, s, p);
    CAP_HIP_CHECK(hipGetLastError());
    return 0;
}

template <typename T, int PRO, int EPI, int NCHAIN, int NW>
int launch_small_rpw(const SmallGemm& p, hipStream_t s) {
    if constexpr (PRO == PRO_LN) {
        const int rpw = (p.R + NW - 1) / NW;
        if (rpw <= 1) return launch_small_cfg<T, PRO, EPI, NCHAIN, NW, 1>(p, s);
        if (rpw <= 2) return launch_small_cfg<T, PRO, EPI, NCHAIN, NW, 2>(p, s);
        return launch_small_cfg<T, PRO, EPI, NCHAIN, NW, 4>(p, s);
    } else {
        return launch_small_cfg<T, PRO, EPI, NCHAIN, NW, 1>(p, s);
    }
}

template <typename T>
int launch_small_t(const SmallGemm& p, hipStream_t s) {
    const int key = p.pro * 100 + p.epi * 10 + p.nchain;
    switch (key) {
        case PRO_LN * 100 + SEPI_PARTIAL * 10 + 4: return launch_small_rpw<T, PRO_LN, SEPI_PARTIAL, 4, 4>(p, s);        // qkv
        case PRO_GLOBAL * 100 + SEPI_PARTIAL * 10 + 4: return launch_small_rpw<T, PRO_GLOBAL, SEPI_PARTIAL, 4, 4>(p, s);  // qkv (layer 0), co, f2
        case PRO_SELFATTN * 100 + SEPI_PARTIAL * 10 + 4: return launch_small_rpw<T, PRO_SELFATTN, SEPI_PARTIAL, 4, 8>(p, s);   // so
        case PRO_LN * 100 + SEPI_ACT_T * 10 + 4: return launch_small_rpw<T, PRO_LN, SEPI_ACT_T, 4, 4>(p, s);            // f1
        case PRO_LN * 100 + SEPI_ACT_F32 * 10 + 1: return launch_small_rpw<T, PRO_LN, SEPI_ACT_F32, 1, 4>(p, s);        // transform, vocabulary
    }
    cap_set_error("launch_small_gemm: no kernel for prologue %d, epilogue %d, %d chain(s)", p.pro, p.epi, p.nchain);
    return -1;
}

}  // namespace

CAP_DEFINE_G8_CLAMP_READER(cap_g8_clamped_decode_small)

#ifdef CAP_EXPERIMENTS
extern "C" int cap_debug_small_stamps(unsigned long long* out64) {
    CAP_HIP_CHECK(hipDeviceSynchronize());
    CAP_HIP_CHECK(hipMemcpyFromSymbol(out64, HIP_SYMBOL(g_small_stamps), sizeof(unsigned long long) * 64));
    return 0;
}
#endif

int launch_small_gemm(int dtype, const SmallGemm& p, hipStream_t s) {
    const int slab = dtype == CAP_DT_BF16 ? 64 : 32;
    if (dtype != CAP_DT_G8 && dtype != CAP_DT_BF16) { cap_set_error("launch_small_gemm: G8 or bf16 operands only (dtype %d)", dtype); return -1; }
    if (p.R < 1 || p.R > SMALL_MAX_ROWS || p.N < 4 || p.N % 4 != 0 || p.S < 1 || p.K % (slab * p.S) != 0) {
        cap_set_error("launch_small_gemm: bad shape R=%d N=%d K=%d S=%d (R <= %d, N %% 4 == 0, K a multiple of %d slabs)", p.R, p.N, p.K,
                      p.S, SMALL_MAX_ROWS, p.S);
        return -1;
    }
    if (p.nchain == 4 && p.N % 16 != 0) { cap_set_error("launch_small_gemm: N = %d is not a multiple of 16", p.N); return -1; }
    if (p.pro == PRO_LN && (p.K > 1024 || p.ln.S < 1)) {
        cap_set_error("launch_small_gemm: the LayerNorm prologue takes rows of at most 1024 columns (K=%d S=%d)", p.K, p.S);
        return -1;
    }
    if (p.pro == PRO_SELFATTN && ((p.K / p.S) % 64 != 0 || p.sa.n_keys < 1 || p.sa.n_keys > 32 || p.sa.qkv_S < 1 || p.sa.qkv_S > 4)) {
        cap_set_error("launch_small_gemm: the self-attention prologue needs whole heads per K slice and 1..32 positions (K=%d S=%d keys=%d)",
                      p.K, p.S, p.sa.n_keys);
        return -1;
    }
    return dtype == CAP_DT_G8 ? launch_small_t<g8_t>(p, s) : launch_small_t<bf16_t>(p, s);
}

int launch_small_cross(int dtype, const SmallCross& p, hipStream_t s) {
    const int slab = dtype == CAP_DT_BF16 ? 64 : 32;
    const int esz = dtype == CAP_DT_BF16 ? 2 : 4;
    if (dtype != CAP_DT_G8 && dtype != CAP_DT_BF16) { cap_set_error("launch_small_cross: G8 or bf16 operands only (dtype %d)", dtype); return -1; }
    if (p.R < 1 || p.R > SMALL_MAX_ROWS || p.D != p.H * 64 || p.D > 1024 || p.S < 1 || p.S > 4 || p.D % (slab * p.S) != 0 || p.n_keys < 1 ||
        p.ln.S < 1) {
        cap_set_error("launch_small_cross: bad shape R=%d D=%d H=%d S=%d keys=%d", p.R, p.D, p.H, p.S, p.n_keys);
        return -1;
    }
    // the K/V copy: KV16 = the groups that cover the block; typed rows otherwise.  It fits when both images + the row + q stay
    // inside the 160 KiB of a CU with room to spare.
    size_t kvb;
    if (p.kv_kind == SMALL_KV_KV16) kvb = ((size_t)p.n_keys + 62) / 32 * KV16_GROUP_BYTES;
    else kvb = (size_t)p.n_keys * 64 * (p.kv_kind == SMALL_KV_BF16 ? 2 : 4);
    const size_t fixed = ((p.D * esz + 255) & ~255) + 256 * 4 + 64 * 4 + (((size_t)p.n_keys + 63) & ~(size_t)63) * 4;
    const bool fits = p.n_keys > 32 && fixed + 2 * kvb <= 150 * 1024;     // (<= 32 keys: the one-round-trip wave unit reads global memory)
    const int lds = (int)(fixed + (fits ? 2 * kvb : 0));
    const int grid = p.R * p.H;
#define CAP_SMALL_CROSS_NV(TT, TKV, GG, LDSB, NVV)                                                                      \
    do {                                                                                                                \
        auto kern = dec_small_cross_kernel<TT, TKV, GG, LDSB, NVV>;                                                     \
        if (cap_kernel_setup((const void*)kern, lds, nullptr) != 0) return -1;                                          \
        hipLaunchKernelGGL(kern, dim3(grid), dim3(512), lds, s, p);                                                     \
    } while (0)
#define CAP_SMALL_CROSS(TT, TKV, GG)                                                                                    \
    do {                                                                                                                \
        if (fits) { if (p.D <= 768) CAP_SMALL_CROSS_NV(TT, TKV, GG, true, 3); else CAP_SMALL_CROSS_NV(TT, TKV, GG, true, 4); }   \
        else { if (p.D <= 768) CAP_SMALL_CROSS_NV(TT, TKV, GG, false, 3); else CAP_SMALL_CROSS_NV(TT, TKV, GG, false, 4); }      \
    } while (0)
    // chunking (G) of the batch path's kernels for the same cache type (attention.hip::launch_decode_attention)
    if (dtype == CAP_DT_BF16 && p.kv_kind == SMALL_KV_BF16) CAP_SMALL_CROSS(bf16_t, bf16_t, 5);
    else if (dtype == CAP_DT_G8 && p.kv_kind == SMALL_KV_KV16) CAP_SMALL_CROSS(g8_t, kv16_t, 5);
    else if (dtype == CAP_DT_G8 && p.kv_kind == SMALL_KV_F32) CAP_SMALL_CROSS(g8_t, float, 7);
    else { cap_set_error("launch_small_cross: cache kind %d does not go with operand type %d", p.kv_kind, dtype); return -1; }
#undef CAP_SMALL_CROSS_NV
#undef CAP_SMALL_CROSS
    CAP_HIP_CHECK(hipGetLastError());
    return 0;
}
