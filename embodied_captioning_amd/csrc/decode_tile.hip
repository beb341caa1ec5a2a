// Fused decode kernels of the BATCH path (rows = images x beams > 16; the headline batch is 256).  A decoder layer-step of
// captioner.hip::run_decoder_step is 11 dependent launches; at a few hundred rows none of them is bound by flops or by HBM
// except the cross-attention stream, and in the engine pool's decode phase - three chains in lock step, profiles/
// r05_pool_timeline_before.txt - every launch costs its ~3 us share of the layer period whatever it does.  The small-batch
// kernels' trick (decode_small.hip: every consumer redundantly in the prologue of the kernel that needs it) does not carry over
// as it is: a LayerNorm needs whole rows and a GEMM workgroup that streamed whole rows' worth of split-K slabs for a 64-row
// tile would move five times its own operands.  What does carry over is the cross-attention block, whose cost is its K/V
// stream anyway:
//
//   dec_tile_cross_kernel   workgroup = (16-row tile, head), 16 waves, each wave:
//       LayerNorm of ONE of the tile's rows (split-K consumer of the self-attention output projection: slabs + bias + residual,
//       every load in flight at once) -> G8 / bf16 row image in LDS, fp32 rows by the head-0 workgroup;
//       one 16-column block of one K slice of the head's 64 query columns for all 16 rows on the MFMA pipe: chains j % 4 - the
//       sums of the batch path's cq GEMM (gemm_rows_kernel: S slices x 4 chains, chains then slices then bias);
//       ONE (row, head) attention unit - decode_attn.h's online unit with the chunking of the batch kernel: all 16 units of
//       the tile stream at once, a wave per unit as in the batch path's attention kernel.
//
// replaces reduce_layernorm_row_kernel + gemm_rows_kernel (cq) + decode_attention_online_kernel: three launches -> one, and the
// q partial sums, the G8 LayerNorm rows and their re-reads never leave the CU.  Same bits: every function that forms a sum is
// the one the batch kernels call (decode_frag.h, decode_attn.h, ln.h), in the same order (tests/test_tile_decode_gpu.py).
// MEASURED LEVEL with the three launches (40.1 us against 29.3 + 7.0 + 9.0 inside a 256-frame generate; docs/experiments.md,
// round 5: the prologue streams 441 KB per workgroup through the CU's fill path) - selectable (cap_set_decode_path 3), not what
// the automatic selection runs.  (First version: 8 waves, two units per wave, the next chunk in a second register buffer: 44.8 us.)
#include "gemm_tile.h"
#include "ln.h"
#include "decode_attn.h"
#include "ops.h"
#include "decode_tile.h"
#include "decode_frag.h"

namespace {

constexpr int TILE_ROWS = 16;

// NW = 16 waves: one LayerNorm row, one 16-column block of one K slice and ONE attention unit per wave - all 16 units of the tile
// stream at once, as in the batch path's attention kernel (a wave per unit) - inside 128 registers per lane.
template <typename T, typename TKV, int G, bool DB, bool NT, int NV, int NW = 16>
__global__ __launch_bounds__(NW * 64, NW / 4) void dec_tile_cross_kernel(SmallCross p) {
    static_assert(NW == 16, "one row / column block / unit per wave");
    p.W = glob(p.W); p.bias = glob(p.bias); p.kbase = glob(p.kbase); p.vbase = glob(p.vbase); p.skip = glob(p.skip); p.out = glob(p.out);
    p.ln = glob_ln(p.ln);
    constexpr int SLAB = is_g8<T> ? 32 : 64;
    constexpr int ESZ = is_g8<T> ? 4 : 2;
    constexpr int NF = 6;
    using TA = typename AttT<T>::type;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int D = p.D, H = p.H;
    const int ntiles = (p.R + TILE_ROWS - 1) / TILE_ROWS;
    // consecutive workgroups = consecutive row tiles of one head: blockIdx % 8 (the XCD, by observed placement) follows the row
    // tile, so the 12 head workgroups of a tile read its slabs through one L2
    const int rt = blockIdx.x % ntiles, h = blockIdx.x / ntiles;
    const int row0 = rt * TILE_ROWS, nrows = min(TILE_ROWS, p.R - row0);
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int r16 = lane & 15, kg = lane >> 4;

    const int pitch = D * ESZ + 16;
    char* ximg = smem;                                               // [16] LayerNorm rows as GEMM operands
    float* qpart = (float*)(smem + ((TILE_ROWS * pitch + 255) & ~255));   // [4 slices][16 rows][64]
    float* qfin = qpart + 4 * TILE_ROWS * 64;                        // [16 rows][64]

    // ---- LayerNorm of row `wave` of the tile: every load first (one round trip to the slabs)
    const SmallLN& ln = p.ln;
    {
        LnRow lv[NV];
        LnCols lk[NV];
        ln_load_row<NV>(ln, p.R, D, min(row0 + wave, p.R - 1), lane, lv);
        ln_load_cols<NV>(ln, D, lane, lk);
        if (wave < nrows)
            ln_finish_row<T, NV>(ln, p.R, D, row0 + wave, lane, lv, lk, (T*)(ximg + (size_t)wave * pitch),
                                 h == 0 && ln.x_out ? ln.x_out + (size_t)(row0 + wave) * D : nullptr);
    }
    // ---- W of this head's query columns: wave = (K slice sl, 16-column block cq), the slice's slabs
    const int S = p.S, Ks = D / S, nkb = Ks / SLAB;
    const int sl = wave & 3, cq = wave >> 2;
    const bool gw = sl < S;
    const char* wbase = (const char*)p.W + ((size_t)(h * 64 + cq * 16 + r16) * D + (size_t)sl * Ks) * ESZ;
    Frag wq[NF];
    if (gw) {
#pragma unroll
        for (int j = 0; j < NF; ++j)
            if (j < nkb) wq[j] = load_frag<T>(wbase + (size_t)j * 128, kg);
    }
    __syncthreads();

    // ---- the query columns: chain c of slice sl = slabs c, c + 4, ...; MFMA column r16 = row r16 of the tile
    if (gw) {
        f32x4 acc[4];
#pragma unroll
        for (int c = 0; c < 4; ++c) acc[c] = 0.f;
        const char* arow = ximg + (size_t)r16 * pitch + (size_t)sl * nkb * 128;
        const bool alive = r16 < nrows;
#pragma unroll
        for (int j = 0; j < NF; ++j)
            if (j < nkb) {
                const Frag af = alive ? load_frag<T>(arow + (size_t)j * 128, kg) : zero_frag();
                mma_slab<T>(acc[j & 3], wq[j], af);
            }
        for (int j = NF; j < nkb; ++j) {                      // longer slices than the register batch: one slab at a time
            const Frag af = alive ? load_frag<T>(arow + (size_t)j * 128, kg) : zero_frag();
            const Frag w = load_frag<T>(wbase + (size_t)j * 128, kg);
#pragma unroll
            for (int c = 0; c < 4; ++c)
                if ((j & 3) == c) { f32x4 t = acc[c]; mma_slab<T>(t, w, af); acc[c] = t; }
        }
        // acc[c][e] = q[row r16][h 64 + cq 16 + 4 kg + e] of chain c: chains in order, then the weight scale
        f32x4 v = acc[0];
        v += acc[1]; v += acc[2]; v += acc[3];
        if constexpr (is_g8<T>) v *= (1.0f / G8_WSCALE);
        *(f32x4*)(qpart + ((size_t)sl * TILE_ROWS + r16) * 64 + cq * 16 + 4 * kg) = v;
    }
    __syncthreads();
    // ---- q = slices in order + bias, through the attention's value type (Part8::finish): 16 x 64 values, one per thread
    {
        const int rr = tid >> 6, c = tid & 63;
        float s = qpart[(size_t)rr * 64 + c];
#pragma unroll
        for (int z = 1; z < 4; ++z) {
            const float w = z < S ? 1.f : 0.f;
            s += qpart[((size_t)(z < S ? z : 0) * TILE_ROWS + rr) * 64 + c] * w;
        }
        s += p.bias[h * 64 + c];
        qfin[(size_t)rr * 64 + c] = to_f32(from_f32<TA>(s));
    }
    // (each wave wrote the q row it reads: no barrier)

    // ---- the attention unit of row `wave` (rows of ended captions keep their context row, as in the batch kernel)
    const int rr = wave, row = row0 + rr;
    if (rr >= nrows) return;
    if (p.skip && p.skip[row]) return;
    QSource qs;
    qs.part = nullptr; qs.bias = nullptr; qs.S = 0; qs.part_ld = 0; qs.col0 = 0; qs.append_kv = 0;
    T* out_row = (T*)p.out + (size_t)row * H * 64;
    const size_t rib = p.kv_row0 + ((size_t)row * H + h) * p.kv_ld;          // one row per image (rows_per_kv == 1)
    const float* qr = qfin + (size_t)rr * 64;
    if constexpr (!std::is_same<TKV, kv16_t>::value) {
        if (p.n_keys <= 32) {      // fixture-sized image towers: the batch path's one-round-trip wave kernel
            TA* kb = (TA*)p.kbase + p.kv_row0 * 64;
            TA* vb = (TA*)p.vbase + p.kv_row0 * 64;
            const int ng8 = (p.n_keys + 7) / 8;
            if (ng8 <= 1) decode_attention_wave_unit<TA, 1, T>(nullptr, kb, vb, nullptr, 0, 1, p.kv_ld, p.n_keys, out_row, p.R, H, qs, row, h, lane, false, qr);
            else if (ng8 <= 2) decode_attention_wave_unit<TA, 2, T>(nullptr, kb, vb, nullptr, 0, 1, p.kv_ld, p.n_keys, out_row, p.R, H, qs, row, h, lane, false, qr);
            else decode_attention_wave_unit<TA, 4, T>(nullptr, kb, vb, nullptr, 0, 1, p.kv_ld, p.n_keys, out_row, p.R, H, qs, row, h, lane, false, qr);
            return;
        }
    }
    decode_attention_online_unit<TA, G, DB, NT, T, TKV, false>(nullptr, p.kbase, p.vbase, nullptr, 0, p.kv_ld, p.n_keys, out_row, p.R, H, qs,
                                                               row, h, lane, 0, rib, qr);
}

}  // namespace

CAP_DEFINE_G8_CLAMP_READER(cap_g8_clamped_decode_tile)

bool tile_cross_takes(int dtype, const SmallCross& p) {
    const int slab = dtype == CAP_DT_BF16 ? 64 : 32;
    if (dtype != CAP_DT_G8 && dtype != CAP_DT_BF16) return false;
    if (p.rows_per_kv != 1 || p.R < 1 || p.D != p.H * 64 || p.D > 768 || p.D % 16 != 0) return false;    // (1024-wide rows: 4 vectors per lane spill at 128 registers)
    if (p.S < 1 || p.S > 4 || p.D % (slab * p.S) != 0 || p.n_keys < 1 || p.ln.S < 1 || p.ln.S > 4) return false;
    if (p.kv_kind == SMALL_KV_KV16) return dtype == CAP_DT_G8 && p.n_keys > 32;
    if (p.kv_kind == SMALL_KV_F32) return dtype == CAP_DT_G8 && p.n_keys <= 32;     // fp32 rows: fixture-sized towers only (chunks of 56 keys
                                                                                     // do not fit 128 registers; cross_kv_fp32 engines keep the batch kernels)
    return p.kv_kind == SMALL_KV_BF16 && dtype == CAP_DT_BF16;
}

int launch_tile_cross(int dtype, const SmallCross& p, hipStream_t s) {
    if (!tile_cross_takes(dtype, p)) {
        cap_set_error("launch_tile_cross: shape / cache kind not taken (dtype %d R=%d D=%d H=%d S=%d keys=%d rows_per_kv=%d kind %d)", dtype, p.R,
                      p.D, p.H, p.S, p.n_keys, p.rows_per_kv, p.kv_kind);
        return -1;
    }
    const int esz = dtype == CAP_DT_BF16 ? 2 : 4;
    const int pitch = p.D * esz + 16;
    const int lds = ((TILE_ROWS * pitch + 255) & ~255) + (4 * TILE_ROWS * 64 + TILE_ROWS * 64) * 4;
    const int grid = ((p.R + TILE_ROWS - 1) / TILE_ROWS) * p.H;
#define CAP_TILE_CROSS_NV(TT, TKV, GG, DBB, NTT, NVV)                                                                   \
    do {                                                                                                                \
        auto kern = dec_tile_cross_kernel<TT, TKV, GG, DBB, NTT, NVV>;                                                  \
        if (cap_kernel_setup((const void*)kern, lds, nullptr) != 0) return -1;                                          \
        hipLaunchKernelGGL(kern, dim3(grid), dim3(1024), lds, s, p);                                                     \
    } while (0)
#define CAP_TILE_CROSS(TT, TKV, GG, DBB, NTT)                                                                           \
    do {                                                                                                                \
        CAP_TILE_CROSS_NV(TT, TKV, GG, DBB, NTT, 3);                                                                    \
    } while (0)
    // chunking (G) of the batch path's kernels for the same cache type (attention.hip::launch_decode_attention); two chunks in
    // flight per wave: the launch has 8 waves per CU where the batch kernel has 12
    if (dtype == CAP_DT_BF16) CAP_TILE_CROSS(bf16_t, bf16_t, 5, false, true);
    else if (p.kv_kind == SMALL_KV_KV16) CAP_TILE_CROSS(g8_t, kv16_t, 5, false, false);
    else CAP_TILE_CROSS(g8_t, float, 1, false, false);                 // <= 32 keys: the wave unit; the chunked unit is never entered
#undef CAP_TILE_CROSS_NV
#undef CAP_TILE_CROSS
    CAP_HIP_CHECK(hipGetLastError());
    return 0;
}
