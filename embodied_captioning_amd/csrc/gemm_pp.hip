// Split-fp16 (G8) 256x256 persistent GEMM with the two wave groups HALF A STAGE APART (gfx950 / CDNA4).
//
// gemm_big2_kernel<g8_t> (gemm.hip) runs its eight waves in phase: after the one barrier of a 32-k stage every wave issues
// its eight LDS-DMA instructions, pulls its W and first A fragments out of LDS and only then starts on its 96 MFMAs.  In-kernel
// cycle stamps on an otherwise idle chip (tools/bench_gemm_pp.py --cycles, 64 workgroups, 2.39 GHz): 4 600 cycles per stage
// against 3 072 of MFMA issue - of which 700-1 100 per wave go into ISSUING the DMA (the CU's one texture-address unit takes a
// 1 KiB DMA instruction every ~16 cycles, 64 of them per stage, and all eight waves queue up behind it right after the
// barrier while neither wave of a SIMD issues an MFMA).  Here the waves of a SIMD never start a stage together:
//   * wave grid 2 x 4 as before; group 0 = waves 0-3 = the upper BM/2 rows of the tile, group 1 = waves 4-7 = the lower half;
//     waves w and w + 4 share a SIMD and the same 64 W columns;
//   * time runs in HALF-stage steps with one barrier each; group 0 multiplies stage S in steps 2S and 2S+1, group 1 in steps
//     2S+1 and 2S+2.  At every barrier one wave of each SIMD is at a stage start (twelve fragment reads, then its DMA issue, then
//     its first MFMA) while its partner is in the middle of a stage with its operands in registers and 48 MFMAs to issue;
//   * the group that starts a stage issues the step's DMA in that start-up gap: group 0 in step 2S the upper A half, W and the
//     tile's bias row of stage S+1, group 1 in step 2S+1 the lower A half of stage S+1.  Each group confirms ITS OWN pieces
//     (vmcnt(0)) right before the barrier at which it starts the stage they belong to - one whole stage time after issuing them;
//   * A fragments are double buffered in registers (block i+1 is read under block i's 12 MFMAs); fragment reads and their
//     counted lgkmcnt waits are asm statements (hipcc waits lgkmcnt(0), i.e. also for the block it has just requested);
//   * epilogue through two LDS strips per wave slot, whole 128-byte lines per store (pp_epilogue_body); with a residual operand
//     (fp32 output: C = acc + bias + resid, resid may be C) the ViT branch GEMMs add into the residual stream in place.
// Hazards (buffers S & 1):
//   RAW  upper A / W of stage S+1: issued by group 0 in step 2S, confirmed by group 0 before the barrier of step 2S+2 (their first
//        reader; group 1 reads W in step 2S+3).  Lower A of stage S+1: issued by group 1 in step 2S+1, confirmed before the barrier
//        of step 2S+3, its first reader.
//   WAR  the DMA of step 2S (stage S+1) overwrites upper A / W of stage S-1: group 0 read them in steps 2S-2, 2S-1, group 1 read its
//        W fragments at the start of step 2S-1 - all retired (consumed by MFMAs) before the barrier of step 2S.  The DMA of step
//        2S+1 overwrites lower A of stage S-1, read by group 1 in steps 2S-1, 2S.
// Sums are formed exactly as in every other G8 kernel (three MFMAs per product, k ascending): results are bit-identical to
// gemm_big2_kernel's and to the register-staged tiles' (tests/test_split_gpu.py).
// Measured (MI355X, 256 frames of ViT-B/16, same process, us per launch against gemm_big2_kernel): qkv 435 / 476, proj 157 /
// 178, fc1 640 / 690, fc2 551 / 605 (-7 .. -11 %).  In cycles (64 workgroups, clock at its 2.39 GHz ceiling): 3 710-3 750 per
// stage against 4 600, epilogue 3 900 (fp32 out: the L1's store rate) / 7 200 (G8 out: VALU) per tile and group against 14 800
// for all eight waves together.  On the whole chip the shader clock settles ~10 % lower than under gemm_big2_kernel (1.93
// against 2.13 GHz: the board's power limit), so about half of the gain in cycles arrives as time.
// Tried on this structure and not kept (tools/bench_gemm_pp.py --cycles, gpurun_out/r3_pp1[1-6]c.log): the epilogue spread over
// the tile boundary, one row block (or one strip piece, pipelined) after each MFMA block - a wave issues a block's 12 MFMAs
// before anything else, so its epilogue arithmetic still runs while ITS matrix instructions wait for the pipe, the steps at a
// tile boundary got longer than the serial epilogue they replaced (boundary 10.6k against 8.5k cycles per tile, and the stage
// loop itself 12 % slower); A fragments two blocks ahead instead of one (level); three accumulator chains per block in a row
// instead of two interleaved (level); the W rows of a column group staged half by each of the two waves that read them (9 + 8 DMA
// instructions per wave and stage instead of 13 + 4, group 1's half two stages ahead, right after its own W fragments landed):
// correct, 3 940-3 980 cycles per stage against 3 710-3 750 - the second issue burst sits in front of MFMAs that were ready.
// Round 6: group 0's epilogue deferred past the next barrier (its strips in the W rows of the stage buffer it had just finished), so
// that the two groups' epilogues run side by side instead of one after the other: bit-identical, and LEVEL (tools/ab_gemm_libs.py,
// same process: qkv 420.3 / 419.5, fc1 620-634 / 616-626, proj 152-156 / 152-161 us) - the epilogue is bound by what the two waves
// of a SIMD share (VALU issue, the CU's store path), so two at once take twice as long each.  Not kept.  `s_setprio 1` around every
// 12-MFMA block (0 after it), same-process A/B on six shapes: qkv 423.1 / 423.6, fc1 623.5 / 626.4, proj 151.9 / 152.9, fc2 532.9 /
// 534.8, qkv and fc1 at 1 024 frames 1 688 / 1 687 and 2 442 / 2 448 us: level.  Not kept.
#include <stdlib.h>

#include <algorithm>

#include "gemm_tile.h"

namespace {

template <int N> __device__ __forceinline__ void wait_vm() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }

// Fragment reads and their waits are asm statements: hipcc waits with lgkmcnt(0) for the fragments it reads itself, i.e. also
// for the block it has just requested.  Here block i + 1's two reads stay in flight while block i is multiplied (LDS returns
// in order: lgkmcnt(2)).  The wait "modifies" the fragments it releases, so no MFMA that uses them can be placed before it.
template <int OFF, typename V> __device__ __forceinline__ void ds_read16(V& d, unsigned addr) {
    asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(d) : "v"(addr), "n"(OFF));
}
template <int N, typename V> __device__ __forceinline__ void wait_lgkm(V& a, V& b) {
    asm volatile("s_waitcnt lgkmcnt(%2)" : "+v"(a), "+v"(b) : "n"(N));
}
template <int N, typename V> __device__ __forceinline__ void wait_lgkm(V& a, V& b, V& c, V& d) {
    asm volatile("s_waitcnt lgkmcnt(%4)" : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : "n"(N));
}
template <int I, int N, typename F> __device__ __forceinline__ void static_for(F&& f) {
    if constexpr (I < N) {
        f(std::integral_constant<int, I>{});
        static_for<I + 1, N>(f);
    }
}

// Epilogue.  acc[i][j][e] = C[row0 + 16 i + r16][col0 + 16 j + 4 kg + e]: consecutive LANES hold consecutive ROWS, and a store
// instruction costs one L1 line access per lane group that is contiguous in memory - storing straight from this layout (16
// bytes per lane, 64 scattered pieces per instruction) measured ~270 cycles per instruction, 8 500-9 500 cycles per tile and
// group whatever the arithmetic in front of it (tools/bench_gemm_pp.py --cycles).  So the 16 x 32-column piece of a row block goes
// through an LDS strip (16 rows x 128 payload bytes, pitch 144: conflict-free both ways) and leaves as 2 x 8 whole 128-byte
// lines, as in big2_epilogue - but here one wave per SIMD runs the epilogue while its partner waits at a barrier, so every
// cycle of it is matrix-pipe idle time and the instruction count per value matters:
//   * two strips per wave slot, software pipelined: piece n + 1 is computed and written while piece n is read back and stored
//     (waves w and w + 4 run their epilogues in different steps and share a slot);
//   * ACT is a template parameter (0 none, 1 GELU, 2 ReLU; -1 = read p.gelu per piece, edge tiles only), interior tiles
//     (FULL) store unguarded through two running pointers, clamped groups are counted in a register and added to the
//     translation unit's counter once per tile;
//   * a G8 output piece is the strip row image [8 hi | 8 lo] per 8 columns: two 8-byte writes per four values.
template <typename T, bool OUT_F32, int EPI, int MI, int NI, int ACT, bool FULL, bool RESID>
__device__ __forceinline__ void pp_epilogue_body(const GemmParams& p, const f32x4 (&acc)[MI][NI], const f32x4 (&biasv)[NI], char* strip2,
                                                 int row0, int col0, int lane) {
    constexpr bool G8 = is_g8<T>;
    constexpr bool F32OUT = OUT_F32 || EPI == EPI_PARTIAL || EPI == EPI_PATCH;
    static_assert(F32OUT || !G8 || EPI == EPI_STORE, "G8 output exists for plain row-major stores only");
    static_assert(NI == 4, "two 32-column pieces per row block");
    static_assert(!RESID || (OUT_F32 && EPI == EPI_STORE), "the residual operand exists for fp32 row-major stores only");
    if constexpr (!G8 && !F32OUT) {
        // bf16 output: a strip row is the row block's whole 64 columns (128 bytes): one piece per row block
        constexpr int SPITCH = 144, SBYTES = 16 * SPITCH;
        const int r16 = lane & 15, kg = lane >> 4, srow = lane >> 3, spiece = lane & 7;
        const int act = ACT >= 0 ? ACT : p.gelu;
        auto compute = [&](int i) __attribute__((always_inline)) {
            char* st = strip2 + (i & 1) * SBYTES + r16 * SPITCH + kg * 8;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                f32x4 v = acc[i][j];
                if (EPI != EPI_PARTIAL) v += biasv[j];
                if (act == 1) {
#pragma unroll
                    for (int e = 0; e < 4; ++e) v[e] = gelu_for<T>(v[e]);
                } else if (act == 2) {
#pragma unroll
                    for (int e = 0; e < 4; ++e) v[e] = fmaxf(v[e], 0.f);
                }
                bf16x4 w;
#pragma unroll
                for (int e = 0; e < 4; ++e) w[e] = (bf16_t)v[e];
                *(bf16x4*)(st + j * 32) = w;             // columns 16 j + 4 kg .. + 3
            }
        };
        auto drain = [&](int i) __attribute__((always_inline)) {
            const char* st = strip2 + (i & 1) * SBYTES + srow * SPITCH + spiece * 16;
            const u32x4 r0 = *(const u32x4*)st, r1 = *(const u32x4*)(st + 8 * SPITCH);
            const int col = col0 + spiece * 8, rowa = row0 + i * 16 + srow, rowb = rowa + 8;
            if (FULL || (col < p.N && rowa < p.M)) epi_store_raw<T, EPI>(p, rowa, col, r0, FULL || col + 8 <= p.N);
            if (FULL || (col < p.N && rowb < p.M)) epi_store_raw<T, EPI>(p, rowb, col, r1, FULL || col + 8 <= p.N);
        };
        compute(0);
#pragma unroll
        for (int i = 0; i < MI; ++i) {
            if (i + 1 < MI) compute(i + 1);
            drain(i);
        }
        return;
    }
    constexpr int NP = MI * 2;                           // pieces: (i, jp)
    constexpr int SPITCH = 144, SBYTES = 16 * SPITCH;
    const int r16 = lane & 15, kg = lane >> 4, srow = lane >> 3, spiece = lane & 7;
    const int act = ACT >= 0 ? ACT : p.gelu;
    unsigned sat = 0;
    auto compute = [&](int n) __attribute__((always_inline)) {           // piece n -> strip n & 1
        const int i = n >> 1, jp = n & 1;
        char* st = strip2 + (n & 1) * SBYTES + r16 * SPITCH;
#pragma unroll
        for (int jj = 0; jj < 2; ++jj) {
            const int j = jp * 2 + jj;
            f32x4 v = acc[i][j];
            if constexpr (G8) v *= (1.0f / G8_WSCALE);
            if (EPI != EPI_PARTIAL) v += biasv[j];
            if (act == 1) {
#pragma unroll
                for (int e = 0; e < 4; ++e) v[e] = gelu_for<T>(v[e]);
            } else if (act == 2) {
#pragma unroll
                for (int e = 0; e < 4; ++e) v[e] = fmaxf(v[e], 0.f);
            }
            if constexpr (F32OUT) {
                *(f32x4*)(st + (jj * 16 + 4 * kg) * 4) = v;
            } else {
                const bool ok = FULL || (row0 + i * 16 + r16 < p.M && col0 + j * 16 + 4 * kg < p.N);
                sat += (ok && !(fmaxf(fmaxf(fabsf(v[0]), fabsf(v[1])), fmaxf(fabsf(v[2]), fabsf(v[3]))) <= G8_AMAX)) ? 1u : 0u;
                f16x4 hi, lo;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    f16_t h, l;
                    g8_split(v[e], h, l);
                    hi[e] = h; lo[e] = l;
                }
                // columns c = jj * 16 + 4 kg .. + 3 of the 32-column piece: group c >> 3 (32 bytes), hi halves at 2 (c & 7).  An 8-byte
                // LDS store is served in groups of 16 consecutive lanes over 32 banks: the 16 rows of a (kg, half) at pitch 144 are
                // 4 r mod 32 - rows r and r + 8 on the same banks (2-way on every store: SQ_LDS_BANK_CONFLICT 12 % of this kernel's
                // LDS cycles, round 5).  Rows 8-15 therefore swap the two 8-byte halves of each 16-byte chunk (the drain swaps back).
                char* gp = st + ((((jj * 16 + 4 * kg) >> 3) * 32 + ((4 * kg) & 7) * 2) ^ (r16 & 8));
                *(f16x4*)gp = hi;
                *(f16x4*)(gp + 16) = lo;
            }
        }
    };
    // read side: lane (srow, spiece) moves 16 bytes of strip row rr * 8 + srow: 4 columns at col0 + jp * 32 + spiece * 4
    char* ptr0 = nullptr;
    char* ptr1 = nullptr;
    size_t step = 0;
    if constexpr (EPI == EPI_STORE) {
        ptr0 = (char*)p.C + ((size_t)(row0 + srow) * p.ldc + col0 + spiece * 4) * 4;
        ptr1 = ptr0 + (size_t)p.ldc * 32;
        step = (size_t)p.ldc * 64;
    }
    // RESID: C = f(acc + bias) + resid (resid may be C itself: the ViT branch GEMMs add into the residual stream in place).  The
    // two 16-byte pieces of residual a lane needs for piece n + 1 are requested before piece n is drained (one piece of memory
    // latency under a piece of epilogue); a lane loads exactly the addresses it stores, load first: aliasing is safe.
    // (The loads cost 20-25 us per launch: every workgroup wants its block at the same moment.  Touching the block ahead of time
    // with 4-byte LDS-DMA into a sink - all of it two stages early, or a quarter every other stage over the last nine - measured
    // the same or worse, tools/bench_branch_add.py.  What the waits say: vmcnt counts loads and stores in one in-order queue, so
    // the wait for piece n's residual is also a wait for the stores of piece n - 2; requesting two or more pieces ahead - also with
    // the look-ahead growing as accumulator registers fall free - pushes the kernel from 256 registers into scratch INSIDE the stage
    // loop (the DMA offsets go first): fc2 557 -> 823 us.)
    const char* rp0 = nullptr;
    const char* rp1 = nullptr;
    size_t rstep = 0;
    f32x4 q0[2], q1[2];
    if constexpr (RESID) {
        rp0 = (const char*)p.resid + ((size_t)(row0 + srow) * p.ldr + col0 + spiece * 4) * 4;
        rp1 = rp0 + (size_t)p.ldr * 32;
        rstep = (size_t)p.ldr * 64;
    }
    auto fetch = [&](int n) __attribute__((always_inline)) {
        if constexpr (RESID) {
            const int i = n >> 1, jp = n & 1;
            const int col = col0 + jp * 32 + spiece * 4;
            const int rowa = row0 + i * 16 + srow, rowb = rowa + 8;
            const bool cok = FULL || col < p.N;
            q0[n & 1] = 0.f; q1[n & 1] = 0.f;
            if (FULL || (cok && rowa < p.M)) q0[n & 1] = *(const f32x4*)(rp0 + jp * 128);
            if (FULL || (cok && rowb < p.M)) q1[n & 1] = *(const f32x4*)(rp1 + jp * 128);
            if (jp == 1) { rp0 += rstep; rp1 += rstep; }
        }
    };
    auto drain = [&](int n) __attribute__((always_inline)) {
        const int i = n >> 1, jp = n & 1;
        const char* st = strip2 + (n & 1) * SBYTES + srow * SPITCH + spiece * 16;
        u32x4 r0 = *(const u32x4*)st, r1 = *(const u32x4*)(st + 8 * SPITCH);
        if constexpr (!F32OUT) r1 = u32x4{r1[2], r1[3], r1[0], r1[1]};     // rows 8-15 of a G8 strip: halves swapped by the writer
        if constexpr (RESID) {
            r0 = __builtin_bit_cast(u32x4, __builtin_bit_cast(f32x4, r0) + q0[n & 1]);
            r1 = __builtin_bit_cast(u32x4, __builtin_bit_cast(f32x4, r1) + q1[n & 1]);
        }
        const int col = col0 + jp * 32 + spiece * 4;
        const int rowa = row0 + i * 16 + srow, rowb = rowa + 8;
        const bool cok = FULL || col < p.N;
        if constexpr (EPI == EPI_STORE) {
            if (FULL || (cok && rowa < p.M)) *(u32x4*)(ptr0 + jp * 128) = r0;
            if (FULL || (cok && rowb < p.M)) *(u32x4*)(ptr1 + jp * 128) = r1;
            if (jp == 1) { ptr0 += step; ptr1 += step; }
        } else {
            if (FULL || (cok && rowa < p.M)) epi_store_f32<EPI>(p, rowa, col, __builtin_bit_cast(f32x4, r0));
            if (FULL || (cok && rowb < p.M)) epi_store_f32<EPI>(p, rowb, col, __builtin_bit_cast(f32x4, r1));
        }
    };
    fetch(0);
    compute(0);
#pragma unroll
    for (int n = 0; n < NP; ++n) {
        if (n + 1 < NP) {
            fetch(n + 1);
            compute(n + 1);
        }
        drain(n);
    }
    if constexpr (!F32OUT) {
        if (sat) atomicAdd(&g_g8_clamped, sat);          // same count as g8_note_range per stored group of four
    }
}

// EPI_CROSSKV into a KV16 cache (common.h): a wave's 64 columns are one head of one (layer, k | v) block, so a row block's 16 x 64
// values are 16 head rows: row maximum (16 values in the lane, then across the four lanes of the row), one scale per row,
// int16 quantisation, and the 128-byte rows leave through the strips like every other output (two whole lines per lane pair of
// reads).  Row r of a block = (image * heads + head) * tokens + token; rows live in groups of 32 (kv16_row_off / kv16_scale_off).
template <int MI>
__device__ __forceinline__ void pp_epilogue_kv16(const GemmParams& p, const f32x4 (&acc)[MI][4], const f32x4 (&biasv)[4], char* strip2,
                                                 int row0, int col0, int lane) {
    constexpr int SPITCH = 144, SBYTES = 16 * SPITCH;
    if (col0 >= p.N) return;                             // (N % 64 == 0: a wave's columns are inside or outside together)
    const int r16 = lane & 15, kg = lane >> 4, srow = lane >> 3, spiece = lane & 7;
    const int NT = p.p0, H = p.p1, Dh = H * 64;
    const int blk = col0 / Dh, h = (col0 - blk * Dh) >> 6;
    char* base = (char*)p.C + (size_t)blk * kv16_block_bytes((size_t)p.p2 * H * NT);
    // row -> (image, token) without a division per lane: rows of this block are row0 + x, x < 16 MI; n / NT for n < 2^16
    const unsigned magic = 0xFFFFFFFFu / (unsigned)NT + 1u;
    const int b0 = row0 / NT, t0 = row0 - b0 * NT;
    auto block_row = [&](int x) -> size_t {
        const unsigned n = (unsigned)(t0 + x), qd = __umulhi(n, magic);
        return ((size_t)(b0 + (int)qd) * H + h) * NT + (n - qd * (unsigned)NT);
    };
    auto compute = [&](int i) __attribute__((always_inline)) {
        f32x4 v[4];
        float am = 0.f;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            v[j] = acc[i][j] * (1.0f / G8_WSCALE) + biasv[j];
            am = fmaxf(am, fmaxf(fmaxf(fabsf(v[j][0]), fabsf(v[j][1])), fmaxf(fabsf(v[j][2]), fabsf(v[j][3]))));
        }
        am = fmaxf(am, __shfl_xor(am, 16, 64));
        am = fmaxf(am, __shfl_xor(am, 32, 64));
        float sc, inv;
        kv16_scales(am, sc, inv);
        char* st = strip2 + (i & 1) * SBYTES + r16 * SPITCH + kg * 8;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            uint2 w;
            w.x = kv16_pack2(v[j][0], v[j][1], inv);
            w.y = kv16_pack2(v[j][2], v[j][3], inv);
            *(uint2*)(st + j * 32) = w;                  // dims 16 j + 4 kg .. + 3
        }
        if (kg == 0 && row0 + i * 16 + r16 < p.M) *(float*)(base + kv16_scale_off(block_row(i * 16 + r16))) = sc;
    };
    auto drain = [&](int i) __attribute__((always_inline)) {
        const char* st = strip2 + (i & 1) * SBYTES + srow * SPITCH + spiece * 16;
        const u32x4 q0 = *(const u32x4*)st, q1 = *(const u32x4*)(st + 8 * SPITCH);
        const int xa = i * 16 + srow, xb = xa + 8;
        if (row0 + xa < p.M) *(u32x4*)(base + kv16_row_off(block_row(xa)) + spiece * 16) = q0;
        if (row0 + xb < p.M) *(u32x4*)(base + kv16_row_off(block_row(xb)) + spiece * 16) = q1;
    };
    compute(0);
#pragma unroll
    for (int i = 0; i < MI; ++i) {
        if (i + 1 < MI) compute(i + 1);
        drain(i);
    }
}

template <typename T, bool OUT_F32, int EPI, int MI, int NI, bool RESID>
__device__ __forceinline__ void pp_epilogue(const GemmParams& p, const f32x4 (&acc)[MI][NI], const char* bias_w, char* strip2, int row0,
                                            int col0, int lane) {
    const int kg = lane >> 4;
    f32x4 biasv[NI];
#pragma unroll
    for (int j = 0; j < NI; ++j) biasv[j] = bias_w ? *(const f32x4*)(bias_w + (j * 16 + 4 * kg) * 4) : f32x4(0.f);
    if constexpr (EPI == EPI_CROSSKV && is_g8<T>) {
        if (p.kv16) {
            pp_epilogue_kv16<MI>(p, acc, biasv, strip2, row0, col0, lane);
            return;
        }
    }
    const int act = EPI != EPI_PARTIAL ? p.gelu : 0;
    if (row0 + MI * 16 <= p.M && col0 + NI * 16 <= p.N) {
        if (act == 0) pp_epilogue_body<T, OUT_F32, EPI, MI, NI, 0, true, RESID>(p, acc, biasv, strip2, row0, col0, lane);
        else if (act == 1) pp_epilogue_body<T, OUT_F32, EPI, MI, NI, 1, true, RESID>(p, acc, biasv, strip2, row0, col0, lane);
        else pp_epilogue_body<T, OUT_F32, EPI, MI, NI, 2, true, RESID>(p, acc, biasv, strip2, row0, col0, lane);
    } else {
        pp_epilogue_body<T, OUT_F32, EPI, MI, NI, -1, false, RESID>(p, acc, biasv, strip2, row0, col0, lane);
    }
}

// ABL (-DCAP_EXPERIMENTS builds, CAP_EXP_ABLATE): what a launch costs WITHOUT parts of the kernel - bit 0 = no DMA after the second
// stage, bit 1 = no fragment reads after the first stage, bit 2 = no epilogue (careful: the compiler then deletes the MFMAs too), bit 3 = the
// DMA of every stage re-reads the first stage's addresses (issue and LDS-write cost stay, the traffic beyond L2 goes); results are garbage
template <typename T, bool OUT_F32, int EPI, bool PROF, int BM = 256, bool RESID = false, int ABL = 0>
__global__ __launch_bounds__(512, 2) void gemm_pp_kernel(GemmParams p) {
    // T = g8_t: 32 k values per 128-byte stage row, chunks [H0 L0 H1 L1 ..]: fragment 0 / 1 = the hi / lo chunk of k-group kg,
    // three MFMAs per product.  T = bf16_t: 64 k values per row, fragment 0 / 1 = k-step 0 / 1 (chunks kg / 4 + kg), one MFMA each.
    using vec = typename Mma<T>::vec;
    constexpr bool G8 = is_g8<T>;
    constexpr int ESZ = G8 ? 4 : 2, EPC = Mma<T>::EPC;   // bytes per element; elements per 16-byte chunk
    constexpr int BN = 256, WM = BM / 2, MI = WM / 16, MH = MI / 2, NI = 4;
    constexpr int SUB = 256 / BM;                        // work items per 256-row tile
    constexpr int STAGE = (BM + BN) * 128;               // 64 KiB (48 KiB for half tiles)
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* const bias_rows = smem + 2 * STAGE;            // [2 slots][4 column groups][64 floats]
    char* const strips = bias_rows + 2048;               // [4 wave slots][2 strips][16 rows x 144 B]

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int g = wave >> 2, wn0 = (wave & 3) * 64, wm0 = g * WM;
    const int r16 = lane & 15, kg = lane >> 4;
    const int ntn = (p.N + BN - 1) / BN;
    const int tile0 = p.tile1 > 0 ? p.tile0 : 0;
    const int nitems = ((p.tile1 > 0 ? p.tile1 : ((p.M + 255) / 256) * ntn) - tile0) * SUB;
    const int nk = p.K / (8 * EPC);
    // Tile t -> (row tile tm, column tile tn).  The 32 workgroups of an XCD work on 32 consecutive tiles at a time and share what
    // they fetch through that XCD's L2.  Row-major numbering makes those 32 tiles one strip of 32 / ntn rows x ntn columns - fine
    // up to ntn ~ 12 (qkv: 3.6 rows + 9 column tiles of operands per 32 output tiles), but for the cross-K/V GEMM (ntn = 72) it is
    // half a row: ONE A tile and 32 different W tiles, every work item pulls its own 0.8 MB of W through the fabric (11.8 GB per
    // launch by FETCH_SIZE against 0.2 GB of operands).  Wide problems are therefore numbered in bands of 4 tile rows, column by
    // column inside a band: 32 consecutive tiles = 4 rows x 8 columns, 12 operand tiles instead of 33.
    const int ntm_all = (p.M + 255) / 256;
    const bool banded = ntn > 16;
    auto tile_coords = [&](int t, int& tm, int& tn) __attribute__((always_inline)) {
        if (!banded) {
            tm = t / ntn;
            tn = t - tm * ntn;
        } else {
            const int band = t / (4 * ntn), r = t - band * 4 * ntn;
            const int h = min(4, ntm_all - band * 4);    // rows of this band (the last one may be short)
            tn = r / h;
            tm = band * 4 + (r - tn * h);
        }
    };

    // XCD-aware walk, as in gemm_big2_kernel: XCD x owns the contiguous run [c0, c1) of work items
    const int xcd = blockIdx.x & 7, li = blockIdx.x >> 3;
    const int nl = ((int)gridDim.x >> 3) + (xcd < ((int)gridDim.x & 7) ? 1 : 0);
    const int tq = nitems >> 3, tr = nitems & 7;
    const int c0 = xcd < tr ? xcd * (tq + 1) : tr * (tq + 1) + (xcd - tr) * tq;
    const int c1 = c0 + tq + (xcd < tr ? 1 : 0);
    const int first = c0 + li;
    if (first >= c1) return;
    const int ntl = (c1 - first + nl - 1) / nl;          // items of this workgroup
    const int NS = ntl * nk;                             // stages of this workgroup

    // ---- issue side.  The group that STARTS a stage in a step issues that step's DMA, between its twelve fragment reads and its
    // first MFMA - where its issue slot is idle anyway, while the partner wave on its SIMD (mid-stage, operands in registers) has
    // the matrix pipe to itself.  (All eight waves issuing right after the barrier, as gemm_big2_kernel does, costs every wave
    // 700-1 100 cycles per stage in which neither wave of a SIMD issues an MFMA: the CU's one texture-address unit takes a 1 KiB
    // DMA instruction every ~16 cycles, 64 of them per stage - tools/bench_gemm_pp.py --cycles.)
    //   group 0, step 2S   : slot a of stage S + 1 = the upper A half (PQ pieces per wave), W (8 per wave), the tile's bias row
    //   group 1, step 2S+1 : slot b of stage S + 1 = the lower A half (PQ pieces per wave)
    constexpr int PQ = WM / 32;                          // 8-row pieces of one A half per wave of a group
    const int wq = wave & 3;
    const int prow = lane >> 3, ppos = lane & 7;
    unsigned oa[PQ], ow[8], obias = 0;                   // byte offsets (k = 0) of this lane's pieces in A / W / bias
    const bool has_bias = EPI != EPI_PARTIAL && p.bias != nullptr;
    auto set_ptrs = [&](int x) __attribute__((always_inline)) {
        const int item = first + x * nl;
        const int t = tile0 + item / SUB, sub = item % SUB;
        int tm, tn;
        tile_coords(t, tm, tn);
#pragma unroll
        for (int j = 0; j < PQ; ++j) {
            const int row = (wq * PQ + j) * 8 + prow;    // row inside this group's half; WM % 16 == 0: both halves swizzle alike
            const int gch = ppos ^ swz_key<G8>(row);
            const int ga = min(tm * 256 + sub * BM + wm0 + row, p.M - 1);
            oa[j] = ((unsigned)ga * (unsigned)p.lda + gch * EPC) * (unsigned)ESZ;
        }
        if (g == 0) {
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const int row = (wq * 8 + j) * 8 + prow;
                const int gch = ppos ^ swz_key<G8>(row);
                const int gw = min(tn * BN + row, p.N - 1);
                ow[j] = ((unsigned)gw * (unsigned)p.ldw + gch * EPC) * (unsigned)ESZ;
            }
            obias = has_bias ? (unsigned)min(tn * BN + wn0 + lane, p.N - 1) * 4u : 0u;
        }
    };
    int i_kt = 0, i_x = 0;
    auto issue = [&](int s) __attribute__((always_inline)) {                            // this group's slot of stage s; then on to the next stage
        char* st = smem + (s & 1) * STAGE;
        const char* ab = (const char*)p.A + (size_t)((ABL & 8) ? 0 : i_kt) * 128;        // (ABL bit 3: every stage re-reads k = 0 of the first tile: L2 hits)
#pragma unroll
        for (int j = 0; j < PQ; ++j)
            __builtin_amdgcn_global_load_lds(CAP_GPTR(ab + oa[j]), CAP_LPTR(st + (wm0 + (wq * PQ + j) * 8) * 128), 16, 0, 0);
        if (g == 0) {
            const char* wb = (const char*)p.W + (size_t)((ABL & 8) ? 0 : i_kt) * 128;
#pragma unroll
            for (int j = 0; j < 8; ++j)
                __builtin_amdgcn_global_load_lds(CAP_GPTR(wb + ow[j]), CAP_LPTR(st + BM * 128 + (wq * 8 + j) * 1024), 16, 0, 0);
            if (has_bias && i_kt == 0)                   // waves w and w + 4 share the 64 columns and the row
                __builtin_amdgcn_global_load_lds(CAP_GPTR((const char*)p.bias + obias), CAP_LPTR(bias_rows + (i_x & 1) * 1024 + wq * 256), 4, 0, 0);
        }
        if (++i_kt == nk) {
            i_kt = 0;
            if (++i_x < ntl && !(ABL & 8)) set_ptrs(i_x);
        }
    };

    // ---- compute side
    f32x4 acc[MI][NI];
#pragma unroll
    for (int i = 0; i < MI; ++i)
#pragma unroll
        for (int j = 0; j < NI; ++j) acc[i][j] = 0.f;
    vec bh[NI], bl[NI], ah[2], al[2];
    const unsigned lds0 = (unsigned)(size_t)CAP_LPTR(smem);
    // fragment 0 / 1 of a row (names: "h" / "l" after the G8 case); + 128 * (16-aligned row): same swizzle
    const unsigned f_hi = swz_off<G8>(r16, G8 ? 2 * kg : kg), f_lo = swz_off<G8>(r16, G8 ? 2 * kg + 1 : 4 + kg);
    // G8: a_hi.w_lo, a_lo.w_hi, a_hi.w_hi per accumulator; bf16: k-step 0, k-step 1 - the order of every kernel of the type.  The
    // chains of two accumulators are interleaved, so that a wave that has the matrix pipe to itself never issues an MFMA that
    // waits for the one before it
    auto mma_pair = [&](f32x4& c0, f32x4& c1, int j, int sl) __attribute__((always_inline)) {
        if constexpr (G8) {
            c0 = __builtin_amdgcn_mfma_f32_16x16x32_f16(bl[j], ah[sl], c0, 0, 0, 0);
            c1 = __builtin_amdgcn_mfma_f32_16x16x32_f16(bl[j + 1], ah[sl], c1, 0, 0, 0);
            c0 = __builtin_amdgcn_mfma_f32_16x16x32_f16(bh[j], al[sl], c0, 0, 0, 0);
            c1 = __builtin_amdgcn_mfma_f32_16x16x32_f16(bh[j + 1], al[sl], c1, 0, 0, 0);
            c0 = __builtin_amdgcn_mfma_f32_16x16x32_f16(bh[j], ah[sl], c0, 0, 0, 0);
            c1 = __builtin_amdgcn_mfma_f32_16x16x32_f16(bh[j + 1], ah[sl], c1, 0, 0, 0);
        } else {
            c0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bh[j], ah[sl], c0, 0, 0, 0);
            c1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bh[j + 1], ah[sl], c1, 0, 0, 0);
            c0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bl[j], al[sl], c0, 0, 0, 0);
            c1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bl[j + 1], al[sl], c1, 0, 0, 0);
        }
        __builtin_amdgcn_sched_barrier(0);
    };
    auto mma_block = [&](int i) __attribute__((always_inline)) {
        mma_pair(acc[i][0], acc[i][1], 0, i & 1);
        mma_pair(acc[i][2], acc[i][3], 2, i & 1);
    };

    long long prof_bar = 0, prof_epi = 0, prof_vm = 0, prof_iss = 0, prof_ew = 0;
    const long long prof_c0 = PROF ? clock64() : 0, prof_w0 = PROF ? wall_clock64() : 0;

    // A step = barrier + half a stage of MFMAs per group.  Before the barrier at which a group starts stage S it confirms that
    // the slot IT issued for stage S (one stage time ago) has landed: the barrier then makes it visible to the other group too.
    bool confirmed = false;                              // the wait already happened (before this wave's epilogue stores)
    auto sync = [&](bool starts) __attribute__((always_inline)) {
        const long long t0 = PROF ? clock64() : 0;
        if (starts && !confirmed) wait_vm<0>();
        confirmed = false;
        const long long t1 = PROF ? clock64() : 0;
        CAP_RAW_BARRIER();
        if constexpr (PROF) { prof_vm += t1 - t0; prof_bar += clock64() - t1; }
    };
    int c_kt = 0, c_x = 0;
    auto half0 = [&](int S) __attribute__((always_inline)) {
        const unsigned sa = lds0 + (S & 1) * STAGE;
        const unsigned aH = sa + wm0 * 128 + f_hi, aL = sa + wm0 * 128 + f_lo;
        const unsigned bH = sa + (BM + wn0) * 128 + f_hi, bL = sa + (BM + wn0) * 128 + f_lo;
        const bool rd = !((ABL & 2) && S >= 1);
        // 12 reads; the first product needs the first four of them
        if (rd) {
            ds_read16<0>(bl[0], bL); ds_read16<0>(ah[0], aH); ds_read16<0>(bh[0], bH); ds_read16<0>(al[0], aL);
            ds_read16<2048>(bl[1], bL); ds_read16<2048>(bh[1], bH);
            ds_read16<4096>(bl[2], bL); ds_read16<4096>(bh[2], bH);
            ds_read16<6144>(bl[3], bL); ds_read16<6144>(bh[3], bH);
            ds_read16<2048>(ah[1], aH); ds_read16<2048>(al[1], aL);
        }
        __builtin_amdgcn_sched_barrier(0);
        {
            const long long t2 = PROF ? clock64() : 0;
            if (S + 1 < NS && !((ABL & 1) && S >= 1)) issue(S + 1);
            if constexpr (PROF) prof_iss += clock64() - t2;
        }
        __builtin_amdgcn_sched_barrier(0);
        wait_lgkm<6>(bl[0], ah[0], bh[0], al[0]); wait_lgkm<6>(bl[1], bh[1]);
        mma_pair(acc[0][0], acc[0][1], 0, 0);
        wait_lgkm<2>(bl[2], bh[2]); wait_lgkm<2>(bl[3], bh[3]);
        mma_pair(acc[0][2], acc[0][3], 2, 0);
        static_for<1, MH>([&](auto ic) {
            constexpr int i = decltype(ic)::value;       // i + 1 <= MH < MI: the first block of the second half included
            if (rd) { ds_read16<(i + 1) * 2048>(ah[(i + 1) & 1], aH); ds_read16<(i + 1) * 2048>(al[(i + 1) & 1], aL); }
            wait_lgkm<2>(ah[i & 1], al[i & 1]);
            mma_block(i);
        });
    };
    auto half1 = [&](int S) __attribute__((always_inline)) {
        const unsigned sa = lds0 + (S & 1) * STAGE;
        const unsigned aH = sa + wm0 * 128 + f_hi, aL = sa + wm0 * 128 + f_lo;
        const bool rd = !((ABL & 2) && S >= 1);
        static_for<MH, MI>([&](auto ic) {
            constexpr int i = decltype(ic)::value;
            if constexpr (i + 1 < MI) {
                if (rd) { ds_read16<(i + 1) * 2048>(ah[(i + 1) & 1], aH); ds_read16<(i + 1) * 2048>(al[(i + 1) & 1], aL); }
                wait_lgkm<2>(ah[i & 1], al[i & 1]);
            } else {
                wait_lgkm<0>(ah[i & 1], al[i & 1]);
            }
            mma_block(i);
        });
        if (++c_kt == nk) {
            // tile finished (for this group): bias / activation / convert / whole-line stores through this wave's strip; no
            // barrier inside.  The other group is one half-stage away from the same point.
            const long long e0 = PROF ? clock64() : 0;
            wait_vm<0>();                                // this wave's pieces of the next stage: before the stores enter the counter
            confirmed = true;
            if constexpr (PROF) prof_ew += clock64() - e0;
            const int item = first + c_x * nl;
            const int t = tile0 + item / SUB, sub = item % SUB;
            int tm, tn;
        tile_coords(t, tm, tn);
            if constexpr (!(ABL & 4))
                pp_epilogue<T, OUT_F32, EPI, MI, NI, RESID>(p, acc, has_bias ? bias_rows + (c_x & 1) * 1024 + wq * 256 : nullptr, strips + wq * (2 * 16 * 144),
                                                            tm * 256 + sub * BM + wm0, tn * BN + wn0, lane);
#pragma unroll
            for (int i = 0; i < MI; ++i)
#pragma unroll
                for (int j = 0; j < NI; ++j) acc[i][j] = 0.f;
            c_kt = 0;
            ++c_x;
            if constexpr (PROF) prof_epi += clock64() - e0;
        }
    };

    set_ptrs(0);
    issue(0);
    // Both groups pass the same 2 NS + 1 barriers; each group's loop holds one whole stage of ITS work, so the fragment
    // registers are not live around the back edge (one loop over half-steps for both groups spilled them).
    if (g == 0) {
        for (int S = 0; S < NS; ++S) {
            sync(true);
            half0(S);
            sync(false);
            half1(S);
        }
        sync(false);
    } else {
        sync(false);
        for (int S = 0; S < NS; ++S) {
            sync(true);
            half0(S);
            sync(false);
            half1(S);
        }
    }
    if constexpr (PROF) {
        if (lane == 0 && p.aux) {
            long long* d = (long long*)p.aux + ((size_t)blockIdx.x * 8 + wave) * 8;      // eight counters per wave
            d[0] = clock64() - prof_c0; d[1] = wall_clock64() - prof_w0; d[2] = prof_vm; d[3] = prof_bar; d[4] = prof_epi;
            d[5] = ntl; d[6] = prof_iss; d[7] = prof_ew;
        }
    }
}

template <typename T, bool OUT_F32, int EPI, bool PROF, bool RESID = false>
int launch_pp_t(const GemmParams& p, hipStream_t stream) {
    constexpr int EXTRA = 2 * 1024 + 4 * 2 * 16 * 144;   // bias rows + epilogue strips
    constexpr int LDS = 2 * 512 * 128 + EXTRA, LDS_H = 2 * 384 * 128 + EXTRA;
    auto kern = gemm_pp_kernel<T, OUT_F32, EPI, PROF, 256, RESID>;
    int n_cu = 0;
    if (cap_kernel_setup((const void*)kern, LDS, &n_cu) != 0) return -1;
#ifdef CAP_EXPERIMENTS
    if (const char* e = getenv("CAP_EXP_CUS")) n_cu = std::min(n_cu, std::max(8, atoi(e)));
    if (!PROF && !RESID) {
        if (const char* e = getenv("CAP_EXP_ABLATE")) {   // ablated forms of the product kernel (timing only: see gemm_pp_kernel)
            const int nt = ((p.M + 255) / 256) * ((p.N + 255) / 256), gr = nt < n_cu ? nt : n_cu, abl = atoi(e);
#define CAP_ABL(A)                                                                                                     \
    do {                                                                                                               \
        auto ka = gemm_pp_kernel<T, OUT_F32, EPI, false, 256, false, A>;                                               \
        if (cap_kernel_setup((const void*)ka, LDS, nullptr) != 0) return -1;                                            \
        hipLaunchKernelGGL(ka, dim3(gr), dim3(512), LDS, stream, p);                                                   \
    } while (0)
            if (abl == 1) CAP_ABL(1); else if (abl == 3) CAP_ABL(3); else if (abl == 8) CAP_ABL(8); else if (abl == 10) CAP_ABL(10); else CAP_ABL(0);
#undef CAP_ABL
            CAP_HIP_CHECK(hipGetLastError());
            return 0;
        }
    }
    if (getenv("CAP_EXP_NONPERSIST")) {                  // one tile per workgroup: workgroups retire all the time (tools/stagger_experiment.py)
        hipLaunchKernelGGL(kern, dim3(((p.M + 255) / 256) * ((p.N + 255) / 256)), dim3(512), LDS, stream, p);
        CAP_HIP_CHECK(hipGetLastError());
        return 0;
    }
#endif
    const int ntiles = ((p.M + 255) / 256) * ((p.N + 255) / 256);
    const int rounds = ntiles / n_cu, tail = ntiles - rounds * n_cu;
    // The last, partial round of tiles of a SHORT launch runs as half tiles (128 rows) in a second launch (proj / fc2: 2.31 rounds;
    // a row's sums do not depend on the tile height).  Not for long launches, and not as quarter tiles either: fc1 (9.23 rounds; its
    // 60 leftover tiles as 240 workgroups of 64 rows) measured level both ways, 632-640 us against 629-638 (tools/ab_gemm_libs.py) -
    // a round that fills a quarter of the chip runs with the clock and the memory system to itself and is much shorter than a full one.
    if (!PROF && rounds >= 1 && rounds <= 4 && tail > 0 && 2 * tail <= n_cu) {
        auto kern_h = gemm_pp_kernel<T, OUT_F32, EPI, PROF, 128, RESID>;
        if (cap_kernel_setup((const void*)kern_h, LDS_H, nullptr) != 0) return -1;
        GemmParams q = p;
        q.tile0 = 0; q.tile1 = rounds * n_cu;
        hipLaunchKernelGGL(kern, dim3(n_cu), dim3(512), LDS, stream, q);
        q.tile0 = rounds * n_cu; q.tile1 = ntiles;
        hipLaunchKernelGGL(kern_h, dim3(2 * tail), dim3(512), LDS_H, stream, q);
        CAP_HIP_CHECK(hipGetLastError());
        return 0;
    }
    const int grid = ntiles < n_cu ? ntiles : n_cu;
    hipLaunchKernelGGL(kern, dim3(grid), dim3(512), LDS, stream, p);
    CAP_HIP_CHECK(hipGetLastError());
    return 0;
}

template <typename T>
int launch_pp_type(const GemmParams& p, bool prof, hipStream_t stream) {
    constexpr bool G8 = is_g8<T>;
#ifdef CAP_EXPERIMENTS
    if (prof) {
        if (p.epi == EPI_STORE && !p.out_f32) return launch_pp_t<T, false, EPI_STORE, true>(p, stream);
        if (p.epi == EPI_STORE && p.out_f32) return launch_pp_t<T, true, EPI_STORE, true>(p, stream);
        return -2;
    }
#endif
    switch (p.epi) {
        case EPI_STORE:
            if (p.resid) return launch_pp_t<T, true, EPI_STORE, false, true>(p, stream);
            return p.out_f32 ? launch_pp_t<T, true, EPI_STORE, false>(p, stream) : launch_pp_t<T, false, EPI_STORE, false>(p, stream);
        case EPI_PATCH: return launch_pp_t<T, true, EPI_PATCH, false>(p, stream);
        case EPI_CROSSKV: return launch_pp_t<T, G8, EPI_CROSSKV, false>(p, stream);       // G8: fp32 rows or KV16; bf16: bf16 rows
        default: return -2;
    }
}

}  // namespace

CAP_DEFINE_G8_CLAMP_READER(cap_g8_clamped_gemm_pp)

// G8 or bf16 operands.  Returns -2 (nothing launched, no error set) when the shape or epilogue is not one this kernel takes:
// the caller falls back to another 256x256 kernel.  prof: cycle stamps to p.aux (-DCAP_EXPERIMENTS builds only).
int launch_gemm_pp(int dtype, const GemmParams& p, bool prof, hipStream_t stream) {
    const int kstage = dtype == CAP_DT_G8 ? 32 : 64, esz = dtype == CAP_DT_G8 ? 4 : 2;
    if ((dtype != CAP_DT_G8 && dtype != CAP_DT_BF16) || p.K < 2 * kstage || p.K % kstage != 0) return -2;
    if (p.resid && (prof || p.epi != EPI_STORE || !p.out_f32)) return -2;            // residual operand: fp32 row-major output only
    // 32-bit byte offsets inside A and W
    if ((size_t)p.M * p.lda * esz >= (1ull << 32) || (size_t)p.N * p.ldw * esz >= (1ull << 32)) return -2;
    if (dtype == CAP_DT_BF16 && !p.out_f32 && p.N % 8 != 0) return -2;              // 16-byte bf16 stores
    return dtype == CAP_DT_G8 ? launch_pp_type<g8_t>(p, prof, stream) : launch_pp_type<bf16_t>(p, prof, stream);
}
