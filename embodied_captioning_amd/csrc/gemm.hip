// MFMA GEMM for the captioner hot path (gfx950 / CDNA4).
//
//   C[M,N] = A[M,K] . W[N,K]^T  (+bias, GELU, residual, layout-remapping epilogues)
//
// Both operands are K-contiguous (activations row-major, torch Linear weights [out,in]), so A and W tiles
// have the same shape in LDS: rows of one 128-byte K-slab (64 bf16 / 32 fp32), 16-byte chunks XOR-swizzled
// by ((row>>1)&7) so that the ds_read_b128 fragment reads of a 16- or 32-row MFMA operand are bank-conflict free
// (MI355X LDS: ds_read_b128 is served in 16-lane groups over 64 banks).
//   bf16:  v_mfma_f32_16x16x32_bf16 in every kernel (lane (r16, kg) supplies row r16, 16-byte chunk 4*ks+kg); the
//          first-generation 256x256 kernel (A/B id 5) still uses 32x32x16 - measured bit-identical results.
//   fp32:  v_mfma_f32_32x32x2_f32 (exact fp32 FMA chain, 1/16 the bf16 rate); a 16-byte chunk holds 4 k values,
//          lane half h takes chunk 2*ks+h and feeds 4 MFMAs - A and W use the same k permutation so the sum is
//          unchanged.
// Three kernels:
//   gemm_kernel       generic register-staged tiles (128x128, 64x64, 256x256): D K-slabs in flight in VGPRs (2 for
//                     128x128, 3..6 for the latency-bound 64x64 decode tile), LDS double buffered, one barrier per slab,
//                     epilogue through LDS strips; split-K (EPI_PARTIAL), residual operand, every epilogue kind.
//   gemm_big_kernel   256x256 persistent, LDS-DMA staging (fp32 mode; bf16 first generation).
//   gemm_big2_kernel  bf16 256x256 persistent, second generation (the encoder GEMMs): see its header.
// Edge tiles clamp their load rows and guard their stores.  Workgroup ids are remapped so each XCD (blockIdx % 8) walks
// a contiguous run of tiles that share A panels in its L2.
#include <stdlib.h>

#include "gemm_tile.h"

namespace {

// Store 4 consecutive output columns (col % 4 == 0, never straddles a 64-wide head).
template <typename T, bool OUT_F32, int EPI, bool RESID = true>
__device__ __forceinline__ void epi_store4(const GemmParams& p, int row, int col, f32x4 v, f32x4 biasv) {
    // bias is loaded once per lane by the caller (it depends on the column only); residual rows are loaded here
    if constexpr (is_g8<T>) v *= (1.0f / G8_WSCALE);      // weights are stored scaled (common.h): exact power of two
    if (EPI != EPI_PARTIAL) v += biasv;
    if (EPI != EPI_PARTIAL && p.gelu == 1) {   // p.gelu: 1 = exact-erf GELU, 2 = ReLU (OPT).  Two uniform branches: a
#pragma unroll                                 // per-element select computes the GELU polynomial for ReLU launches too
        for (int i = 0; i < 4; ++i) v[i] = gelu_for<T>(v[i]);
    } else if (EPI != EPI_PARTIAL && p.gelu == 2) {
#pragma unroll
        for (int i = 0; i < 4; ++i) v[i] = fmaxf(v[i], 0.f);
    }
    size_t o; void* base = p.C;
    if constexpr (EPI == EPI_PARTIAL) {          // split-K slice z: raw fp32 partial sums, reduced by the consumer
        float* dst = (float*)p.C + ((size_t)p.p3 * p.M + row) * p.ldc + col;
        *(f32x4*)dst = v;
        return;
    } else if constexpr (EPI == EPI_STORE) {
        o = (size_t)row * p.ldc + col;
        if (RESID && p.resid) v += *(const f32x4*)(p.resid + (size_t)row * p.ldr + col);
    } else if constexpr (EPI == EPI_PATCH) {
        const int b = row / p.p0, pp = row - b * p.p0;
        v += *(const f32x4*)(p.aux + (size_t)(1 + pp) * p.N + col);
        o = ((size_t)b * (p.p0 + 1) + 1 + pp) * p.ldc + col;
    } else if constexpr (EPI == EPI_CROSSKV) {
        const int NT = p.p0, H = p.p1, B = p.p2, Dh = H * 64;
        const int b = row / NT, t = row - b * NT;
        const int l = col / (2 * Dh), r = col - l * 2 * Dh, kv = r / Dh, hd = r - kv * Dh, h = hd >> 6, d = hd & 63;
        o = (((((size_t)l * 2 + kv) * B + b) * H + h) * NT + t) * 64 + d;
    } else {  // EPI_QKVCACHE
        const int H = p.p1, Dh = H * 64;
        if (col < Dh) {
            o = (size_t)row * Dh + col;
        } else {
            const int kv = col / Dh - 1, hd = col % Dh, h = hd >> 6, d = hd & 63;
            o = ((((size_t)kv * p.p0 + row) * H + h) * p.p2 + p.p3) * 64 + d;
            base = p.C2;
        }
    }
    if constexpr (OUT_F32) {
        *(f32x4*)((float*)base + o) = v;
    } else if constexpr (is_g8<T>) {
        static_assert(!is_g8<T> || EPI == EPI_STORE || EPI == EPI_PARTIAL, "G8 output exists for plain row-major stores only");
        store4((g8_t*)base + (o - col), col, make_float4(v[0], v[1], v[2], v[3]));
    } else if constexpr (sizeof(T) == 4) {
        *(f32x4*)((float*)base + o) = v;
    } else {
        bf16x4 w;
#pragma unroll
        for (int i = 0; i < 4; ++i) w[i] = (bf16_t)v[i];
        *(bf16x4*)((bf16_t*)base + o) = w;
    }
}

// D = K-slabs kept in flight in registers (host guarantees nk % D == 0); WPE = min waves per SIMD for the allocator.
template <typename T, int BM, int BN, int WM, int WN, int D, int WPE, bool OUT_F32, int EPI>
__global__ __launch_bounds__((BM / WM) * (BN / WN) * 64, WPE) void gemm_kernel(GemmParams p) {
    // p.splitk > 1: consecutive block ids are the K-slices of one tile; slice index travels in p.p3 (EPI_PARTIAL)
    constexpr int NWM = BM / WM, NWN = BN / WN, NT = NWM * NWN * 64;
    constexpr int MI = WM / 32, NI = WN / 32;
    constexpr int LA = BM * 8 / NT, LB = BN * 8 / NT;      // 16-byte chunks per thread per slab
    static_assert(LA >= 1 && LB >= 1, "tile too small for the thread count");
    constexpr int EPC = Mma<T>::EPC;
    constexpr int SLAB = 8 * EPC;                          // K elements per 128-byte slab
    using vec = typename Mma<T>::vec;

    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int BUF = (BM + BN) * 128;                   // bytes per LDS buffer: A tile then W tile

    // XCD-aware, bijective remap of the linear block id (guide T1): blocks b, b+8, ... share an XCD, and each XCD walks a
    // contiguous run of logical ids.  Logical order: row tile fastest, then column tile, K slice slowest - so the blocks that
    // read the same W slice (all row tiles of one (column tile, K slice)) sit in ONE XCD's L2 and the slice leaves HBM once,
    // and an XCD's run stays inside one or two K slices, so it pulls only that part of A.  (With the column tile fastest, as
    // before, the four row tiles of a 256-row decode GEMM landed on four XCDs and W crossed the fabric four times: 3.6x the
    // algorithmic bytes by the counters, profiles/r02_bench_pmc.json.)  Placement is a speed heuristic only.
    const int S = EPI == EPI_PARTIAL ? p.splitk : 1;
    const int ntm = (p.M + BM - 1) / BM, ntn = (p.N + BN - 1) / BN, nwg = ntm * ntn * S;
    int bid = blockIdx.x;
    {
        int q = nwg >> 3, r = nwg & 7, xcd = bid & 7, idx = bid >> 3;
        bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
    }
    const int tm = bid % ntm;
    bid /= ntm;
    const int tn = bid % ntn, kz = bid / ntn;
    if constexpr (EPI == EPI_PARTIAL) p.p3 = kz;
    const int Ks = p.K / S;                                  // K extent of this block
    const int m0 = tm * BM, n0 = tn * BN;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm0 = (wave / NWN) * WM, wn0 = (wave % NWN) * WN;
    const int r32 = lane & 31, h = lane >> 5;

    const T* A = (const T*)p.A;
    const T* W = (const T*)p.W;

    // per-thread global source pointers and LDS destinations for the staging copies
    const u32x4* ga[LA]; const u32x4* gb[LB]; int da[LA], db[LB];
#pragma unroll
    for (int i = 0; i < LA; ++i) {
        int c = tid + i * NT, row = c >> 3, ch = c & 7;
        int gr = min(m0 + row, p.M - 1);
        ga[i] = (const u32x4*)(A + (size_t)gr * p.lda + (size_t)kz * Ks + ch * EPC);
        da[i] = swz_off<is_g8<T>>(row, ch);
    }
#pragma unroll
    for (int i = 0; i < LB; ++i) {
        int c = tid + i * NT, row = c >> 3, ch = c & 7;
        int gr = min(n0 + row, p.N - 1);
        gb[i] = (const u32x4*)(W + (size_t)gr * p.ldw + (size_t)kz * Ks + ch * EPC);
        db[i] = BM * 128 + swz_off<is_g8<T>>(row, ch);
    }

    constexpr bool G8 = is_g8<T>;
    constexpr bool B16 = sizeof(T) == 2 || G8;            // bf16 / split fp16: 16x16x32 blocks (see Mma<bf16_t>::run16)
    constexpr int MI16 = WM / 16, NI16 = WN / 16;
    const int r16 = lane & 15, kg = lane >> 4;
    f32x16 acc[B16 ? 1 : MI][B16 ? 1 : NI];
    f32x4 acc16[B16 ? MI16 : 1][B16 ? NI16 : 1];
    if constexpr (B16) {
#pragma unroll
        for (int i = 0; i < MI16; ++i)
#pragma unroll
            for (int j = 0; j < NI16; ++j) acc16[i][j] = 0.f;
    } else {
#pragma unroll
        for (int i = 0; i < MI; ++i)
#pragma unroll
            for (int j = 0; j < NI; ++j)
#pragma unroll
                for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;
    }

    u32x4 ra[D][LA], rb[D][LB];
    const int nk = Ks / SLAB;
#pragma unroll
    for (int s = 0; s < D; ++s) {
#pragma unroll
        for (int i = 0; i < LA; ++i) ra[s][i] = ga[i][s * 8];
#pragma unroll
        for (int i = 0; i < LB; ++i) rb[s][i] = gb[i][s * 8];
    }

    for (int k0 = 0; k0 < nk; k0 += D) {
#pragma unroll
        for (int s = 0; s < D; ++s) {
            const int kt = k0 + s;
            char* buf = smem + (kt & 1) * BUF;
            // slab kt: registers -> LDS (waits only for this slab's loads; younger slabs stay in flight)
#pragma unroll
            for (int i = 0; i < LA; ++i) *(u32x4*)(buf + da[i]) = ra[s][i];
#pragma unroll
            for (int i = 0; i < LB; ++i) *(u32x4*)(buf + db[i]) = rb[s][i];
            if (kt + D < nk) {           // refill this register stage with slab kt + D
                const int ko = (kt + D) * 8;
#pragma unroll
                for (int i = 0; i < LA; ++i) ra[s][i] = ga[i][ko];
#pragma unroll
                for (int i = 0; i < LB; ++i) rb[s][i] = gb[i][ko];
            }
            __syncthreads();
            const char* a_s = buf;
            const char* b_s = buf + BM * 128;
            if constexpr (G8) {
                vec ah[MI16], al[MI16], bh[NI16], bl[NI16];
#pragma unroll
                for (int i = 0; i < MI16; ++i) {
                    ah[i] = *(const vec*)(a_s + swz_off<is_g8<T>>(wm0 + i * 16 + r16, 2 * kg));
                    al[i] = *(const vec*)(a_s + swz_off<is_g8<T>>(wm0 + i * 16 + r16, 2 * kg + 1));
                }
#pragma unroll
                for (int j = 0; j < NI16; ++j) {
                    bh[j] = *(const vec*)(b_s + swz_off<is_g8<T>>(wn0 + j * 16 + r16, 2 * kg));
                    bl[j] = *(const vec*)(b_s + swz_off<is_g8<T>>(wn0 + j * 16 + r16, 2 * kg + 1));
                }
#pragma unroll
                for (int i = 0; i < MI16; ++i)
#pragma unroll
                    for (int j = 0; j < NI16; ++j) {
                        Mma<g8_t>::run16(acc16[i][j], ah[i], bl[j]);
                        Mma<g8_t>::run16(acc16[i][j], al[i], bh[j]);
                        Mma<g8_t>::run16(acc16[i][j], ah[i], bh[j]);
                    }
            } else if constexpr (B16) {
#pragma unroll
                for (int ks = 0; ks < 2; ++ks) {
                    vec af[MI16], bf[NI16];
#pragma unroll
                    for (int i = 0; i < MI16; ++i) af[i] = *(const vec*)(a_s + swz_off<is_g8<T>>(wm0 + i * 16 + r16, ks * 4 + kg));
#pragma unroll
                    for (int j = 0; j < NI16; ++j) bf[j] = *(const vec*)(b_s + swz_off<is_g8<T>>(wn0 + j * 16 + r16, ks * 4 + kg));
#pragma unroll
                    for (int i = 0; i < MI16; ++i)
#pragma unroll
                        for (int j = 0; j < NI16; ++j) Mma<bf16_t>::run16(acc16[i][j], af[i], bf[j]);
                }
            } else {
#pragma unroll
                for (int ks = 0; ks < 4; ++ks) {
                    vec af[MI], bf[NI];
#pragma unroll
                    for (int i = 0; i < MI; ++i) af[i] = *(const vec*)(a_s + swz_off<is_g8<T>>(wm0 + i * 32 + r32, ks * 2 + h));
#pragma unroll
                    for (int j = 0; j < NI; ++j) bf[j] = *(const vec*)(b_s + swz_off<is_g8<T>>(wn0 + j * 32 + r32, ks * 2 + h));
#pragma unroll
                    for (int i = 0; i < MI; ++i)
#pragma unroll
                        for (int j = 0; j < NI; ++j) Mma<T>::run(acc[i][j], af[i], bf[j]);
                }
            }
        }
    }
    __syncthreads();   // every wave is done with the staging buffers: reuse them for the epilogue

    // Epilogue through LDS: a wave parks one 32 x WN fp32 strip of its accumulators in a private LDS region
    // (acc[i][j][e] is C[row = (e&3) + 8*(e>>2) + 4*h][col = lane&31] of its 32x32 tile), then reads it back
    // row-wise so each lane owns 4 consecutive columns: 16-byte (fp32) / 8-byte (bf16) coalesced global stores.
    float* strip = (float*)(smem + wave * (32 * WN * 4));
    constexpr int LPR = WN / 4;          // lanes per row
    constexpr int RPP = 64 / LPR;        // rows per pass
    const int rl = lane / LPR, cl = (lane % LPR) * 4;
#pragma unroll
    for (int i = 0; i < MI; ++i) {
        if constexpr (B16) {
            // rows i*32 .. i*32+31 of the wave tile = 16-row blocks 2i, 2i+1; acc16[.][j][e] is C[4 kg + e][r16]
#pragma unroll
            for (int ii = 0; ii < 2; ++ii)
#pragma unroll
                for (int j = 0; j < NI16; ++j)
#pragma unroll
                    for (int e = 0; e < 4; ++e)
                        strip[(ii * 16 + 4 * kg + e) * WN + j * 16 + r16] = acc16[i * 2 + ii][j][e];
        } else {
#pragma unroll
            for (int j = 0; j < NI; ++j)
#pragma unroll
                for (int e = 0; e < 16; ++e)
                    strip[((e & 3) + 8 * (e >> 2) + 4 * h) * WN + j * 32 + r32] = acc[i][j][e];
        }
        // same-wave LDS accesses complete in order; the compiler's lgkmcnt wait orders the read-back
        const int col = n0 + wn0 + cl;
        f32x4 biasv = 0.f;
        if (EPI != EPI_PARTIAL && p.bias && col < p.N) biasv = *(const f32x4*)(p.bias + col);
#pragma unroll
        for (int ps = 0; ps < 32 / RPP; ++ps) {
            const int rr = ps * RPP + rl;
            const int row = m0 + wm0 + i * 32 + rr;
            const f32x4 v = *(const f32x4*)(strip + rr * WN + cl);
            if (row < p.M && col < p.N) epi_store4<T, OUT_F32, EPI>(p, row, col, v, biasv);
        }
    }
}

// ------------------------------------------------------------------------------------------------------------------
// 256x256 persistent kernel for the encoder-sized GEMMs.  One 8-wave workgroup per CU walks a run of output tiles.
//   * global -> LDS by LDS-DMA (`global_load_lds_dwordx4`, 1 KiB = 8 tile rows per wave-instruction, no VGPR staging,
//     no ds_write); the XOR chunk swizzle is applied on the per-lane SOURCE address because the DMA destination is
//     lane-linear, and undone by the same XOR on the fragment reads;
//   * two 64 KiB stage buffers, the DMA of slab k+1 is issued right after the barrier that retires slab k and lands
//     under slab k's 32 MFMAs per wave;
//   * the next tile's first slab is issued BEFORE this tile's epilogue, so the C write-back (through wave-private LDS
//     strips that alias the stage buffer just drained) overlaps the DMA instead of leaving the CU idle.

// VAR selects the scheduling of the inner k-step loop: 0 compiler scheduled (fp32), 2 iglp_opt(0), 3 iglp_opt(1) (bf16),
// 4 = 3 + cycle stamps (diagnostic build, tools/gemm_cycles.py).  (A manual register double-buffer + sched_group_barrier
// variant measured below iglp_opt(1) and was removed.)
template <typename T, bool OUT_F32, int EPI, int VAR = 0>
__global__ __launch_bounds__(512, 2) void gemm_big_kernel(GemmParams p) {
    constexpr int BM = 256, BN = 256, WM = 128, WN = 64, MI = 4, NI = 2;
    constexpr int EPC = Mma<T>::EPC, SLAB = 8 * EPC;
    constexpr int STAGE = (BM + BN) * 128;               // 64 KiB
    using vec = typename Mma<T>::vec;
    extern __shared__ __attribute__((aligned(16))) char smem[];

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm0 = (wave >> 2) * WM, wn0 = (wave & 3) * WN;
    const int r32 = lane & 31, h = lane >> 5;
    const int ntm = (p.M + BM - 1) / BM, ntn = (p.N + BN - 1) / BN, ntiles = ntm * ntn;
    const int nk = p.K / SLAB;

    // XCD-aware tile schedule: XCD x (= blockIdx % 8) owns a contiguous run of tiles; its blocks stride through it,
    // so the tiles in flight on one XCD are consecutive (n fastest -> shared A panel and neighbouring W panels in L2).
    const int xcd = blockIdx.x & 7, li = blockIdx.x >> 3;
    const int nl = ((int)gridDim.x >> 3) + (xcd < ((int)gridDim.x & 7) ? 1 : 0);
    const int tq = ntiles >> 3, tr = ntiles & 7;
    const int c0 = xcd < tr ? xcd * (tq + 1) : tr * (tq + 1) + (xcd - tr) * tq;
    const int c1 = c0 + tq + (xcd < tr ? 1 : 0);

    // DMA addressing: wave w issues 4 A pieces and 4 W pieces per slab; piece j covers tile rows (w*4+j)*8 .. +7.
    const int prow = lane >> 3, ppos = lane & 7;
    const T* A = (const T*)p.A;
    const T* W = (const T*)p.W;

    auto issue = [&](int tile, int kt, char* stage) {
        const int tm = tile / ntn, tn = tile - tm * ntn;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int rowbase = (wave * 4 + j) * 8, row = rowbase + prow;
            const int gch = ppos ^ swz_key<is_g8<T>>(row);
            const int ga = min(tm * BM + row, p.M - 1), gb = min(tn * BN + row, p.N - 1);
            const T* sa = A + (size_t)ga * p.lda + (size_t)kt * SLAB + gch * EPC;
            const T* sb = W + (size_t)gb * p.ldw + (size_t)kt * SLAB + gch * EPC;
            __builtin_amdgcn_global_load_lds(CAP_GPTR(sa), CAP_LPTR(stage + rowbase * 128), 16, 0, 0);
            __builtin_amdgcn_global_load_lds(CAP_GPTR(sb), CAP_LPTR(stage + BM * 128 + rowbase * 128), 16, 0, 0);
        }
    };

    // bias rides the same LDS-DMA path (a register-destination load would make hipcc drain the DMA queue at every use):
    // wave 0 fetches the tile's 256 bias values (1 KiB) into a ping-pong slot behind the stage buffers
    char* bias_lds = smem + 2 * STAGE;
    const bool has_bias = EPI != EPI_PARTIAL && p.bias != nullptr;
    auto issue_bias = [&](int tile, int slot) {
        if (has_bias && wave == 0) {
            const int tn = tile % ntn;
            const float* sb = p.bias + min(tn * BN + lane * 4, p.N - 4);
            __builtin_amdgcn_global_load_lds(CAP_GPTR(sb), CAP_LPTR(bias_lds + slot * 1024), 16, 0, 0);
        }
    };

    long long prof_dma = 0, prof_bar = 0, prof_epi = 0;
    const long long prof_c0 = VAR == 4 ? clock64() : 0, prof_w0 = VAR == 4 ? wall_clock64() : 0;
    int cnt = 0;                                          // running slab counter: slab uses stage buffer cnt & 1
    int tile = c0 + li;
    int tcount = 0;                                       // tiles done by this block: bias slot = tcount & 1
    if (tile < c1) { issue_bias(tile, 0); issue(tile, 0, smem); }
    for (; tile < c1; tile += nl, ++tcount) {
        const int tm = tile / ntn, tn = tile - tm * ntn;
        const int m0 = tm * BM, n0 = tn * BN;
        constexpr int LPR = WN / 4, RPP = 64 / LPR, NPS = 32 / RPP;
        const int rl = lane / LPR, cl = (lane % LPR) * 4;
        const int col = n0 + wn0 + cl;
        f32x16 acc[MI][NI];
#pragma unroll
        for (int i = 0; i < MI; ++i)
#pragma unroll
            for (int j = 0; j < NI; ++j)
#pragma unroll
                for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

        for (int kt = 0; kt < nk; ++kt, ++cnt) {
            if constexpr (VAR == 4) {   // instrumented: cycles waiting for this wave's DMA vs. for the other waves
                const long long t0 = clock64();
                asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
                const long long t1 = clock64();
                CAP_RAW_BARRIER();
                const long long t2 = clock64();
                prof_dma += t1 - t0; prof_bar += t2 - t1;
            } else
            __syncthreads();      // vmcnt(0): this wave's pieces of slab kt landed; barrier: everybody's did, and
                                  // every wave is done reading the other stage buffer
            if (kt + 1 < nk) issue(tile, kt + 1, smem + ((cnt + 1) & 1) * STAGE);
            const char* a_s = smem + (cnt & 1) * STAGE;
            const char* b_s = a_s + BM * 128;
            {
                if constexpr (VAR == 2) __builtin_amdgcn_iglp_opt(0);
                if constexpr (VAR >= 3) __builtin_amdgcn_iglp_opt(1);
#pragma unroll
                for (int ks = 0; ks < 4; ++ks) {
                    vec af[MI], bf[NI];
#pragma unroll
                    for (int i = 0; i < MI; ++i) af[i] = *(const vec*)(a_s + swz_off<is_g8<T>>(wm0 + i * 32 + r32, ks * 2 + h));
#pragma unroll
                    for (int j = 0; j < NI; ++j) bf[j] = *(const vec*)(b_s + swz_off<is_g8<T>>(wn0 + j * 32 + r32, ks * 2 + h));
#pragma unroll
                    for (int i = 0; i < MI; ++i)
#pragma unroll
                        for (int j = 0; j < NI; ++j) Mma<T>::run(acc[i][j], af[i], bf[j]);
                }
            }
        }
        // Epilogue.  Phase 1: transpose the accumulators through wave-private LDS strips (in the stage buffer that was
        // just multiplied) back into registers, row-major: each lane ends up with 4 consecutive columns of 32 rows.
        // Phase 2 (after the next tile's first slab DMA has been issued - no LDS access from here on, so hipcc has no
        // reason to drain the DMA queue): bias / GELU / residual, convert, 16-byte (fp32) or 8-byte (bf16) stores.
        const long long prof_e0 = VAR == 4 ? clock64() : 0;
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        CAP_RAW_BARRIER();                       // every wave is done reading stage (cnt-1) & 1
        float* strip = (float*)(smem + ((cnt + 1) & 1) * STAGE + wave * (32 * WN * 4));
        f32x4 tr[MI][NPS];
#pragma unroll
        for (int i = 0; i < MI; ++i) {
#pragma unroll
            for (int j = 0; j < NI; ++j)
#pragma unroll
                for (int e = 0; e < 16; ++e)
                    strip[((e & 3) + 8 * (e >> 2) + 4 * h) * WN + j * 32 + r32] = acc[i][j][e];
#pragma unroll
            for (int ps = 0; ps < NPS; ++ps) tr[i][ps] = *(const f32x4*)(strip + (ps * RPP + rl) * WN + cl);
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        // stage buffer cnt & 1 was last read one slab ago and everyone has passed a barrier since: prefetch the next
        // tile's first slab into it now; it lands under phase 2
        if (tile + nl < c1) { issue_bias(tile + nl, (tcount + 1) & 1); issue(tile + nl, 0, smem + (cnt & 1) * STAGE); }
        f32x4 biasv = 0.f;
        if (has_bias) biasv = *(const f32x4*)(bias_lds + (tcount & 1) * 1024 + (wn0 + cl) * 4);
        // (no residual operand in this kernel: the ViT branch outputs go to `delta` and the add+LayerNorm kernel folds
        //  them into the residual stream - a register-destination load here would make hipcc drain the DMA queue)
#pragma unroll
        for (int i = 0; i < MI; ++i)
#pragma unroll
            for (int ps = 0; ps < NPS; ++ps) {
                const int row = m0 + wm0 + i * 32 + ps * RPP + rl;
                if (row < p.M && col < p.N) epi_store4<T, OUT_F32, EPI, false>(p, row, col, tr[i][ps], biasv);
            }
        if constexpr (VAR == 4) prof_epi += clock64() - prof_e0;
    }
    if constexpr (VAR == 4) {
        if (lane == 0 && p.aux) {
            long long* d = (long long*)p.aux + ((size_t)blockIdx.x * 8 + wave) * 6;
            d[0] = clock64() - prof_c0; d[1] = wall_clock64() - prof_w0; d[2] = prof_dma; d[3] = prof_bar; d[4] = prof_epi;
            d[5] = tcount;
        }
    }
}

// Epilogue stores of the second-generation kernel: values are final (bias / GELU applied, converted); these only map
// (row, col) to the destination of the epilogue in use and store 16 bytes (8 bf16 or 4 fp32 consecutive columns).
// ------------------------------------------------------------------------------------------------------------------
// bf16 256x256 persistent kernel, second generation.  Same LDS-DMA staging, swizzle and XCD-aware tile walk as
// gemm_big_kernel; what changed follows from in-kernel cycle accounting (tools/gemm_cycles.py: the old epilogue cost
// ~10 k cycles per tile against ~3.6 k per K-slab, and the chip holds ~1.85 GHz under this loop):
//   * v_mfma_f32_16x16x32_bf16 instead of 32x32x16 (same LDS bytes, same cycles per flop; the chip sustains a higher
//     clock on this shape - MI355X_MICROARCH.md, DVFS give-back item 7);
//   * operands swapped (W fragment as MFMA A, activation fragment as MFMA B), so a lane's 4 accumulator registers of a
//     16x16 block are 4 CONSECUTIVE OUTPUT COLUMNS of one row: bias/GELU/convert happen in that layout and the transpose
//     to whole-cache-line stores goes through a small wave-private strip with 8/16-byte LDS writes (the old epilogue
//     wrote 4 bytes per lane into strips aliasing the stage buffers and needed a barrier on each side; storing straight
//     from this layout - 16 rows x 32 B per instruction - measured 14-18 k cycles per tile, worse than the old 10 k);
//   * the slab pipeline runs across tile boundaries: the next tile's first slab is issued during this tile's last slab;
//     a wave confirms its own pieces landed (vmcnt(0)) BEFORE its epilogue stores, so the barrier after the epilogue
//     does not wait for those stores.
// VAR: 0 compiler schedule, 1 iglp_opt(0), 2 iglp_opt(1); PROF: cycle stamps to p.aux (diagnostic build only).
// (Cache-policy A/B on MI355X: non-temporal A loads -5..-20 %, non-temporal C stores within noise - neither kept.)
// T = g8_t (split fp16): the same pipeline with 32 k values per 64 KiB stage ([hi | lo] chunk pairs) and three MFMAs per
// product (see Mma<g8_t>); per stage and wave 96 MFMAs against 24 ds_read_b128.
// BM = 128: the same kernel on HALF tiles (the upper / lower 128 rows of a 256 x 256 tile; waves 64 x 64 each) - what the
// launcher runs over the last, partial round of tiles (launch_big2).
template <typename T, bool OUT_F32, int EPI, int VAR, bool PROF, int NWM = 2, int NWN = 4, int BM = 256>
__global__ __launch_bounds__(NWM * NWN * 64, NWM * NWN == 8 ? 2 : 1) void gemm_big2_kernel(GemmParams p) {
    using vec = typename Mma<T>::vec;
    // wave grid NWM x NWN over the 256x256 tile: 2x4 = two waves per SIMD, 128x64 each (shipped).  2x2 = one wave per
    // SIMD with 128x128 each (256 accumulator registers, a third less LDS read traffic per flop) measured 7-25 % SLOWER
    // with the compiler's schedule (212 B/lane of scratch at 512 registers), so it is not instantiated.
    constexpr int BN = 256, NW = NWM * NWN, WM = BM / NWM, WN = BN / NWN, MI = WM / 16, NI = WN / 16;
    constexpr int SUB = 256 / BM;                        // work items per 256-row tile (1, or 2 halves)
    constexpr int PPA = BM / 8 / NW, PPW = BN / 8 / NW;  // 8-row DMA pieces of A / of W per wave per slab
    constexpr int EPC = Mma<T>::EPC, STAGE = (BM + BN) * 128;      // 64 KiB (48 KiB for half tiles)
    constexpr int SCHED = VAR;
    extern __shared__ __attribute__((aligned(16))) char smem[];

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm0 = (wave / NWN) * WM, wn0 = (wave % NWN) * WN;
    const int r16 = lane & 15, kg = lane >> 4;
    // the tile grid is always that of 256 x 256 tiles; this launch covers tiles [tile0, tile1) of it (0, 0 = all), as
    // (tile1 - tile0) * SUB work items: item x = sub-tile x % SUB of tile tile0 + x / SUB
    const int ntm = (p.M + 255) / 256, ntn = (p.N + BN - 1) / BN;
    const int tile0 = p.tile1 > 0 ? p.tile0 : 0;
    const int ntiles = ((p.tile1 > 0 ? p.tile1 : ntm * ntn) - tile0) * SUB;
    const int nk = p.K / (8 * EPC);

    const int xcd = blockIdx.x & 7, li = blockIdx.x >> 3;
    const int nl = ((int)gridDim.x >> 3) + (xcd < ((int)gridDim.x & 7) ? 1 : 0);
    const int tq = ntiles >> 3, tr = ntiles & 7;
    const int c0 = xcd < tr ? xcd * (tq + 1) : tr * (tq + 1) + (xcd - tr) * tq;
    const int c1 = c0 + tq + (xcd < tr ? 1 : 0);

    // issue side of the pipeline: byte pointers (k = 0) of this lane's 4 A and 4 W pieces of the tile being fetched
    const int prow = lane >> 3, ppos = lane & 7;
    const char* pa[PPA];
    const char* pb[PPW];
    auto set_ptrs = [&](int x) {
        const int t = tile0 + x / SUB, sub = x % SUB;
        const int tm = t / ntn, tn = t - tm * ntn;
#pragma unroll
        for (int j = 0; j < PPA; ++j) {
            const int row = (wave * PPA + j) * 8 + prow;
            const int gch = ppos ^ swz_key<is_g8<T>>(row);
            const int ga = min(tm * 256 + sub * BM + row, p.M - 1);
            pa[j] = (const char*)((const T*)p.A + (size_t)ga * p.lda + gch * EPC);
        }
#pragma unroll
        for (int j = 0; j < PPW; ++j) {
            const int row = (wave * PPW + j) * 8 + prow;
            const int gch = ppos ^ swz_key<is_g8<T>>(row);
            const int gb = min(tn * BN + row, p.N - 1);
            pb[j] = (const char*)((const T*)p.W + (size_t)gb * p.ldw + gch * EPC);
        }
    };
    auto issue = [&](int kt, char* stage) {
#pragma unroll
        for (int j = 0; j < PPW; ++j) {
            if (j < PPA)
                __builtin_amdgcn_global_load_lds(CAP_GPTR(pa[j] + (size_t)kt * 128), CAP_LPTR(stage + (wave * PPA + j) * 8 * 128), 16, 0, 0);
            __builtin_amdgcn_global_load_lds(CAP_GPTR(pb[j] + (size_t)kt * 128), CAP_LPTR(stage + BM * 128 + (wave * PPW + j) * 8 * 128), 16, 0, 0);
        }
    };
    char* bias_lds = smem + 2 * STAGE;
    const bool has_bias = EPI != EPI_PARTIAL && p.bias != nullptr;
    auto issue_bias = [&](int x, int slot) {
        if (has_bias && wave == 0) {
            const int tn = (tile0 + x / SUB) % ntn;
            const float* sb = p.bias + min(tn * BN + lane * 4, p.N - 4);
            __builtin_amdgcn_global_load_lds(CAP_GPTR(sb), CAP_LPTR(bias_lds + slot * 1024), 16, 0, 0);
        }
    };

    long long prof_dma = 0, prof_bar = 0, prof_epi = 0;
    const long long prof_c0 = PROF ? clock64() : 0, prof_w0 = PROF ? wall_clock64() : 0;

    int cnt = 0, tcount = 0;
    int tile = c0 + li;
    int itile = tile, ikt = 0;
    auto advance = [&]() {
        if (++ikt == nk) {
            ikt = 0;
            itile += nl;
            if (itile < c1) set_ptrs(itile);
        }
    };
    if (itile < c1) { set_ptrs(itile); issue_bias(itile, 0); issue(0, smem); advance(); }

    for (; tile < c1; tile += nl, ++tcount) {
        const int ft = tile0 + tile / SUB;
        const int tm = ft / ntn, tn = ft - tm * ntn;
        const int m0 = tm * 256 + (tile % SUB) * BM, n0 = tn * BN;
        f32x4 acc[MI][NI];
#pragma unroll
        for (int i = 0; i < MI; ++i)
#pragma unroll
            for (int j = 0; j < NI; ++j) acc[i][j] = 0.f;

        for (int kt = 0; kt < nk; ++kt, ++cnt) {
            if (kt > 0 || tcount == 0) {
                // vmcnt(0): this wave's pieces of the slab landed; barrier: everybody's did and every wave is done
                // reading the other stage buffer.  (First slab of later tiles: see the end of the tile loop.)
                if constexpr (PROF) {
                    const long long t0 = clock64();
                    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
                    const long long t1 = clock64();
                    CAP_RAW_BARRIER();
                    prof_dma += t1 - t0; prof_bar += clock64() - t1;
                } else {
                    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
                    CAP_RAW_BARRIER();
                }
            }
            if (itile < c1) {
                if (ikt == 0) issue_bias(itile, (tcount + 1) & 1);
                issue(ikt, smem + ((cnt + 1) & 1) * STAGE);
                advance();
            }
            const char* a_s = smem + (cnt & 1) * STAGE;
            const char* b_s = a_s + BM * 128;
            if constexpr (SCHED == 1) __builtin_amdgcn_iglp_opt(0);
            if constexpr (SCHED == 2) __builtin_amdgcn_iglp_opt(1);
            if constexpr (is_g8<T>) {
                vec bh[NI], bl[NI];
#pragma unroll
                for (int j = 0; j < NI; ++j) {
                    bh[j] = *(const vec*)(b_s + swz_off<is_g8<T>>(wn0 + j * 16 + r16, 2 * kg));
                    bl[j] = *(const vec*)(b_s + swz_off<is_g8<T>>(wn0 + j * 16 + r16, 2 * kg + 1));
                }
#pragma unroll
                for (int i = 0; i < MI; ++i) {
                    const vec ah = *(const vec*)(a_s + swz_off<is_g8<T>>(wm0 + i * 16 + r16, 2 * kg));
                    const vec al = *(const vec*)(a_s + swz_off<is_g8<T>>(wm0 + i * 16 + r16, 2 * kg + 1));
#pragma unroll
                    for (int j = 0; j < NI; ++j) {
                        // three dependent MFMAs in a row on one accumulator: measured FASTER than three passes over j (the
                        // matrix pipe forwards the accumulator; tools/bench_gemm_split.py: 458 vs 478 us on the qkv shape)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(bl[j], ah, acc[i][j], 0, 0, 0);
                        acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(bh[j], al, acc[i][j], 0, 0, 0);
                        acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(bh[j], ah, acc[i][j], 0, 0, 0);
                    }
                }
            } else {
#pragma unroll
                for (int ks = 0; ks < 2; ++ks) {
                    vec af[MI], bf[NI];
#pragma unroll
                    for (int j = 0; j < NI; ++j) bf[j] = *(const vec*)(b_s + swz_off<is_g8<T>>(wn0 + j * 16 + r16, ks * 4 + kg));
#pragma unroll
                    for (int i = 0; i < MI; ++i) af[i] = *(const vec*)(a_s + swz_off<is_g8<T>>(wm0 + i * 16 + r16, ks * 4 + kg));
#pragma unroll
                    for (int i = 0; i < MI; ++i)
#pragma unroll
                        for (int j = 0; j < NI; ++j)
                            acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bf[j], af[i], acc[i][j], 0, 0, 0);
                }
            }
        }
        // This wave's pieces of the next tile's first slab (issued one slab ago) have landed - checked now, before the
        // epilogue stores enter the same in-order counter.
        const long long prof_e0 = PROF ? clock64() : 0;
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        big2_epilogue<T, OUT_F32, EPI, MI, NI>(p, acc, smem + 2 * STAGE + 2048 + wave * (16 * 144),
                                            has_bias ? bias_lds + (tcount & 1) * 1024 + wn0 * 4 : nullptr, m0 + wm0, n0 + wn0, lane);
        if (tile + nl < c1) {
            // every wave is done reading the last slab's stage buffer and the bias slot, and has seen its own pieces of
            // the next tile's first slab land: that slab may be read and the other buffer overwritten
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            CAP_RAW_BARRIER();
        }
        if constexpr (PROF) prof_epi += clock64() - prof_e0;
    }
    if constexpr (PROF) {
        if (lane == 0 && p.aux) {
            long long* d = (long long*)p.aux + ((size_t)blockIdx.x * 8 + wave) * 6;
            d[0] = clock64() - prof_c0; d[1] = wall_clock64() - prof_w0; d[2] = prof_dma; d[3] = prof_bar; d[4] = prof_epi;
            d[5] = tcount;
        }
    }
}

// ------------------------------------------------------------------------------------------------------------------
// Third structure (A/B): FOUR 32 KiB stages of HALF K-slabs (256+256 rows x 32 k = 64-byte rows, one 16x16x32 MFMA
// k-step each).  With three half-slabs of DMA look-ahead a half-slab has landed - and been confirmed by a barrier - one
// whole iteration before it is multiplied, so its fragments are read into a second register set UNDER the previous
// half-slab's MFMAs: after a barrier the matrix pipe has 32 MFMAs per wave whose operands are already in registers, the
// ds_read latency that gemm_big2_kernel exposes at the start of every slab is gone.  Cost: a barrier every 32 MFMAs per
// wave instead of every 64, counted vmcnt (one half-slab of this wave's DMA may stay in flight across the barrier).
// Measured against gemm_big2_kernel (tools/gemm_cycles.py, interleaved rounds): +3..7 % on the K = 768 encoder shapes
// (fragments of the NEXT tile's first half-slab are already in registers when a tile ends), equal at K = 3072, -20 % on
// 8192^3 - so the dispatcher uses it for K <= 1024.  A variant with one barrier per 64-wide slab and the same prefetch
// (two half-slabs issued per barrier, vmcnt(0)) was not better than gemm_big2_kernel anywhere and was dropped; iglp_opt(1)
// (SCHED = 1) makes no difference here.
//   64-byte rows: chunk c of row r is stored at chunk c ^ P[(r >> 2) & 3], P = {0, 2, 3, 1} - with that permutation the
//   16 lanes of every ds_read_b128 service group ({0-3,12-15,20-27}, ...) hit 16 distinct 4-bank groups.
__device__ __forceinline__ int swz64_off(int row, int chunk) {
    return row * 64 + ((chunk ^ ((0x78 >> (2 * ((row >> 2) & 3))) & 3)) << 4);
}

template <bool OUT_F32, int EPI, int SCHED = 0>
__global__ __launch_bounds__(512, 2) void gemm_big3_kernel(GemmParams p) {
    using T = bf16_t;
    using vec = bf16x8;
    constexpr int BM = 256, BN = 256, WM = 128, WN = 64, MI = WM / 16, NI = WN / 16;
    constexpr int EPC = 8, HSTAGE = (BM + BN) * 64;      // 32 KiB per half-slab
    extern __shared__ __attribute__((aligned(16))) char smem[];

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm0 = (wave >> 2) * WM, wn0 = (wave & 3) * WN;
    const int r16 = lane & 15, kg = lane >> 4;
    const int ntm = (p.M + BM - 1) / BM, ntn = (p.N + BN - 1) / BN, ntiles = ntm * ntn;
    const int nh = p.K >> 5;                              // half-slabs per tile (even: K % 64 == 0)

    const int xcd = blockIdx.x & 7, li = blockIdx.x >> 3;
    const int nl = ((int)gridDim.x >> 3) + (xcd < ((int)gridDim.x & 7) ? 1 : 0);
    const int tq = ntiles >> 3, tr = ntiles & 7;
    const int c0 = xcd < tr ? xcd * (tq + 1) : tr * (tq + 1) + (xcd - tr) * tq;
    const int c1 = c0 + tq + (xcd < tr ? 1 : 0);

    // DMA: a wave-instruction covers 16 rows x 64 B; wave w issues pieces 2w, 2w+1 of A and of W per half-slab
    const int prow = lane >> 2, ppos = lane & 3;
    const char* pa[2];
    const char* pb[2];
    auto set_ptrs = [&](int t) {
        const int tm = t / ntn, tn = t - tm * ntn;
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int row = (wave * 2 + j) * 16 + prow;
            const int gch = ppos ^ ((0x78 >> (2 * ((row >> 2) & 3))) & 3);
            const int ga = min(tm * BM + row, p.M - 1), gb = min(tn * BN + row, p.N - 1);
            pa[j] = (const char*)((const T*)p.A + (size_t)ga * p.lda + gch * EPC);
            pb[j] = (const char*)((const T*)p.W + (size_t)gb * p.ldw + gch * EPC);
        }
    };
    auto issue = [&](int h, char* stage) {                // half-slab h of the tile being fetched: k = 32 h .. 32 h + 31
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int off = (wave * 2 + j) * 1024;
            __builtin_amdgcn_global_load_lds(CAP_GPTR(pa[j] + (size_t)h * 64), CAP_LPTR(stage + off), 16, 0, 0);
            __builtin_amdgcn_global_load_lds(CAP_GPTR(pb[j] + (size_t)h * 64), CAP_LPTR(stage + BM * 64 + off), 16, 0, 0);
        }
    };
    char* bias_lds = smem + 4 * HSTAGE;
    const bool has_bias = EPI != EPI_PARTIAL && p.bias != nullptr;
    // the bias DMA must not change the per-wave count of outstanding vector-memory operations that the counted waits
    // rely on, so EVERY wave issues one (waves 1..7 re-fetch the same 1 KiB into a scratch slot of their own)
    auto issue_bias = [&](int t, int slot) {
        const int tn = t % ntn;
        const float* sb = has_bias ? p.bias + min(tn * BN + lane * 4, p.N - 4) : (const float*)p.W;
        char* dst = wave == 0 ? bias_lds + slot * 1024 : bias_lds + 2048 + wave * 1024;
        __builtin_amdgcn_global_load_lds(CAP_GPTR(sb), CAP_LPTR(dst), 16, 0, 0);
    };
    auto load_frags = [&](const char* stage, vec (&af)[MI], vec (&bf)[NI]) {
#pragma unroll
        for (int j = 0; j < NI; ++j) bf[j] = *(const vec*)(stage + BM * 64 + swz64_off(wn0 + j * 16 + r16, kg));
#pragma unroll
        for (int i = 0; i < MI; ++i) af[i] = *(const vec*)(stage + swz64_off(wm0 + i * 16 + r16, kg));
    };

    int g = 0;                                            // global half-slab counter of this block: stage = g & 3
    int tcount = 0;
    int tile = c0 + li;
    int itile = tile, ih = 0;                             // issue side
    auto issue_next = [&](int gi) {                       // fetch the issue side's next half-slab into stage gi & 3
        if (itile < c1) {
            issue(ih, smem + (gi & 3) * HSTAGE);
            if (++ih == nh) {
                ih = 0;
                itile += nl;
                if (itile < c1) set_ptrs(itile);
            }
        } else {
            // past the end: a dummy 4-operation group keeps the counted waits exact (re-reads 1 KiB of the last tile's
            // operands into a sink)
            char* sink = smem + 4 * HSTAGE + 2048 + 8 * 1024;
#pragma unroll
            for (int j = 0; j < 4; ++j) __builtin_amdgcn_global_load_lds(CAP_GPTR(pa[0]), CAP_LPTR(sink), 16, 0, 0);
        }
    };
    if (tile >= c1) return;
    set_ptrs(itile);
    issue_bias(itile, 0);
    issue_next(0); issue_next(1); issue_next(2);          // half-slabs 0, 1, 2 in flight: 4 operations per group per wave
    asm volatile("s_waitcnt vmcnt(8)" ::: "memory");      // bias + half-slab 0 landed (this wave's pieces)
    CAP_RAW_BARRIER();
    vec fa0[MI], fb0[NI], fa1[MI], fb1[NI];
    load_frags(smem, fa0, fb0);

    for (; tile < c1; tile += nl, ++tcount) {
        const int tm = tile / ntn, tn = tile - tm * ntn;
        const int m0 = tm * BM, n0 = tn * BN;
        f32x4 acc[MI][NI];
#pragma unroll
        for (int i = 0; i < MI; ++i)
#pragma unroll
            for (int j = 0; j < NI; ++j) acc[i][j] = 0.f;

        for (int h = 0; h < nh; h += 2, g += 2) {
            // ---- even half-slab g: operands in (fa0, fb0).  Half-slab g+1 must have landed: outstanding groups of this
            // wave, oldest first, are g+1, g+2 (+ younger epilogue stores at a tile start - waiting for them too is
            // safe, only slower).
            asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
            CAP_RAW_BARRIER();                 // everybody's pieces of g+1 landed; stage (g+3)&3 = (g-1)&3 is free
            issue_next(g + 3);
            if constexpr (SCHED == 1) __builtin_amdgcn_iglp_opt(1);
            load_frags(smem + ((g + 1) & 3) * HSTAGE, fa1, fb1);
#pragma unroll
            for (int i = 0; i < MI; ++i)
#pragma unroll
                for (int j = 0; j < NI; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fb0[j], fa0[i], acc[i][j], 0, 0, 0);
            // ---- odd half-slab g+1: operands in (fa1, fb1); half-slab g+2 must have landed (it may be the next tile's)
            asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
            CAP_RAW_BARRIER();
            issue_next(g + 4);
            if constexpr (SCHED == 1) __builtin_amdgcn_iglp_opt(1);
            load_frags(smem + ((g + 2) & 3) * HSTAGE, fa0, fb0);
#pragma unroll
            for (int i = 0; i < MI; ++i)
#pragma unroll
                for (int j = 0; j < NI; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fb1[j], fa1[i], acc[i][j], 0, 0, 0);
        }
        // bias of this tile: fetched with the tile's first half-slab (slot tcount & 1), long landed; the next tile's
        // bias goes to the other slot now (one operation per wave, see issue_bias)
        if (tile + nl < c1) issue_bias(tile + nl, (tcount + 1) & 1);
        else issue_bias(tile, (tcount + 1) & 1);
        big2_epilogue<bf16_t, OUT_F32, EPI, MI, NI>(p, acc, smem + 4 * HSTAGE + 2048 + 8 * 1024 + 2048 + wave * (16 * 144),
                                            has_bias ? bias_lds + (tcount & 1) * 1024 + wn0 * 4 : nullptr, m0 + wm0, n0 + wn0, lane);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // drain the dummy groups before the LDS goes away
}

// ------------------------------------------------------------------------------------------------------------------
// Decode-sized GEMM ("rows": a few hundred activation rows against a weight matrix that is read once).  These launches are
// not bound by flops or by HBM but by the serial chain load -> LDS -> barrier -> fragments -> MFMA of a small tile: the
// register-staged 64x64 tile (gemm_kernel) spends one barrier per 128-byte K-slab for 12 MFMAs per wave.  Here a workgroup
// still owns a 64x64 output tile (the decomposition with the least operand traffic per CU when ~200 workgroups are wanted:
// (64 + 64) rows per slab), but its K range is split over its FOUR WAVES, each with a PRIVATE pipeline:
//   * wave w takes the block's slabs w, w + 4, ... and accumulates the whole 64x64 tile over them (48 MFMAs per slab for
//     16 ds_read_b128 of fragments: half the LDS bytes per MFMA of the 2x2 wave grid);
//   * its slabs arrive by LDS-DMA into a wave-private ring of two 16 KiB buffers (A rows then W rows, XOR chunk swizzle on
//     the source address); the only ordering is the wave's own counted vmcnt - NO workgroup barrier in the loop, a wave
//     starts its MFMAs when ITS first slab lands and the second slab's DMA runs under them;
//   * operands swapped (W fragment as the MFMA's A): a lane's 4 accumulators are 4 consecutive output columns;
//   * one barrier per launch: the four partial tiles meet in LDS (each wave parks its tile in its own ring), are summed in
//     wave order 0..3 - a fixed order: the result depends on (N, K, split) only, never on the row count - and leave through
//     the common epilogue (scale / bias / GELU / G8 or fp32 or split-K slab) as whole 256-byte rows.
// Block ids: XCD-aware remap, row tile fastest (see gemm_kernel).  T = g8_t (32 k per slab, 3 MFMAs per product) or bf16
// (64 k per slab).
// AP = 8-row pieces of A a slab brings in: 8 (64 rows), 4 or 2 when the launch has at most 32 / 16 rows in all (one image, a
// few dozen rows of the OPT decoder): the rows beyond M are clamped duplicates whose traffic - as much as the W tile's - and
// MFMAs would be wasted.  A row's sums are the same for every AP (same slabs, same order): batch invariance is not touched.
template <typename T, bool OUT_F32, int EPI, int AP = 8>
__global__ __launch_bounds__(256, 1) void gemm_rows_kernel(GemmParams p) {
    using vec = typename Mma<T>::vec;
    constexpr bool G8 = is_g8<T>;
    constexpr int EPC = Mma<T>::EPC, SLAB = 8 * EPC;      // K elements per 128-byte slab row
    constexpr int BUF = 128 * 128, RING = 2 * BUF;        // one slab of the tile: 64 A rows + 64 W rows; two per wave
    constexpr int PITCH = 272;                            // reduce image: 64 rows x 256 payload bytes, pitch 272 (conflict-free b128)
    static_assert(64 * PITCH <= RING, "the partial tile must fit in the wave's ring");
    extern __shared__ __attribute__((aligned(16))) char smem[];

    const int S = EPI == EPI_PARTIAL ? p.splitk : 1;
    // compacted decode loop (GemmParams::m_live): the open captions are the first *m_live rows.  The grid is the host's (every
    // row tile of M); the workgroups are numbered over the LIVE row tiles only, so the ones without work are the highest block
    // ids - dispatched last, behind every workgroup that has a tile - and return at once.  (Numbered over all row tiles with
    // the dead ones returning in place, row tile fastest, a dead workgroup sat between every few live ones in the dispatch
    // order and held a CU - 128 KiB of LDS - for its launch and one scalar load each.)
    const int m_rows = p.m_live ? min(p.M, *p.m_live) : p.M;
    const int ntm = (m_rows + 63) / 64, ntn = (p.N + 63) / 64, nwg = ntm * ntn * S;
    int bid = blockIdx.x;
    if (bid >= nwg) return;
    {
        int q = nwg >> 3, r = nwg & 7, xcd = bid & 7, idx = bid >> 3;
        bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
    }
    const int tm = bid % ntm;
    bid /= ntm;
    const int tn = bid % ntn, kz = bid / ntn;
    if constexpr (EPI == EPI_PARTIAL) p.p3 = kz;
    const int m0 = tm * 64, n0 = tn * 64;
    const int nkb = p.K / S / SLAB;                        // slabs of this block

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int r16 = lane & 15, kg = lane >> 4;
    char* ring = smem + wave * RING;

    // this lane's 8 A and 8 W source pointers (slab 0 of the block): piece q = rows 8q .. 8q+7, lane = (row, 16-byte position)
    const int prow = lane >> 3, ppos = lane & 7;
    const char* pa[8];
    const char* pb[8];
#pragma unroll
    for (int q = 0; q < 8; ++q) {
        const int row = q * 8 + prow;
        const int gch = ppos ^ swz_key<is_g8<T>>(row);
        const int ga = min(m0 + row, p.M - 1), gb = min(n0 + row, p.N - 1);
        pa[q] = (const char*)((const T*)p.A + (size_t)ga * p.lda + (size_t)kz * (p.K / S) + gch * EPC);
        pb[q] = (const char*)((const T*)p.W + (size_t)gb * p.ldw + (size_t)kz * (p.K / S) + gch * EPC);
    }
    auto issue = [&](int kt, char* buf) {                  // 8 + AP DMA instructions: the slab's W rows and its live A rows
#pragma unroll
        for (int q = 0; q < 8; ++q) {
            if (q < AP) __builtin_amdgcn_global_load_lds(CAP_GPTR(pa[q] + (size_t)kt * 128), CAP_LPTR(buf + q * 1024), 16, 0, 0);
            __builtin_amdgcn_global_load_lds(CAP_GPTR(pb[q] + (size_t)kt * 128), CAP_LPTR(buf + 64 * 128 + q * 1024), 16, 0, 0);
        }
    };
    constexpr int MIA = (AP + 1) / 2;                      // 16-row blocks of A that hold rows of the problem

    f32x4 acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = 0.f;

    const int n = (nkb - wave + 3) >> 2;                   // slabs of this wave: wave, wave + 4, ...
    if (n > 0) issue(wave, ring);
    if (n > 1) issue(wave + 4, ring + BUF);
    for (int t = 0; t < n; ++t) {
        // the wave's own DMA is all that writes its ring: slab t has landed once at most the next slab's 16 pieces are open
        if (t + 1 < n) {
            if constexpr (AP == 8) asm volatile("s_waitcnt vmcnt(16)" ::: "memory");
            else if constexpr (AP == 4) asm volatile("s_waitcnt vmcnt(12)" ::: "memory");
            else asm volatile("s_waitcnt vmcnt(10)" ::: "memory");
        } else {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        const char* a_s = ring + (t & 1) * BUF;
        const char* b_s = a_s + 64 * 128;
        if constexpr (G8) {
            vec ah[4], al[4], bh[4], bl[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                bh[j] = *(const vec*)(b_s + swz_off<is_g8<T>>(j * 16 + r16, 2 * kg));
                bl[j] = *(const vec*)(b_s + swz_off<is_g8<T>>(j * 16 + r16, 2 * kg + 1));
            }
#pragma unroll
            for (int i = 0; i < MIA; ++i) {
                ah[i] = *(const vec*)(a_s + swz_off<is_g8<T>>(i * 16 + r16, 2 * kg));
                al[i] = *(const vec*)(a_s + swz_off<is_g8<T>>(i * 16 + r16, 2 * kg + 1));
            }
            // fragments are in registers: the buffer may take the slab after next
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            if (t + 2 < n) issue(wave + 4 * (t + 2), ring + (t & 1) * BUF);
#pragma unroll
            for (int i = 0; i < MIA; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(bl[j], ah[i], acc[i][j], 0, 0, 0);
                    acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(bh[j], al[i], acc[i][j], 0, 0, 0);
                    acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(bh[j], ah[i], acc[i][j], 0, 0, 0);
                }
        } else {
            vec af[2][4], bf[2][4];
#pragma unroll
            for (int ks = 0; ks < 2; ++ks)
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    bf[ks][j] = *(const vec*)(b_s + swz_off<is_g8<T>>(j * 16 + r16, ks * 4 + kg));
                    if (j < MIA) af[ks][j] = *(const vec*)(a_s + swz_off<is_g8<T>>(j * 16 + r16, ks * 4 + kg));
                }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            if (t + 2 < n) issue(wave + 4 * (t + 2), ring + (t & 1) * BUF);
#pragma unroll
            for (int ks = 0; ks < 2; ++ks)
#pragma unroll
                for (int i = 0; i < MIA; ++i)
#pragma unroll
                    for (int j = 0; j < 4; ++j)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bf[ks][j], af[ks][i], acc[i][j], 0, 0, 0);
        }
    }
    // park the partial tile in the wave's own ring (its DMA and its reads are done): C[16 i + r16][16 j + 4 kg .. + 3]
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) *(f32x4*)(ring + (i * 16 + r16) * PITCH + (j * 16 + 4 * kg) * 4) = acc[i][j];
    __syncthreads();
    // sum the four partial tiles (wave order) and store: thread -> 16 bytes of a row, 16 threads per 256-byte row
    const int c4 = tid & 15, col = n0 + c4 * 4;
    f32x4 biasv = 0.f;
    if (EPI != EPI_PARTIAL && p.bias && col < p.N) biasv = *(const f32x4*)(p.bias + col);
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const int rr = (tid >> 4) + q * 16, row = m0 + rr;
        const char* src = smem + rr * PITCH + c4 * 16;
        f32x4 v = *(const f32x4*)src;
        v += *(const f32x4*)(src + RING);
        v += *(const f32x4*)(src + 2 * RING);
        v += *(const f32x4*)(src + 3 * RING);
        if (row < p.M && col < p.N) epi_store4<T, OUT_F32, EPI>(p, row, col, v, biasv);
    }
}

template <typename T, bool OUT_F32, int EPI, int AP>
int launch_rows_ap(const GemmParams& p, hipStream_t stream) {
    constexpr int LDS = 4 * 2 * 128 * 128;               // four wave-private rings of two 16 KiB slabs
    auto kern = gemm_rows_kernel<T, OUT_F32, EPI, AP>;
    if (cap_kernel_setup((const void*)kern, LDS, nullptr) != 0) return -1;
    const int S = EPI == EPI_PARTIAL ? p.splitk : 1;
    const int grid = ((p.M + 63) / 64) * ((p.N + 63) / 64) * S;
    hipLaunchKernelGGL(kern, dim3(grid), dim3(256), LDS, stream, p);
    CAP_HIP_CHECK(hipGetLastError());
    return 0;
}
template <typename T, bool OUT_F32, int EPI>
int launch_rows(const GemmParams& p, hipStream_t stream) {
    if (p.M <= 16) return launch_rows_ap<T, OUT_F32, EPI, 2>(p, stream);
    if (p.M <= 32) return launch_rows_ap<T, OUT_F32, EPI, 4>(p, stream);
    return launch_rows_ap<T, OUT_F32, EPI, 8>(p, stream);
}

template <bool OUT_F32, int EPI, int SCHED = 0>
int launch_big3(const GemmParams& p, hipStream_t stream) {
    // four half-slab stages, bias ping-pong (2 KiB) + per-wave bias scratch (8 KiB), dummy DMA sink (2 KiB), strips
    constexpr int LDS = 4 * 512 * 64 + 2048 + 8 * 1024 + 2048 + 8 * 16 * 144;
    auto kern = gemm_big3_kernel<OUT_F32, EPI, SCHED>;
    int n_cu = 0;
    if (cap_kernel_setup((const void*)kern, LDS, &n_cu) != 0) return -1;
    const int ntiles = ((p.M + 255) / 256) * ((p.N + 255) / 256);
    const int grid = ntiles < n_cu ? ntiles : n_cu;
    hipLaunchKernelGGL(kern, dim3(grid), dim3(512), LDS, stream, p);
    CAP_HIP_CHECK(hipGetLastError());
    return 0;
}

// Tiles go round-robin over one workgroup per CU.  When the last round is at most half full, its tiles are cut into their
// upper and lower 128 rows and run as a second launch of the half-tile instantiation on twice as many workgroups: the cut is
// along M, so every output element is the same sum as before (bit-identical: tests/test_split_gpu.py), and the tail round costs
// ~0.7 of a tile time instead of a whole one (a half tile moves 48 KiB per stage for half the MFMAs: it runs at the LDS-DMA fill
// rate).  ViT-B/16 at batch 256: 197 x 3 tiles of N = 768 over 256 CUs are 2.31 rounds - 3 before, ~2.7 now (fc2 -10 %, proj -6 %).
template <typename T, bool OUT_F32, int EPI, int VAR, bool PROF, int NWM = 2, int NWN = 4>
int launch_big2(const GemmParams& p, hipStream_t stream) {
    constexpr int LDS = 2 * 512 * 128 + 2 * 1024 + 8 * 16 * 144;   // two stages + bias ping-pong + epilogue strips
    constexpr int LDS_H = 2 * 384 * 128 + 2 * 1024 + 8 * 16 * 144;
    auto kern = gemm_big2_kernel<T, OUT_F32, EPI, VAR, PROF, NWM, NWN>;
    int n_cu = 0;
    if (cap_kernel_setup((const void*)kern, LDS, &n_cu) != 0) return -1;
#ifdef CAP_EXPERIMENTS      // power probe: the same kernel on a part of the chip (tools/bench_gemm_split.py --cus)
    if (const char* e = getenv("CAP_EXP_CUS")) n_cu = std::min(n_cu, std::max(8, atoi(e)));
#endif
    const int ntiles = ((p.M + 255) / 256) * ((p.N + 255) / 256);
    const int rounds = ntiles / n_cu, tail = ntiles - rounds * n_cu;
    if constexpr (NWM * NWN == 8 && !PROF) {
        // only where a round is a large share of the launch: with many rounds (fc1: 9 + a 0.23 tail, cross-K/V: 55) the second
        // launch's boundary and ramp cost what the shorter tail saves (measured: fc1 688 -> 699 us with the cut, fc2 683 -> 613)
        if (rounds >= 1 && rounds <= 4 && tail > 0 && 2 * tail <= n_cu) {
            auto kern_h = gemm_big2_kernel<T, OUT_F32, EPI, VAR, PROF, NWM, NWN, 128>;
            if (cap_kernel_setup((const void*)kern_h, LDS_H, nullptr) != 0) return -1;
            GemmParams q = p;
            q.tile0 = 0; q.tile1 = rounds * n_cu;
            hipLaunchKernelGGL(kern, dim3(n_cu), dim3(NWM * NWN * 64), LDS, stream, q);
            q.tile0 = rounds * n_cu; q.tile1 = ntiles;
            hipLaunchKernelGGL(kern_h, dim3(2 * tail), dim3(NWM * NWN * 64), LDS_H, stream, q);
            CAP_HIP_CHECK(hipGetLastError());
            return 0;
        }
    }
    const int grid = ntiles < n_cu ? ntiles : n_cu;
    hipLaunchKernelGGL(kern, dim3(grid), dim3(NWM * NWN * 64), LDS, stream, p);
    CAP_HIP_CHECK(hipGetLastError());
    return 0;
}

template <typename T, bool OUT_F32, int EPI, int VAR = 0>
int launch_big(const GemmParams& p, hipStream_t stream) {
    constexpr int LDS = 2 * 512 * 128 + 2 * 1024;       // two stages + bias ping-pong
    auto kern = gemm_big_kernel<T, OUT_F32, EPI, VAR>;
    int n_cu = 0;
    if (cap_kernel_setup((const void*)kern, LDS, &n_cu) != 0) return -1;
    const int ntiles = ((p.M + 255) / 256) * ((p.N + 255) / 256);
    const int grid = ntiles < n_cu ? ntiles : n_cu;
    hipLaunchKernelGGL(kern, dim3(grid), dim3(512), LDS, stream, p);
    CAP_HIP_CHECK(hipGetLastError());
    return 0;
}

template <typename T, int BM, int BN, int WM, int WN, int D, int WPE, bool OUT_F32, int EPI>
int launch_cfg(const GemmParams& p, hipStream_t stream) {
    constexpr int NT = (BM / WM) * (BN / WN) * 64;
    constexpr int LDS = 2 * (BM + BN) * 128;
    static_assert((BM / WM) * (BN / WN) * 32 * WN * 4 <= LDS, "epilogue strips must fit in the staging buffers");
    const int ntm = (p.M + BM - 1) / BM, ntn = (p.N + BN - 1) / BN;
    auto kern = gemm_kernel<T, BM, BN, WM, WN, D, WPE, OUT_F32, EPI>;
    if (cap_kernel_setup((const void*)kern, LDS, nullptr) != 0) return -1;
    const int S = EPI == EPI_PARTIAL ? p.splitk : 1;
    hipLaunchKernelGGL(kern, dim3(ntm * ntn * S), dim3(NT), LDS, stream, p);
    CAP_HIP_CHECK(hipGetLastError());
    return 0;
}

template <typename T, bool OUT_F32, int EPI>
int launch_tile(const GemmParams& p, int tile, int nk, hipStream_t stream) {
#ifdef CAP_EXPERIMENTS
    // split fp16: gemm_big2_kernel<g8_t>, the kernel gemm_pp.hip replaced, exists in experiments builds only - 10 = its schedule
    // (iglp_opt(0)) as the A/B partner of tools/bench_gemm_pp.py, 13 = the same with cycle stamps to p.aux
    if (tile == 10 || tile == 13) {
        if constexpr (is_g8<T>) {
            if (p.K >= 64 && !p.resid) {
                if constexpr (!OUT_F32 && EPI == EPI_STORE) {
                    if (tile == 13) return launch_big2<T, OUT_F32, EPI, 1, true>(p, stream);
                }
                return launch_big2<T, OUT_F32, EPI, 1, false>(p, stream);
            }
        }
    }
#endif
#ifdef CAP_EXPERIMENTS
    // bf16 second generation (gemm_big2_kernel) and the half-slab structure (gemm_big3_kernel): the kernels gemm_pp.hip replaced,
    // experiments builds only - A/B partners of tools/bench_gemm_pp.py --bf16 (12 / 14) and tools/gemm_cycles.py, 13 = instrumented
    if (tile >= 10 && tile <= 15) {
        if constexpr (sizeof(T) == 2) {
            if (p.K >= 128 && !p.resid) {
                if (tile == 10) return launch_big2<T, OUT_F32, EPI, 0, false>(p, stream);
                if (tile == 11) return launch_big2<T, OUT_F32, EPI, 1, false>(p, stream);
                if (tile == 12) return launch_big2<T, OUT_F32, EPI, 2, false>(p, stream);
                if (tile == 14) return launch_big3<OUT_F32, EPI>(p, stream);
                if (tile == 15) return launch_big3<OUT_F32, EPI, 1>(p, stream);
                if constexpr (!OUT_F32 && EPI == EPI_STORE) return launch_big2<T, OUT_F32, EPI, 2, true>(p, stream);
            }
        }
    }
#endif
    if (tile >= 10 && tile <= 15) tile = 3;
    if (tile == 9) {                                    // instrumented main loop: per-wave cycle counts to p.aux
#ifdef CAP_EXPERIMENTS
        if constexpr (sizeof(T) == 2 && !is_g8<T> && !OUT_F32 && EPI == EPI_STORE) return launch_big<T, OUT_F32, EPI, 4>(p, stream);
#endif
        tile = 3;
    }
    if (tile == 6) {                                    // decode "rows" kernel: wave-private K pipelines (gemm_rows_kernel)
        if constexpr ((sizeof(T) == 2 || is_g8<T>) && (EPI == EPI_STORE || EPI == EPI_PARTIAL)) return launch_rows<T, OUT_F32, EPI>(p, stream);
        tile = 2;                                       // fp32 mode / cache-scatter epilogues: the register-staged 64x64 tile
    }
    if ((tile == 3 || tile == 5) && p.resid) tile = 4;  // the LDS-DMA kernels have no residual operand
    if (tile == 3) {
        // bf16: launch_gemm sends tile 3 to gemm_pp.hip; what that kernel does not take (K < 128, operands beyond 4 GB, an odd
        // N) runs on the first-generation LDS-DMA kernel
        if constexpr (sizeof(T) == 2) {
            return launch_big<T, OUT_F32, EPI, 3>(p, stream);
        } else if constexpr (is_g8<T>) {
            // split fp16: launch_gemm sends tile 3 to gemm_pp.hip; what that kernel does not take (K < 64, operands beyond 4 GB)
            // runs on the register-staged 256x256 tile
            return launch_cfg<T, 256, 256, 128, 64, 1, 2, OUT_F32, EPI>(p, stream);
        } else {
            return launch_big<T, OUT_F32, EPI, 0>(p, stream);
        }
    }
    if (tile == 5) {                                    // first-generation kernel, kept for A/B
        if constexpr (sizeof(T) == 2) return launch_big<T, OUT_F32, EPI, 3>(p, stream);
        else if constexpr (is_g8<T>) return launch_cfg<T, 256, 256, 128, 64, 1, 2, OUT_F32, EPI>(p, stream);
        else return launch_big<T, OUT_F32, EPI, 0>(p, stream);
    }
    if (tile == 4) return launch_cfg<T, 256, 256, 128, 64, 1, 2, OUT_F32, EPI>(p, stream);
    if (tile == 1) {
        if (nk % 2 == 0) return launch_cfg<T, 128, 128, 64, 64, 2, 2, OUT_F32, EPI>(p, stream);
        return launch_cfg<T, 128, 128, 64, 64, 1, 2, OUT_F32, EPI>(p, stream);
    }
    // decode-sized GEMMs are bound by the global-load round trip: keep as many slabs in flight as divide nk
    if (nk % 6 == 0) return launch_cfg<T, 64, 64, 32, 32, 6, 3, OUT_F32, EPI>(p, stream);
    if (nk % 4 == 0) return launch_cfg<T, 64, 64, 32, 32, 4, 4, OUT_F32, EPI>(p, stream);
    if (nk % 3 == 0) return launch_cfg<T, 64, 64, 32, 32, 3, 4, OUT_F32, EPI>(p, stream);
    if (nk % 2 == 0) return launch_cfg<T, 64, 64, 32, 32, 2, 4, OUT_F32, EPI>(p, stream);
    return launch_cfg<T, 64, 64, 32, 32, 1, 4, OUT_F32, EPI>(p, stream);
}

template <typename T>
int launch_t(const GemmParams& p, int tile, hipStream_t stream) {
    const int nk = p.K / (8 * Mma<T>::EPC) / (p.epi == EPI_PARTIAL ? p.splitk : 1);
    switch (p.epi) {
        case EPI_PARTIAL: return launch_tile<T, true, EPI_PARTIAL>(p, tile, nk, stream);
        case EPI_STORE:
            return p.out_f32 ? launch_tile<T, true, EPI_STORE>(p, tile, nk, stream)
                             : launch_tile<T, false, EPI_STORE>(p, tile, nk, stream);
        case EPI_PATCH: return launch_tile<T, true, EPI_PATCH>(p, tile, nk, stream);
        case EPI_CROSSKV: return launch_tile<T, is_g8<T>, EPI_CROSSKV>(p, tile, nk, stream);     // split mode: the K/V caches
        case EPI_QKVCACHE: return launch_tile<T, is_g8<T>, EPI_QKVCACHE>(p, tile, nk, stream);   // are fp32
    }
    cap_set_error("launch_gemm: unknown epilogue %d", p.epi);
    return -1;
}

}  // namespace

int launch_gemm(int dtype, const GemmParams& p, int tile, hipStream_t stream) {
    const int slab = dtype == CAP_DT_BF16 ? 64 : 32;
    const int esz = dtype == CAP_DT_BF16 ? 2 : 4;
    if (dtype == CAP_DT_G8 && (p.lda % 8 != 0 || p.ldw % 8 != 0 || (p.epi == EPI_STORE && !p.out_f32 && (p.N % 8 != 0 || p.ldc % 8 != 0)))) {
        cap_set_error("launch_gemm: G8 operands need lda / ldw (and N / ldc of a G8 output) to be multiples of 8 (lda=%d ldw=%d N=%d ldc=%d)",
                      p.lda, p.ldw, p.N, p.ldc);
        return -1;
    }
    if (p.epi == EPI_PARTIAL && (p.splitk < 1 || p.K % (slab * p.splitk) != 0)) {
        cap_set_error("launch_gemm: split-K %d does not divide K=%d into whole %d-element slabs", p.splitk, p.K, slab);
        return -1;
    }
    if (p.M <= 0 || p.N <= 0 || p.K <= 0 || p.K % slab != 0) {
        cap_set_error("launch_gemm: bad shape M=%d N=%d K=%d (K must be a multiple of %d)", p.M, p.N, p.K, slab);
        return -1;
    }
    if ((p.lda * esz) % 16 != 0 || (p.ldw * esz) % 16 != 0 || ((uintptr_t)p.A & 15) || ((uintptr_t)p.W & 15)) {
        cap_set_error("launch_gemm: operands must be 16-byte aligned (lda=%d ldw=%d)", p.lda, p.ldw);
        return -1;
    }
    if (p.N % 4 != 0 || (p.epi == EPI_STORE && p.ldc % 4 != 0) || (p.resid && p.ldr % 4 != 0)) {
        cap_set_error("launch_gemm: N / ldc / ldr must be multiples of 4 (N=%d ldc=%d ldr=%d)", p.N, p.ldc, p.ldr);
        return -1;
    }
    // The LDS-DMA kernels fetch a tile's bias with 16-byte DMA reads from bias + min(col, N - 4): N % 4 == 0 (above) keeps
    // the clamped address inside the vector and 16-byte aligned - provided the vector itself is.  The epilogues read bias
    // and residual rows as 16-byte vectors too.
    if (((uintptr_t)p.bias & 15) || ((uintptr_t)p.resid & 15) || ((uintptr_t)p.C & 15)) {
        cap_set_error("launch_gemm: bias / resid / C must be 16-byte aligned");
        return -1;
    }
    if (tile == 0) {
        // 256x256 (one 8-wave block per CU) once at least half the CUs get a tile: between 128 and 255 such tiles the
        // 128x128 kernel would need two rounds of its 512 resident blocks (measured on the OPT prefill, 1056 x 7680 x 2560:
        // 99 us against one round of the big kernel); 128x128 while that still gives every CU work; 64x64 with a deep
        // register prefetch ring for the decode-sized (M <= a few hundred) GEMMs
        const long t256 = (long)((p.M + 255) / 256) * ((p.N + 255) / 256);
        const long t128 = (long)((p.M + 127) / 128) * ((p.N + 127) / 128);
        tile = (t256 >= 256 || (t256 >= 128 && p.M >= 256)) ? 3 : (t128 >= 256 ? 1 : 2);   // few rows: a 256-row tile is mostly padding
        // split fp16 with a long K (the OPT prefill's out_proj / fc2: 1056 x 2560 x 2560 / 10240 with the residual operand): the
        // 128x128 tile already pays at half a round of workgroups - 84 / 271 us against 97 / 371 for the 64x64 tile; at K <= 1024
        // and for bf16 the small tile stays level or ahead (tools/bench_prefill_gemm.py).  Same bits either way.
        if (dtype == CAP_DT_G8 && tile == 2 && t128 >= 128 && p.K >= 2048) tile = 1;
    }
    if (p.epi == EPI_CROSSKV && p.kv16) {               // int16 rows with one scale each: only gemm_pp.hip's epilogue builds them
        const int rc = dtype == CAP_DT_G8 ? launch_gemm_pp(dtype, p, false, stream) : -2;
        if (rc == -2) cap_set_error("launch_gemm: a KV16 cross-K/V cache needs G8 operands, K >= 64 and operands below 4 GB (K=%d)", p.K);
        return rc == -2 ? -1 : rc;
    }
    // split fp16 and bf16, 256x256: the kernel with the wave groups half a stage apart (gemm_pp.hip) wherever it takes the shape;
    // what it declines (K below two stages, operands beyond 4 GB, a residual operand) runs on the older 256x256 kernels
    if ((dtype == CAP_DT_G8 || dtype == CAP_DT_BF16) && (tile == 3 || tile == 20 || tile == 21)) {
        const int rc = launch_gemm_pp(dtype, p, tile == 21, stream);
        if (rc != -2) return rc;
        tile = 3;
    }
    if (dtype == CAP_DT_BF16) return launch_t<bf16_t>(p, tile, stream);
    if (dtype == CAP_DT_F32) return launch_t<float>(p, tile, stream);
    if (dtype == CAP_DT_G8) return launch_t<g8_t>(p, tile, stream);
    cap_set_error("launch_gemm: unknown dtype %d", dtype);
    return -1;
}

CAP_DEFINE_G8_CLAMP_READER(cap_g8_clamped_gemm)
