// Weight-streaming GEMM for a handful of rows (the cached decode step of a large language model: BLIP-2's OPT-2.7b,
// 32 rows against 13-52 MB of weights per projection).  C[M, N] = A[M, K] . W[N, K]^T, bf16 operands, fp32 accumulation.
//
// The operation is bound by reading W once from HBM, so the kernel is built around that stream and nothing else:
//   * a workgroup owns 32 weight rows and a K range; its waves split that range (at most 320 k each), so a wave's whole
//     share is in flight at once - one memory round trip;
//   * the activations (a few rows, L2-resident) go straight to registers in the MFMA operand layout
//     (v_mfma_f32_16x16x32_bf16, W as the A operand so a lane ends up with 4 consecutive output columns);
//   * the waves' fp32 tiles are summed through LDS in wave order (deterministic), then either finished in place
//     (bias, ReLU/GELU, cast - one slice) or written as split-K slice `z` for the reduce+LayerNorm consumer.
// Rows beyond 32 are further workgroups placed on the same XCD as the first (W is re-read from that L2); the k-order of a
// row's sum depends on (N, K) and the output form only, never on the row count.
#include "common.h"
#include "ops.h"
#include <stdlib.h>

#include <algorithm>

namespace {

typedef float f32x2 __attribute__((ext_vector_type(2)));

constexpr int SK_MAXKS = 10;   // LDS row pitch in floats (144 B: the 16 rows of a tile spread over the banks)

struct SkinnyParams {
    const bf16_t* A; int lda;
    const bf16_t* W; int ldw;
    const float* bias;        // finished output only
    bf16_t* out; int ldc;     // part == null (S == 1): act(sum + bias) as bf16
    float* part;              // non-null: part[z][M][N] fp32 slice sums instead of `out`
    int M, N, S, nks, act;
    int G, units;             // row groups of 32 (ceil(M / 32)); units = row tiles x S    // nks = k-steps of 32 per wave; act 0 none, 1 GELU(erf), 2 ReLU
};

// W is staged through LDS: loading it straight into the MFMA operand layout asks for 64 bytes per weight row per
// instruction (4 lanes per row), and half-line requests cap the stream near 2 TB/s (measured, tools/skinny_bench.py); a
// wave's LDS-DMA instruction moves 8 rows x 128 bytes (whole lines) into a wave-private ring of 64-wide slabs instead, and
// the fragments are read back through the XOR swizzle.  No workgroup barrier in the stream: a wave only reads what it
// fetched itself, ordered by counted s_waitcnt vmcnt.
// (key: the bf16 one of gemm_tile.h::swz_key - conflict-free for the lane groups gfx950 serves a ds_read_b128 in)
__device__ __forceinline__ int skl_key(int row) { const int t = (row >> 1) & 7; return t ^ (((t >> 1) ^ (t >> 2)) & 1); }
__device__ __forceinline__ int skl_swz(int row, int chunk) { return row * 128 + ((chunk ^ skl_key(row)) << 4); }
#define SKL_GPTR(p) ((const __attribute__((address_space(1))) void*)(p))
#define SKL_LPTR(p) ((__attribute__((address_space(3))) void*)(p))

// NW waves; SKL_RING 64-k slabs (TR rows x 128 B) in flight per wave; TR = weight rows per workgroup (32, or 40 when that
// makes the grid exactly one workgroup per CU: the third 16-row MFMA tile is then half empty, which costs nothing here).
template <int NW, int SKL_RING, int TR>
__global__ __launch_bounds__(NW * 64) void gemm_skinny_kernel(SkinnyParams p) {
    constexpr int QN = TR / 8, WT = (TR + 15) / 16, SLAB = TR * 128, PITCH = WT * 16 + 4;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63, r16 = lane & 15, kg = lane >> 4;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    // Workgroup id -> (unit = row tile x K slice, row group g of 32 activation rows).  Ids are dealt round-robin to the 8
    // XCDs, so the G groups of one unit take ids 8 apart: same XCD, dispatched together - the unit's weights come from HBM
    // once and the other groups find them in that XCD's L2.
    const int per = 8 * p.G, rr = blockIdx.x % per;
    const int unit = (blockIdx.x / per) * 8 + (rr & 7);
    if (unit >= p.units) return;
    const int nt = unit / p.S, z = unit % p.S;
    const int n0 = nt * TR, m0 = (rr >> 3) * 32;
    const int kw = (z * NW + wave) * p.nks * 32;                    // first k of this wave
    const int nsl = p.nks >> 1;                                     // slabs of 64 k
    char* ring = smem + wave * (SKL_RING * SLAB);

    // activations: every fragment of the wave's K range, straight to registers (L2 hits)
    const bf16_t* ap[2];
#pragma unroll
    for (int t = 0; t < 2; ++t) ap[t] = p.A + (size_t)min(m0 + t * 16 + r16, p.M - 1) * p.lda + kw + kg * 8;
    bf16x8 af[2][SK_MAXKS];
#pragma unroll
    for (int j = 0; j < SK_MAXKS; ++j)
        if (j < p.nks) {
#pragma unroll
            for (int t = 0; t < 2; ++t) af[t][j] = *(const bf16x8*)(ap[t] + j * 32);
        }
    // weights: lane -> (row q*8 + lane/8, 16-byte position lane%8); the swizzle is applied on the source side
    const bf16_t* wsrc[QN];
#pragma unroll
    for (int q = 0; q < QN; ++q) {
        const int row = q * 8 + (lane >> 3);
        wsrc[q] = p.W + (size_t)(n0 + row) * p.ldw + kw + (((lane & 7) ^ skl_key(row)) << 3);
    }
    auto issue = [&](int sl) {
        char* dst = ring + (sl % SKL_RING) * SLAB;
#pragma unroll
        for (int q = 0; q < QN; ++q) __builtin_amdgcn_global_load_lds(SKL_GPTR(wsrc[q] + sl * 64), SKL_LPTR(dst + q * 1024), 16, 0, 0);
    };
#pragma unroll
    for (int sl = 0; sl < SKL_RING; ++sl)
        if (sl < nsl) issue(sl);

    f32x4 acc[WT][2];
#pragma unroll
    for (int a = 0; a < WT; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b) acc[a][b] = f32x4(0.f);
#pragma unroll
    for (int sl = 0; sl < SK_MAXKS / 2; ++sl)
        if (sl < nsl) {
            // slabs still allowed in flight behind slab `sl`: those issued so far minus sl + 1
            const int behind = min(nsl, sl + SKL_RING) - sl - 1;
            static_assert(QN == 4 || QN == 5, "vmcnt ladder below is written for 4 or 5 DMA instructions per slab");
            if constexpr (QN == 4) {
                if (behind >= 3) asm volatile("s_waitcnt vmcnt(12)" ::: "memory");
                else if (behind == 2) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
                else if (behind == 1) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
                else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            } else {
                if (behind >= 3) asm volatile("s_waitcnt vmcnt(15)" ::: "memory");
                else if (behind == 2) asm volatile("s_waitcnt vmcnt(10)" ::: "memory");
                else if (behind == 1) asm volatile("s_waitcnt vmcnt(5)" ::: "memory");
                else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            }
            const char* buf = ring + (sl % SKL_RING) * SLAB;
            bf16x8 wf[WT][2];
#pragma unroll
            for (int ks = 0; ks < 2; ++ks)
#pragma unroll
                for (int t = 0; t < WT; ++t) wf[t][ks] = *(const bf16x8*)(buf + skl_swz(min(t * 16 + r16, TR - 1), ks * 4 + kg));
#pragma unroll
            for (int ks = 0; ks < 2; ++ks)
#pragma unroll
                for (int wt = 0; wt < WT; ++wt)
#pragma unroll
                    for (int xt = 0; xt < 2; ++xt)
                        acc[wt][xt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[wt][ks], af[xt][2 * sl + ks], acc[wt][xt], 0, 0, 0);
            if (sl + SKL_RING < nsl) {                  // refill the buffer just read (its ds_reads have returned: the MFMAs used them)
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                issue(sl + SKL_RING);
            }
        }
    __syncthreads();                                    // every wave is done with its ring: the sums reuse the space
    float* red = (float*)smem;
#pragma unroll
    for (int wt = 0; wt < WT; ++wt)
#pragma unroll
        for (int xt = 0; xt < 2; ++xt)
            *(f32x4*)(red + (wave * 32 + xt * 16 + r16) * PITCH + wt * 16 + 4 * kg) = acc[wt][xt];
    __syncthreads();
    for (int o = tid; o < 16 * TR; o += NW * 64) {
        const int m = o / (TR / 2), n = (o % (TR / 2)) * 2;
        f32x2 v = *(const f32x2*)(red + m * PITCH + n);
#pragma unroll
        for (int w = 1; w < NW; ++w) v += *(const f32x2*)(red + (w * 32 + m) * PITCH + n);
        if (m0 + m >= p.M) continue;
        if (p.part) {
            *(f32x2*)(p.part + ((size_t)z * p.M + m0 + m) * p.N + n0 + n) = v;
            continue;
        }
        if (p.bias) v += *(const f32x2*)(p.bias + n0 + n);
        if (p.act == 2) { v[0] = fmaxf(v[0], 0.f); v[1] = fmaxf(v[1], 0.f); }
        else if (p.act == 1) { v[0] = gelu_erf_fast(v[0]); v[1] = gelu_erf_fast(v[1]); }
        bf16x2 ob;
        ob[0] = (bf16_t)v[0]; ob[1] = (bf16_t)v[1];
        *(bf16x2*)(p.out + (size_t)(m0 + m) * p.ldc + n0 + n) = ob;
    }
}

}  // namespace

// Launch plan for a (N, K) projection: K slices S, waves per workgroup NW and weight rows per workgroup TR, with
// K / (S NW) <= 320 and a multiple of 64.  finished: the epilogue needs the whole K in one workgroup (S = 1).  Otherwise
// the fewest waves per row tile, split into more slices while the grid is small.  40-row workgroups when 32-row ones
// overflow the 256 CUs and 40-row ones fill them exactly or stay inside.  Returns S, or 0 when the shape does not fit.
int skinny_plan(int N, int K, bool finished, int* nw_out, int* tr_out, int M) {
    if (N % 8 != 0 || K % 64 != 0) return 0;
    int bestS = 0, bestNW = 0;
    for (int S = 1; S <= (finished ? 1 : 16); S *= 2)
        for (int nw : {8, 4}) {
            if (K % (S * nw * 64) != 0 || K / (S * nw * 32) > SK_MAXKS) continue;
            const bool fewer = bestS == 0 || S * nw < bestS * bestNW;
            const bool same_but_wider = bestS != 0 && S * nw == bestS * bestNW && (N / 32) * bestS < 128;
            if (fewer || same_but_wider) { bestS = S; bestNW = nw; }
        }
    if (bestS == 0) return 0;
    int tr = 32;
    // (the row count only picks the partition of N - S, NW and with them every row's k-order do not depend on it)
    if (N % 40 == 0 && bestNW == 8 && (N % 32 != 0 || (M <= 32 && (N / 32) * bestS > 256 && (N / 40) * bestS <= 256))) tr = 40;
    if (N % tr != 0) return 0;
    if (nw_out) *nw_out = bestNW;
    if (tr_out) *tr_out = tr;
    return bestS;
}

template <int NW, int RING, int TR>
static int skinny_launch(const SkinnyParams& p, dim3 grid, hipStream_t s) {
    constexpr int ring_b = NW * RING * TR * 128, red_b = NW * 32 * (((TR + 15) / 16) * 16 + 4) * 4;
    constexpr int lds = ring_b > red_b ? ring_b : red_b;
    auto kern = gemm_skinny_kernel<NW, RING, TR>;
    if (cap_kernel_setup((const void*)kern, lds, nullptr) != 0) return -1;
    hipLaunchKernelGGL(kern, grid, dim3(NW * 64), lds, s, p);
    CAP_HIP_CHECK(hipGetLastError());
    return 0;
}

// part == nullptr: out = act(A W^T + bias) as bf16 (plan with finished = true).  part != nullptr: part[z] = the S slice
// sums, fp32 (bias / act are the consumer's).  Returns the slice count (>= 1), or -1.
int launch_gemm_skinny(const void* A, int lda, const void* W, int ldw, const float* bias, int act, void* out, int ldc,
                       float* part, int M, int N, int K, hipStream_t s) {
    int nw = 0, tr = 32;
    const int S = skinny_plan(N, K, part == nullptr, &nw, &tr, M);
    if (S < 1 || M < 1 || (lda & 7) || (ldw & 7) || (!part && (!out || (ldc & 1)))) {
        cap_set_error("gemm_skinny: unsupported shape M=%d N=%d K=%d", M, N, K);
        return -1;
    }
    SkinnyParams p;
    p.A = (const bf16_t*)A; p.lda = lda; p.W = (const bf16_t*)W; p.ldw = ldw; p.bias = bias; p.out = (bf16_t*)out; p.ldc = ldc;
    p.part = part; p.M = M; p.N = N; p.S = S; p.nks = K / (nw * 32 * S); p.act = act;
    p.G = (M + 31) / 32; p.units = (N / tr) * S;
    const dim3 grid(((p.units + 7) / 8) * 8 * p.G);
    // more workgroups than CUs: a two-slab ring (64 KiB per workgroup) lets two share a CU, so the grid is still one round
    const bool small = (int)grid.x > 256;
    int rc;
    if (tr == 40) rc = skinny_launch<8, 3, 40>(p, grid, s);
    else if (nw == 8) rc = small ? skinny_launch<8, 2, 32>(p, grid, s) : skinny_launch<8, 4, 32>(p, grid, s);
    else rc = small ? skinny_launch<4, 2, 32>(p, grid, s) : skinny_launch<4, 4, 32>(p, grid, s);
    return rc == 0 ? S : rc;
}

// ================================================================================================================
// int8 weights (BLIP-2 `load_in_8bit`, reference captioner/models/blip2/blip2.py:19-22): the same weight-streaming GEMM with W
// stored as bitsandbytes stores a Linear8bitLt weight - one signed byte per element, q = rint(w * 127 / absmax(row)), and the
// row's absmax / 127 as an fp32 scale - so a decode step streams half the bytes of the bf16 form.  The activations stay bf16
// (bitsandbytes also quantises them to int8 per token, with columns beyond its threshold kept in fp16: not restated - this is
// the weight half of LLM.int8, "W8A16").  C = (A . q^T) * scale[n] (+ bias, act): the integers are exact in bf16, the sums fp32.
//
// Layout (written once by launch_quant_i8_pack): the bytes sit in MFMA FRAGMENT ORDER, so that a wave's global_load_dwordx4
// IS the operand fetch - no LDS staging, no swizzle, every instruction one contiguous KiB:
//   block (t, s) = weight rows 16 t .. 16 t + 15, k = 64 s .. 64 s + 63: 1 KiB at ((t * K / 64) + s) * 1024;
//   inside it lane l = r + 16 g (row r, k-group g) owns bytes 16 l .. 16 l + 15 = k-step 0 [8 bytes: k = 64 s + 8 g ..] then
//   k-step 1 [8 bytes: k = 64 s + 32 + 8 g ..].
// A row tile's blocks are consecutive in s: a wave walks contiguous memory.  The slice plan (S, waves, k per wave) is the bf16
// kernel's skinny_plan with 32-row workgroups, so sums depend on (N, K) and the output form only, never on the row count.
namespace {

typedef unsigned int sk_u32x4 __attribute__((ext_vector_type(4)));

struct SkinnyI8Params {
    const bf16_t* A; int lda;
    const unsigned char* Wp; const float* wscale; int nslab;      // nslab = K / 64
    const float* bias;
    bf16_t* out; int ldc;
    float* part;
    int M, N, S, nks, act;
    int G, units;
};

__device__ __forceinline__ bf16x8 i8x8_to_bf16(unsigned lo, unsigned hi) {
    bf16x8 w;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        w[e] = (bf16_t)(float)(int)(signed char)((lo >> (8 * e)) & 0xFFu);
        w[4 + e] = (bf16_t)(float)(int)(signed char)((hi >> (8 * e)) & 0xFFu);
    }
    return w;
}

// XT = 16-row tiles of activations per workgroup (1: up to 16 rows per group - the one-crop decode step; 2: 32); WT = 16-row tiles
// of WEIGHT rows per workgroup: 2, or 4 for 17-32 activation rows - every workgroup pulls the whole activation block (rows x K x 2
// bytes) through its L2 port, at 32 rows twice its 32 weight rows' bytes: 64 weight rows per workgroup halve that traffic (fc1 at
// 32 rows: 17.6 -> 12.9 us; the launcher says where).  An output element's MFMA sequence and wave-order sum are the same in every variant: same bits.
template <int NW, int XT, int WT>
__global__ __launch_bounds__(NW * 64) void gemm_skinny_i8_kernel(SkinnyI8Params p) {
    constexpr int TR = 16 * WT, PITCH = WT * 16 + 4, MR = XT * 16;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63, r16 = lane & 15, kg = lane >> 4;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int per = 8 * p.G, rr = blockIdx.x % per;              // (unit, row group): as in gemm_skinny_kernel
    const int unit = (blockIdx.x / per) * 8 + (rr & 7);
    if (unit >= p.units) return;
    const int nt = unit / p.S, z = unit % p.S;
    const int n0 = nt * TR, m0 = (rr >> 3) * MR;
    const int kw = (z * NW + wave) * p.nks * 32;
    const int nsl = p.nks >> 1;

    // the weight stream first: every block of this wave's K range in flight at once (one memory round trip)
    sk_u32x4 wq[WT][SK_MAXKS / 2];
#pragma unroll
    for (int t = 0; t < WT; ++t) {
        const sk_u32x4* wp = (const sk_u32x4*)(p.Wp + ((size_t)(nt * WT + t) * p.nslab + (kw >> 6)) * 1024) + lane;
#pragma unroll
        for (int sl = 0; sl < SK_MAXKS / 2; ++sl)
            if (sl < nsl) wq[t][sl] = __builtin_nontemporal_load(wp + sl * 64);
    }
    // activations (L2 hits): straight into the MFMA operand layout
    bf16x8 af[XT][SK_MAXKS];
#pragma unroll
    for (int t = 0; t < XT; ++t) {
        const bf16_t* ap = p.A + (size_t)min(m0 + t * 16 + r16, p.M - 1) * p.lda + kw + kg * 8;
#pragma unroll
        for (int j = 0; j < SK_MAXKS; ++j)
            if (j < p.nks) af[t][j] = *(const bf16x8*)(ap + j * 32);
    }
    f32x4 acc[WT][XT];
#pragma unroll
    for (int a = 0; a < WT; ++a)
#pragma unroll
        for (int b = 0; b < XT; ++b) acc[a][b] = f32x4(0.f);
#pragma unroll
    for (int sl = 0; sl < SK_MAXKS / 2; ++sl)
        if (sl < nsl) {
#pragma unroll
            for (int wt = 0; wt < WT; ++wt) {
                const bf16x8 w0 = i8x8_to_bf16(wq[wt][sl].x, wq[wt][sl].y), w1 = i8x8_to_bf16(wq[wt][sl].z, wq[wt][sl].w);
#pragma unroll
                for (int xt = 0; xt < XT; ++xt) {
                    acc[wt][xt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w0, af[xt][2 * sl], acc[wt][xt], 0, 0, 0);
                    acc[wt][xt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w1, af[xt][2 * sl + 1], acc[wt][xt], 0, 0, 0);
                }
            }
        }
    // the waves' tiles summed through LDS in wave order, then scale (+ bias, act) or the slice sum
    float* red = (float*)smem;
#pragma unroll
    for (int wt = 0; wt < WT; ++wt)
#pragma unroll
        for (int xt = 0; xt < XT; ++xt)
            *(f32x4*)(red + (wave * MR + xt * 16 + r16) * PITCH + wt * 16 + 4 * kg) = acc[wt][xt];
    __syncthreads();
    for (int o = tid; o < MR * (TR / 2); o += NW * 64) {
        const int m = o / (TR / 2), n = (o % (TR / 2)) * 2;
        f32x2 v = *(const f32x2*)(red + m * PITCH + n);
#pragma unroll
        for (int w = 1; w < NW; ++w) v += *(const f32x2*)(red + (w * MR + m) * PITCH + n);
        if (m0 + m >= p.M) continue;
        v *= *(const f32x2*)(p.wscale + n0 + n);
        if (p.part) {
            *(f32x2*)(p.part + ((size_t)z * p.M + m0 + m) * p.N + n0 + n) = v;
            continue;
        }
        if (p.bias) v += *(const f32x2*)(p.bias + n0 + n);
        if (p.act == 2) { v[0] = fmaxf(v[0], 0.f); v[1] = fmaxf(v[1], 0.f); }
        else if (p.act == 1) { v[0] = gelu_erf_fast(v[0]); v[1] = gelu_erf_fast(v[1]); }
        bf16x2 ob;
        ob[0] = (bf16_t)v[0]; ob[1] = (bf16_t)v[1];
        *(bf16x2*)(p.out + (size_t)(m0 + m) * p.ldc + n0 + n) = ob;
    }
}

// More than 32 activation rows (the prompt pass of a batch: B x 33 rows): one workgroup per unit walks ALL row groups with its
// weight fragments converted once and kept in registers (separate workgroups per row group, the bf16 kernel's way, repeat the
// byte -> bf16 conversion of the unit's weights for every group).  Per row the MFMA sequence and the wave-order sum are
// gemm_skinny_i8_kernel's: the same bits whichever kernel the row count picks.
// Measured (tools/bench_skinny_i8.py: graph replay over rotating weight copies, us per launch, int8 / bf16 kernel):
//   rows      1            16           32           64           1056
//   qkv     6.5 / 10.3   8.0 / 11.2   11.2 / 13.3  18.6 / 20.1  184 / 215
//   out     4.3 /  5.6   5.4 /  6.2    7.2 /  7.1  13.4 /  9.6  111 /  64
//   fc1     8.9 / 12.9  11.5 / 13.9   12.9 / 16.1  34.3 / 27.9  365 / 286      (32 rows: 17.6 / 17.5 with 32 weight rows per workgroup)
//   fc2     8.5 / 13.0  11.4 / 13.7   12.7 / 15.5  33.8 / 26.9  357 / 280
// The mode is built for the reference's call pattern (one crop per call, a few at most): up to 16 rows the byte stream is 20-35 %
// shorter per launch; from ~32 rows on a launch is paced by the ACTIVATION fragments every workgroup pulls from L2 (32 rows x K
// x 2 bytes per 32 weight rows: twice the weight bytes at 32 rows), which the bytes saved on W do not touch.  A batch's prompt
// pass (1 056 rows at 32 crops) is a GEMM proper: beyond 4 crops per call run_opt unpacks each matrix into a bf16 scratch of the
// exact integers (dequant_i8_rowmajor_kernel) for the tiled kernel and applies the row scales to its fp32 output
// (scale_cols_kernel): ms per generate, int8 / bf16, by crops per call: 1: 44.0 / 52.7, 4: 49.9 / 55.4, 8: 55.4 / 59.6 (58.3 with this
// kernel for the prompt), 16: 66.9 / 67.8, 32: 83.9 / 85.5 (113; 91.3 before the 64-weight-row form of the 17-32 row steps).
template <int NW>
__global__ __launch_bounds__(NW * 64) void gemm_skinny_i8_rows_kernel(SkinnyI8Params p) {
    constexpr int TR = 32, WT = 2, XT = 2, PITCH = WT * 16 + 4, MR = XT * 16;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63, r16 = lane & 15, kg = lane >> 4;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int unit = blockIdx.x;
    if (unit >= p.units) return;
    const int nt = unit / p.S, z = unit % p.S;
    const int n0 = nt * TR;
    const int kw = (z * NW + wave) * p.nks * 32;
    const int nsl = p.nks >> 1;
    bf16x8 wf[WT][SK_MAXKS / 2][2];
#pragma unroll
    for (int t = 0; t < WT; ++t) {
        const sk_u32x4* wp = (const sk_u32x4*)(p.Wp + ((size_t)(nt * WT + t) * p.nslab + (kw >> 6)) * 1024) + lane;
#pragma unroll
        for (int sl = 0; sl < SK_MAXKS / 2; ++sl)
            if (sl < nsl) {
                const sk_u32x4 q = __builtin_nontemporal_load(wp + sl * 64);
                wf[t][sl][0] = i8x8_to_bf16(q.x, q.y);
                wf[t][sl][1] = i8x8_to_bf16(q.z, q.w);
            }
    }
    float* red = (float*)smem;
    const f32x2 sc = *(const f32x2*)(p.wscale + n0 + (tid % (TR / 2)) * 2);
    for (int g = 0; g < p.G; ++g) {
        const int m0 = g * MR;
        bf16x8 af[XT][SK_MAXKS];
#pragma unroll
        for (int t = 0; t < XT; ++t) {
            const bf16_t* ap = p.A + (size_t)min(m0 + t * 16 + r16, p.M - 1) * p.lda + kw + kg * 8;
#pragma unroll
            for (int j = 0; j < SK_MAXKS; ++j)
                if (j < p.nks) af[t][j] = *(const bf16x8*)(ap + j * 32);
        }
        f32x4 acc[WT][XT];
#pragma unroll
        for (int a = 0; a < WT; ++a)
#pragma unroll
            for (int b = 0; b < XT; ++b) acc[a][b] = f32x4(0.f);
#pragma unroll
        for (int sl = 0; sl < SK_MAXKS / 2; ++sl)
            if (sl < nsl) {
#pragma unroll
                for (int wt = 0; wt < WT; ++wt)
#pragma unroll
                    for (int xt = 0; xt < XT; ++xt) {
                        acc[wt][xt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[wt][sl][0], af[xt][2 * sl], acc[wt][xt], 0, 0, 0);
                        acc[wt][xt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[wt][sl][1], af[xt][2 * sl + 1], acc[wt][xt], 0, 0, 0);
                    }
            }
#pragma unroll
        for (int wt = 0; wt < WT; ++wt)
#pragma unroll
            for (int xt = 0; xt < XT; ++xt)
                *(f32x4*)(red + (wave * MR + xt * 16 + r16) * PITCH + wt * 16 + 4 * kg) = acc[wt][xt];
        __syncthreads();
        for (int o = tid; o < MR * (TR / 2); o += NW * 64) {                 // (NW * 64 is a multiple of TR / 2: a thread's columns are fixed)
            const int m = o / (TR / 2), n = (o % (TR / 2)) * 2;
            f32x2 v = *(const f32x2*)(red + m * PITCH + n);
#pragma unroll
            for (int w = 1; w < NW; ++w) v += *(const f32x2*)(red + (w * MR + m) * PITCH + n);
            if (m0 + m >= p.M) continue;
            v *= sc;
            if (p.part) {
                *(f32x2*)(p.part + ((size_t)z * p.M + m0 + m) * p.N + n0 + n) = v;
                continue;
            }
            if (p.bias) v += *(const f32x2*)(p.bias + n0 + n);
            if (p.act == 2) { v[0] = fmaxf(v[0], 0.f); v[1] = fmaxf(v[1], 0.f); }
            else if (p.act == 1) { v[0] = gelu_erf_fast(v[0]); v[1] = gelu_erf_fast(v[1]); }
            bf16x2 ob;
            ob[0] = (bf16_t)v[0]; ob[1] = (bf16_t)v[1];
            *(bf16x2*)(p.out + (size_t)(m0 + m) * p.ldc + n0 + n) = ob;
        }
        __syncthreads();                                                     // the next group's tiles reuse the buffer
    }
}

// fp32 [rows, cols] -> row scales + the fragment-ordered bytes.  One workgroup per 16-row tile: row maxima first (16 threads per
// row), then every thread writes whole 16-byte lane pieces.  q = rint(w * (127 / absmax)) in fp32 (IEEE divide, round-half-even):
// the arithmetic of oracle/blip2_ref.py::quantize_int8_rowwise, bit for bit.
__global__ __launch_bounds__(256) void quant_i8_pack_kernel(const float* __restrict__ src, unsigned char* __restrict__ dst, float* __restrict__ scale,
                                                            int cols) {
    __shared__ float rmax[16][17];
    __shared__ float rinv[16];
    const int t = blockIdx.x, tid = threadIdx.x, r = tid >> 4, c16 = tid & 15;
    const float* row = src + (size_t)(t * 16 + r) * cols;
    float am = 0.f;
    for (int c = c16; c < cols; c += 16) am = fmaxf(am, fabsf(row[c]));
    rmax[r][c16] = am;
    __syncthreads();
    if (tid < 16) {
        float a = 0.f;
        for (int j = 0; j < 16; ++j) a = fmaxf(a, rmax[tid][j]);
        scale[t * 16 + tid] = a / 127.0f;
        rinv[tid] = a > 0.f ? 127.0f / a : 0.f;
    }
    __syncthreads();
    const int nslab = cols >> 6;
    unsigned char* blk = dst + (size_t)t * nslab * 1024;
    for (int o = tid; o < nslab * 64; o += 256) {
        const int sl = o >> 6, l = o & 63, rr = l & 15, g = l >> 4;
        const float* sp = src + (size_t)(t * 16 + rr) * cols + sl * 64 + g * 8;
        const float inv = rinv[rr];
        unsigned w[4];
#pragma unroll
        for (int h = 0; h < 4; ++h) {                     // words 0, 1: k-step 0; words 2, 3: k-step 1
            const float* q = sp + (h >> 1) * 32 + (h & 1) * 4;
            unsigned u = 0;
#pragma unroll
            for (int e = 0; e < 4; ++e) u |= ((unsigned)(int)__builtin_rintf(q[e] * inv) & 0xFFu) << (8 * e);
            w[h] = u;
        }
        *(sk_u32x4*)(blk + (size_t)o * 16) = sk_u32x4{w[0], w[1], w[2], w[3]};
    }
}

// The fragment-ordered bytes back as a row-major bf16 matrix of the INTEGERS (exact in bf16), for the tiled GEMM: a batch's prompt
// pass is a GEMM proper (B x 33 rows), and every workgroup of the weight-streaming kernels pulls all of A through its own L2 port.
__global__ __launch_bounds__(256) void dequant_i8_rowmajor_kernel(const unsigned char* __restrict__ src, bf16_t* __restrict__ dst, int cols, size_t nblk) {
    const int nslab = cols >> 6, lane = threadIdx.x & 63;
    for (size_t blk = (size_t)blockIdx.x * 4 + (threadIdx.x >> 6); blk < nblk; blk += (size_t)gridDim.x * 4) {
        const size_t t = blk / nslab;
        const int sl = (int)(blk - t * nslab), r = lane & 15, g = lane >> 4;
        const sk_u32x4 q = *((const sk_u32x4*)(src + blk * 1024) + lane);
        bf16_t* o = dst + (t * 16 + r) * (size_t)cols + sl * 64 + g * 8;
        *(bf16x8*)o = i8x8_to_bf16(q.x, q.y);
        *(bf16x8*)(o + 32) = i8x8_to_bf16(q.z, q.w);
    }
}

// part[m][n] *= scale[n]: the row scales of the int8 weights on the tiled GEMM's fp32 output (its epilogue has no such operand)
__global__ __launch_bounds__(256) void scale_cols_kernel(float* __restrict__ part, const float* __restrict__ scale, size_t n4, int N) {
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (size_t)gridDim.x * 256) {
        const int c = (int)((i * 4) % N);
        f32x4 v = *((f32x4*)part + i);
        v *= *(const f32x4*)(scale + c);
        *((f32x4*)part + i) = v;
    }
}

template <int NW, int XT, int WT = 2>
int skinny_i8_launch(const SkinnyI8Params& p, dim3 grid, hipStream_t s) {
    constexpr int lds = NW * XT * 16 * (WT * 16 + 4) * 4;
    hipLaunchKernelGGL((gemm_skinny_i8_kernel<NW, XT, WT>), grid, dim3(NW * 64), lds, s, p);
    CAP_HIP_CHECK(hipGetLastError());
    return 0;
}

}  // namespace

// the int8 form's plan: skinny_plan with 32-row workgroups only (N % 32 == 0)
int skinny_i8_plan(int N, int K, bool finished, int* nw_out) {
    if (N % 32 != 0) return 0;
    int nw = 0, tr = 0;
    const int S = skinny_plan(N, K, finished, &nw, &tr, 1 << 30);      // (a row count that never asks for 40-row workgroups)
    if (S < 1 || tr != 32) return 0;
    if (nw_out) *nw_out = nw;
    return S;
}

int launch_quant_i8_pack(const float* src, void* dst, float* scale, int rows, int cols, hipStream_t s) {
    if (rows % 16 != 0 || cols % 64 != 0) { cap_set_error("quant_i8_pack: rows %% 16 / cols %% 64 (%d x %d)", rows, cols); return -1; }
    hipLaunchKernelGGL(quant_i8_pack_kernel, dim3(rows / 16), dim3(256), 0, s, src, (unsigned char*)dst, scale, cols);
    CAP_HIP_CHECK(hipGetLastError());
    return 0;
}

int launch_dequant_i8_rowmajor(const void* packed, void* dst_bf16, int rows, int cols, hipStream_t s) {
    if (rows % 16 != 0 || cols % 64 != 0) { cap_set_error("dequant_i8: rows %% 16 / cols %% 64 (%d x %d)", rows, cols); return -1; }
    const size_t nblk = (size_t)(rows / 16) * (cols / 64);
    hipLaunchKernelGGL(dequant_i8_rowmajor_kernel, dim3((unsigned)std::min<size_t>((nblk + 3) / 4, 4096)), dim3(256), 0, s, (const unsigned char*)packed,
                       (bf16_t*)dst_bf16, cols, nblk);
    CAP_HIP_CHECK(hipGetLastError());
    return 0;
}

int launch_scale_cols(float* part, const float* scale, int M, int N, hipStream_t s) {
    if (N % 4 != 0) { cap_set_error("scale_cols: N %% 4"); return -1; }
    const size_t n4 = (size_t)M * N / 4;
    hipLaunchKernelGGL(scale_cols_kernel, dim3((unsigned)std::min<size_t>((n4 + 255) / 256, 4096)), dim3(256), 0, s, part, scale, n4, N);
    CAP_HIP_CHECK(hipGetLastError());
    return 0;
}

int launch_gemm_skinny_i8(const void* A, int lda, const void* Wp, const float* wscale, const float* bias, int act, void* out, int ldc,
                          float* part, int M, int N, int K, hipStream_t s) {
    int nw = 0;
    const int S = skinny_i8_plan(N, K, part == nullptr, &nw);
    if (S < 1 || M < 1 || (lda & 7) || (!part && (!out || (ldc & 1)))) {
        cap_set_error("gemm_skinny_i8: unsupported shape M=%d N=%d K=%d", M, N, K);
        return -1;
    }
    SkinnyI8Params p;
    p.A = (const bf16_t*)A; p.lda = lda; p.Wp = (const unsigned char*)Wp; p.wscale = wscale; p.nslab = K / 64; p.bias = bias;
    p.out = (bf16_t*)out; p.ldc = ldc; p.part = part; p.M = M; p.N = N; p.S = S; p.nks = K / (nw * 32 * S); p.act = act;
    const bool one = M <= 16;                             // one 16-row tile of activations per workgroup
    p.G = one ? 1 : (M + 31) / 32; p.units = (N / 32) * S;
    int rc = 0;
    if (p.G > 1) {                                        // several row groups: one workgroup per unit walks them all
        constexpr int lds8 = 8 * 32 * 36 * 4, lds4 = 4 * 32 * 36 * 4;
        if (nw == 8) hipLaunchKernelGGL(gemm_skinny_i8_rows_kernel<8>, dim3(p.units), dim3(512), lds8, s, p);
        else hipLaunchKernelGGL(gemm_skinny_i8_rows_kernel<4>, dim3(p.units), dim3(256), lds4, s, p);
        CAP_HIP_CHECK(hipGetLastError());
        return S;
    }
    // 17-32 rows: 64 weight rows per workgroup (half the activation traffic) where that still leaves a workgroup for most CUs
    // (tools/bench_skinny_i8.py, 32 rows, us per launch, 32 / 64 weight rows: fc1 17.6 / 12.9, fc2 17.5 / 12.7 - 160 workgroups;
    // qkv 11.2 / 12.4 - 120, out_proj 7.2 / 8.5 - 80: those keep 32)
    if (!one && N % 64 == 0 && (N / 64) * S >= 160) {
        p.units = (N / 64) * S;
        const dim3 grid4(((p.units + 7) / 8) * 8);
        rc = nw == 8 ? skinny_i8_launch<8, 2, 4>(p, grid4, s) : skinny_i8_launch<4, 2, 4>(p, grid4, s);
        return rc == 0 ? S : rc;
    }
    const dim3 grid(((p.units + 7) / 8) * 8 * p.G);
    if (nw == 8) rc = one ? skinny_i8_launch<8, 1>(p, grid, s) : skinny_i8_launch<8, 2>(p, grid, s);
    else rc = one ? skinny_i8_launch<4, 1>(p, grid, s) : skinny_i8_launch<4, 2>(p, grid, s);
    return rc == 0 ? S : rc;
}
