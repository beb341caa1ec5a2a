// MFMA GEMM for the captioner hot path (gfx950 / CDNA4).
//
//   C[M,N] = A[M,K] . W[N,K]^T  (+bias, GELU, residual, layout-remapping epilogues)
//
// Both operands are K-contiguous (activations row-major, torch Linear weights [out,in]), so A and W tiles
// have the same shape in LDS: rows of one 128-byte K-slab (64 bf16 / 32 fp32), 16-byte chunks XOR-swizzled
// by ((row>>1)&7) so that the ds_read_b128 fragment reads of a 32-row MFMA operand are bank-conflict free
// (MI355X LDS: ds_read_b128 is served in 16-lane groups over 64 banks).
//   bf16:  v_mfma_f32_32x32x16_bf16, one 16-byte chunk per lane per k-step (lane half h takes chunk 2*ks+h).
//   fp32:  v_mfma_f32_32x32x2_f32 (exact fp32 FMA chain, 1/16 the bf16 rate); a 16-byte chunk holds 4 k values,
//          lane half h takes chunk 2*ks+h and feeds 4 MFMAs - A and W use the same k permutation so the sum is
//          unchanged.
// Global->LDS goes through registers (next slab prefetched into VGPRs while the current one is multiplied), LDS is
// double buffered: one barrier per K-slab.  Edge tiles clamp their load rows and guard their stores.
// Workgroup ids are remapped so each XCD (blockIdx % 8) walks a contiguous run of tiles that share A panels in its L2.
#include "gemm.h"

namespace {

template <typename T> struct Mma;
template <> struct Mma<bf16_t> {
    using vec = bf16x8;
    static constexpr int EPC = 8;          // elements per 16-byte chunk
    __device__ static __forceinline__ void run(f32x16& acc, const vec& a, const vec& b) {
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc, 0, 0, 0);
    }
};
template <> struct Mma<float> {
    using vec = f32x4;
    static constexpr int EPC = 4;
    __device__ static __forceinline__ void run(f32x16& acc, const vec& a, const vec& b) {
#pragma unroll
        for (int j = 0; j < 4; ++j) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[j], b[j], acc, 0, 0, 0);
    }
};

__device__ __forceinline__ int swz_off(int row, int chunk) { return row * 128 + ((chunk ^ ((row >> 1) & 7)) << 4); }

template <typename T, bool OUT_F32, int EPI>
__device__ __forceinline__ void epi_store(const GemmParams& p, int row, int col, float v) {
    if (p.bias) v += p.bias[col];
    if (p.gelu) v = gelu_erf(v);
    if constexpr (EPI == EPI_STORE) {
        size_t o = (size_t)row * p.ldc + col;
        if (p.resid) v += p.resid[(size_t)row * p.ldr + col];
        if constexpr (OUT_F32) ((float*)p.C)[o] = v; else ((T*)p.C)[o] = from_f32<T>(v);
    } else if constexpr (EPI == EPI_PATCH) {
        int b = row / p.p0, pp = row - b * p.p0;
        v += p.aux[(size_t)(1 + pp) * p.N + col];
        size_t o = ((size_t)b * (p.p0 + 1) + 1 + pp) * p.ldc + col;
        ((float*)p.C)[o] = v;
    } else if constexpr (EPI == EPI_CROSSKV) {
        int NT = p.p0, H = p.p1, B = p.p2, Dh = H * 64;
        int b = row / NT, t = row - b * NT;
        int l = col / (2 * Dh), r = col - l * 2 * Dh, kv = r / Dh, hd = r - kv * Dh, h = hd >> 6, d = hd & 63;
        size_t o = (((((size_t)l * 2 + kv) * B + b) * H + h) * NT + t) * 64 + d;
        ((T*)p.C)[o] = from_f32<T>(v);
    } else {  // EPI_QKVCACHE
        int H = p.p1, Dh = H * 64;
        if (col < Dh) {
            ((T*)p.C)[(size_t)row * Dh + col] = from_f32<T>(v);
        } else {
            int kv = col / Dh - 1, hd = col % Dh, h = hd >> 6, d = hd & 63;
            size_t o = ((((size_t)kv * p.p0 + row) * H + h) * p.p2 + p.p3) * 64 + d;
            ((T*)p.C2)[o] = from_f32<T>(v);
        }
    }
}

template <typename T, int BM, int BN, int WM, int WN, bool OUT_F32, int EPI>
__global__ __launch_bounds__((BM / WM) * (BN / WN) * 64) void gemm_kernel(GemmParams p) {
    constexpr int NWM = BM / WM, NWN = BN / WN, NT = NWM * NWN * 64;
    constexpr int MI = WM / 32, NI = WN / 32;
    constexpr int LA = BM * 8 / NT, LB = BN * 8 / NT;      // 16-byte chunks per thread per slab
    static_assert(LA >= 1 && LB >= 1, "tile too small for the thread count");
    constexpr int EPC = Mma<T>::EPC;
    constexpr int SLAB = 8 * EPC;                          // K elements per 128-byte slab
    using vec = typename Mma<T>::vec;

    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int BUF = (BM + BN) * 128;                   // bytes per LDS buffer: A tile then W tile

    // XCD-aware, bijective remap of the linear block id (guide T1): blocks b, b+8, ... share an XCD.
    const int ntm = (p.M + BM - 1) / BM, ntn = (p.N + BN - 1) / BN, nwg = ntm * ntn;
    int bid = blockIdx.x;
    {
        int q = nwg >> 3, r = nwg & 7, xcd = bid & 7, idx = bid >> 3;
        bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
    }
    const int tm = bid / ntn, tn = bid - tm * ntn;
    const int m0 = tm * BM, n0 = tn * BN;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm0 = (wave / NWN) * WM, wn0 = (wave % NWN) * WN;
    const int r32 = lane & 31, h = lane >> 5;

    const T* A = (const T*)p.A;
    const T* W = (const T*)p.W;

    // per-thread global source pointers and LDS destinations for the staging copies
    const uint4* ga[LA]; const uint4* gb[LB]; int da[LA], db[LB];
#pragma unroll
    for (int i = 0; i < LA; ++i) {
        int c = tid + i * NT, row = c >> 3, ch = c & 7;
        int gr = min(m0 + row, p.M - 1);
        ga[i] = (const uint4*)(A + (size_t)gr * p.lda + ch * EPC);
        da[i] = swz_off(row, ch);
    }
#pragma unroll
    for (int i = 0; i < LB; ++i) {
        int c = tid + i * NT, row = c >> 3, ch = c & 7;
        int gr = min(n0 + row, p.N - 1);
        gb[i] = (const uint4*)(W + (size_t)gr * p.ldw + ch * EPC);
        db[i] = swz_off(row, ch);
    }

    f32x16 acc[MI][NI];
#pragma unroll
    for (int i = 0; i < MI; ++i)
#pragma unroll
        for (int j = 0; j < NI; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

    uint4 ra[LA], rb[LB];
    const int nk = p.K / SLAB;
#pragma unroll
    for (int i = 0; i < LA; ++i) ra[i] = ga[i][0];
#pragma unroll
    for (int i = 0; i < LB; ++i) rb[i] = gb[i][0];
#pragma unroll
    for (int i = 0; i < LA; ++i) *(uint4*)(smem + da[i]) = ra[i];
#pragma unroll
    for (int i = 0; i < LB; ++i) *(uint4*)(smem + BM * 128 + db[i]) = rb[i];
    __syncthreads();

    for (int kt = 0; kt < nk; ++kt) {
        const int cur = kt & 1;
        if (kt + 1 < nk) {   // prefetch the next slab into registers; it lands while the MFMAs run
            const int ko = (kt + 1) * (SLAB * (int)sizeof(T) / 16);
#pragma unroll
            for (int i = 0; i < LA; ++i) ra[i] = ga[i][ko];
#pragma unroll
            for (int i = 0; i < LB; ++i) rb[i] = gb[i][ko];
        }
        const char* a_s = smem + cur * BUF;
        const char* b_s = a_s + BM * 128;
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
            vec af[MI], bf[NI];
#pragma unroll
            for (int i = 0; i < MI; ++i) af[i] = *(const vec*)(a_s + swz_off(wm0 + i * 32 + r32, ks * 2 + h));
#pragma unroll
            for (int j = 0; j < NI; ++j) bf[j] = *(const vec*)(b_s + swz_off(wn0 + j * 32 + r32, ks * 2 + h));
#pragma unroll
            for (int i = 0; i < MI; ++i)
#pragma unroll
                for (int j = 0; j < NI; ++j) Mma<T>::run(acc[i][j], af[i], bf[j]);
        }
        if (kt + 1 < nk) {
#pragma unroll
            for (int i = 0; i < LA; ++i) *(uint4*)(smem + (cur ^ 1) * BUF + da[i]) = ra[i];
#pragma unroll
            for (int i = 0; i < LB; ++i) *(uint4*)(smem + (cur ^ 1) * BUF + BM * 128 + db[i]) = rb[i];
        }
        __syncthreads();
    }

    // epilogue: acc[i][j][e] is C[row = (e&3) + 8*(e>>2) + 4*h][col = lane&31] of its 32x32 tile
#pragma unroll
    for (int i = 0; i < MI; ++i)
#pragma unroll
        for (int j = 0; j < NI; ++j) {
            const int col = n0 + wn0 + j * 32 + r32;
            if (col >= p.N) continue;
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const int row = m0 + wm0 + i * 32 + (e & 3) + 8 * (e >> 2) + 4 * h;
                if (row < p.M) epi_store<T, OUT_F32, EPI>(p, row, col, acc[i][j][e]);
            }
        }
}

template <typename T, int BM, int BN, int WM, int WN, bool OUT_F32, int EPI>
int launch_cfg(const GemmParams& p, hipStream_t stream) {
    constexpr int NT = (BM / WM) * (BN / WN) * 64;
    constexpr int LDS = 2 * (BM + BN) * 128;
    const int ntm = (p.M + BM - 1) / BM, ntn = (p.N + BN - 1) / BN;
    auto kern = gemm_kernel<T, BM, BN, WM, WN, OUT_F32, EPI>;
    hipLaunchKernelGGL(kern, dim3(ntm * ntn), dim3(NT), LDS, stream, p);
    CAP_HIP_CHECK(hipGetLastError());
    return 0;
}

template <typename T, bool OUT_F32, int EPI>
int launch_tile(const GemmParams& p, int tile, hipStream_t stream) {
    if (tile == 1) return launch_cfg<T, 128, 128, 64, 64, OUT_F32, EPI>(p, stream);
    return launch_cfg<T, 64, 64, 32, 32, OUT_F32, EPI>(p, stream);
}

template <typename T>
int launch_t(const GemmParams& p, int tile, hipStream_t stream) {
    switch (p.epi) {
        case EPI_STORE:
            return p.out_f32 ? launch_tile<T, true, EPI_STORE>(p, tile, stream)
                             : launch_tile<T, false, EPI_STORE>(p, tile, stream);
        case EPI_PATCH: return launch_tile<T, true, EPI_PATCH>(p, tile, stream);
        case EPI_CROSSKV: return launch_tile<T, false, EPI_CROSSKV>(p, tile, stream);
        case EPI_QKVCACHE: return launch_tile<T, false, EPI_QKVCACHE>(p, tile, stream);
    }
    cap_set_error("launch_gemm: unknown epilogue %d", p.epi);
    return -1;
}

}  // namespace

int launch_gemm(int dtype, const GemmParams& p, int tile, hipStream_t stream) {
    const int slab = dtype == CAP_DT_BF16 ? 64 : 32;
    const int esz = dtype == CAP_DT_BF16 ? 2 : 4;
    if (p.M <= 0 || p.N <= 0 || p.K <= 0 || p.K % slab != 0) {
        cap_set_error("launch_gemm: bad shape M=%d N=%d K=%d (K must be a multiple of %d)", p.M, p.N, p.K, slab);
        return -1;
    }
    if ((p.lda * esz) % 16 != 0 || (p.ldw * esz) % 16 != 0 || ((uintptr_t)p.A & 15) || ((uintptr_t)p.W & 15)) {
        cap_set_error("launch_gemm: operands must be 16-byte aligned (lda=%d ldw=%d)", p.lda, p.ldw);
        return -1;
    }
    if (tile == 0) {
        // big tile once it still fills the chip (>= 2 tiles per CU), small tile for the decode-sized GEMMs
        long big = (long)((p.M + 127) / 128) * ((p.N + 127) / 128);
        tile = big >= 512 ? 1 : 2;
    }
    if (dtype == CAP_DT_BF16) return launch_t<bf16_t>(p, tile, stream);
    if (dtype == CAP_DT_F32) return launch_t<float>(p, tile, stream);
    cap_set_error("launch_gemm: unknown dtype %d", dtype);
    return -1;
}
