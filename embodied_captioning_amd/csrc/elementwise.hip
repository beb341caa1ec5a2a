// HBM-bound helper kernels of the captioner path: dtype casts, patch gather, LayerNorm, decoder embeddings,
// greedy token selection.  One wave (64 lanes) per row for the row-wise ops; 16-byte accesses where the layout allows.
#include <algorithm>
#include "ops.h"

namespace {

// ------------------------------------------------------------------------------------------------
template <typename T>
__global__ void convert_kernel(const float* __restrict__ src, T* __restrict__ dst, size_t n) {
    size_t i = ((size_t)blockIdx.x * blockDim.x + threadIdx.x) * 4;
    const size_t stride = (size_t)gridDim.x * blockDim.x * 4;
    for (; i + 3 < n; i += stride) store4(dst + (i & ~(size_t)7), (int)(i & 7), *(const float4*)(src + i));
    if (blockIdx.x == 0 && threadIdx.x == 0)
        for (size_t j = n & ~(size_t)3; j < n; ++j) store1(dst + (j & ~(size_t)7), (int)(j & 7), src[j]);
}

// scale: 1 except for G8 weights (G8_WSCALE, see common.h)
template <typename T>
__global__ void convert2d_kernel(const float* __restrict__ src, T* __restrict__ dst, int rows, int cols, int dst_ld, float scale) {
    const size_t n = (size_t)rows * cols;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        const size_t r = i / cols, c = i - r * cols;
        store1(dst + r * dst_ld, (int)c, src[i] * scale);
    }
}

template <typename T>
__global__ void convert2d_t_kernel(const float* __restrict__ src, T* __restrict__ dst, int rows, int cols, float scale) {
    // dst[c][r] = src[r][c]  (one-off weight transposition, e.g. CoCa's text_projection [width, vocab])
    const size_t n = (size_t)rows * cols;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        const size_t c = i / rows, r = i - c * rows;
        store1(dst + c * rows, (int)r, src[r * cols + c] * scale);
    }
}

// ------------------------------------------------------------------------------------------------
// One thread per output element; consecutive threads walk x inside an image row, so the pixel reads are
// coalesced (fp32 NCHW: 4 B/lane contiguous; u8 NHWC: 3 B stride) and each thread writes one element of the
// patch matrix.  k = c*ps*ps + dy*ps + dx  (== Conv2d weight [D,3,ps,ps] flattened).
template <typename T>
__global__ void patchify_kernel(const void* __restrict__ pixels, int fmt, int B, int img, int ps, int Kpad,
                                T* __restrict__ out, float m0, float m1, float m2, float s0, float s1, float s2) {
    const int G = img / ps;
    const size_t total = (size_t)B * 3 * img * img;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        int x = (int)(i % img);
        size_t r = i / img;
        int y = (int)(r % img); r /= img;
        int c = (int)(r % 3);
        int b = (int)(r / 3);
        float v;
        if (fmt == 0) {
            v = ((const float*)pixels)[i];
        } else {
            unsigned char u = ((const unsigned char*)pixels)[(((size_t)b * img + y) * img + x) * 3 + c];
            float mean = c == 0 ? m0 : (c == 1 ? m1 : m2), sd = c == 0 ? s0 : (c == 1 ? s1 : s2);
            v = ((float)u * (1.0f / 255.0f) - mean) / sd;
        }
        int py = y / ps, dy = y - py * ps, px = x / ps, dx = x - px * ps;
        size_t row = (size_t)b * G * G + (size_t)py * G + px;
        store1(out + row * Kpad, c * ps * ps + dy * ps + dx, v);
    }
}

__global__ void cls_rows_kernel(const float* __restrict__ cls, const float* __restrict__ pos, float* __restrict__ X,
                                int B, int tokens, int D) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= B * D) return;
    int b = i / D, d = i - b * D;
    X[(size_t)b * tokens * D + d] = cls[d] + pos[d];
}

#include "ln.h"   // LN_MAXV, ln_row

template <typename T, int MAXV>
__global__ __launch_bounds__(256) void layernorm_kernel(const float* __restrict__ in, int ld_in,
                                                        const float* __restrict__ gamma, const float* __restrict__ beta,
                                                        float eps, T* out_t, float* out_f, int M, int D) {
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (row >= M) return;
    const int nv = (D + 255) / 256;
    float4 v[MAXV];
    const float* x = in + (size_t)row * ld_in;
#pragma unroll
    for (int i = 0; i < MAXV; ++i)
        if (i < nv && lane * 4 + i * 256 < D) v[i] = *(const float4*)(x + lane * 4 + i * 256);
    ln_row<T, MAXV>(v, nv, lane, D, gamma, beta, eps, out_t ? out_t + (size_t)row * D : nullptr,
              out_f ? out_f + (size_t)row * D : nullptr);
}

// Consumer of a split-K GEMM (EPI_PARTIAL): y = sum_z partial[z] + bias + resid, summed in slice order (deterministic),
// then LayerNorm -> out_f (fp32 residual stream) and out_t (next GEMM's A operand).
template <typename P> __device__ __forceinline__ float4 load4f(const P* p);
template <> __device__ __forceinline__ float4 load4f<float>(const float* p) { return *(const float4*)p; }
template <> __device__ __forceinline__ float4 load4f<bf16_t>(const bf16_t* p) {
    const bf16x4 r = *(const bf16x4*)p;
    return make_float4((float)r[0], (float)r[1], (float)r[2], (float)r[3]);
}

// P = element type of the partial sums: fp32 (split-K slices) or the compute type (the ViT branch outputs in `delta`).
template <typename T, typename P, int MAXV>
__global__ __launch_bounds__(256) void reduce_layernorm_kernel(const P* __restrict__ part, int S,
                                                                const float* __restrict__ bias,
                                                                const float* __restrict__ resid,
                                                                const float* __restrict__ gamma,
                                                                const float* __restrict__ beta, float eps, T* out_t,
                                                                float* out_f, float* y_out, int M, int D,
                                                                const int* __restrict__ n_rows) {
#pragma clang fp contract(off)      // sums and LayerNorm as written: the three consumers of a row count range agree bit for bit
    // one wave per row.  Decode (a few hundred rows) launches one wave per block so every row gets its own CU slot and
    // all of its S*nv + 2*nv 16-byte loads are issued before the first add; the encoder launches 4 rows per block.
    const int row = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (row >= M || (n_rows && row >= *n_rows)) return;
    const int nv = (D + 255) / 256;
    float4 v[MAXV];
#pragma unroll
    for (int i = 0; i < MAXV; ++i) {
        const int c = lane * 4 + i * 256;
        if (i < nv && c < D) {
            float4 a = load4f<P>(part + (size_t)row * D + c);
            if (S == 4) {
                const float4 b1 = load4f<P>(part + ((size_t)1 * M + row) * D + c);
                const float4 b2 = load4f<P>(part + ((size_t)2 * M + row) * D + c);
                const float4 b3 = load4f<P>(part + ((size_t)3 * M + row) * D + c);
                a.x = ((a.x + b1.x) + b2.x) + b3.x; a.y = ((a.y + b1.y) + b2.y) + b3.y;
                a.z = ((a.z + b1.z) + b2.z) + b3.z; a.w = ((a.w + b1.w) + b2.w) + b3.w;
            } else {
#pragma unroll 4
                for (int z = 1; z < S; ++z) {
                    const float4 b = load4f<P>(part + ((size_t)z * M + row) * D + c);
                    a.x += b.x; a.y += b.y; a.z += b.z; a.w += b.w;
                }
            }
            if (bias) { const float4 b = *(const float4*)(bias + c); a.x += b.x; a.y += b.y; a.z += b.z; a.w += b.w; }
            if (resid) {
                const float4 b = *(const float4*)(resid + (size_t)row * D + c);
                a.x += b.x; a.y += b.y; a.z += b.z; a.w += b.w;
            }
            v[i] = a;
            if (y_out) *(float4*)(y_out + (size_t)row * D + c) = a;     // pre-LN residual stream (ViT blocks)
        }
    }
    ln_row<T, MAXV>(v, nv, lane, D, gamma, beta, eps, out_t ? out_t + (size_t)row * D : nullptr,
              out_f ? out_f + (size_t)row * D : nullptr);
}

// Decode-sized variant (a few hundred rows): one 256-thread block per row, one float4 per thread, every partial / bias /
// residual / gamma / beta load issued before the first add -> one memory round trip, then two LDS cross-wave reductions.
// Same summation order over slices as the wave-per-row kernel.
template <typename T>
__global__ __launch_bounds__(256) void reduce_layernorm_row_kernel(const float* __restrict__ part, int S,
                                                                    const float* __restrict__ bias,
                                                                    const float* __restrict__ resid,
                                                                    const float* __restrict__ gamma,
                                                                    const float* __restrict__ beta, float eps, T* out_t,
                                                                    float* out_f, float* y_out, int M, int D,
                                                                    const int* __restrict__ n_rows) {
#pragma clang fp contract(off)
    __shared__ float sp[2][256];
    const int row = blockIdx.x, tid = threadIdx.x, c = tid * 4;
    if (n_rows && row >= *n_rows) return;                // (uniform over the block: before any barrier)
    const bool act = c < D;
    const float4 z4 = make_float4(0.f, 0.f, 0.f, 0.f);
    float4 pz[8], bb = z4, rr = z4, g = z4, be = z4;
#pragma unroll
    for (int z = 0; z < 8; ++z) pz[z] = (act && z < S) ? *(const float4*)(part + ((size_t)z * M + row) * D + c) : z4;
    if (act) {
        if (bias) bb = *(const float4*)(bias + c);
        if (resid) rr = *(const float4*)(resid + (size_t)row * D + c);
        g = *(const float4*)(gamma + c);
        be = *(const float4*)(beta + c);
    }
    float4 a = pz[0];
#pragma unroll
    for (int z = 1; z < 8; ++z)
        if (z < S) { a.x += pz[z].x; a.y += pz[z].y; a.z += pz[z].z; a.w += pz[z].w; }
    for (int z = 8; z < S; ++z) {
        const float4 b = act ? *(const float4*)(part + ((size_t)z * M + row) * D + c) : z4;
        a.x += b.x; a.y += b.y; a.z += b.z; a.w += b.w;
    }
    if (bias) { a.x += bb.x; a.y += bb.y; a.z += bb.z; a.w += bb.w; }
    if (resid) { a.x += rr.x; a.y += rr.y; a.z += rr.z; a.w += rr.w; }
    if (act && y_out) *(float4*)(y_out + (size_t)row * D + c) = a;
    // statistics in exactly the wave-per-row kernel's order (thread j + 64 i holds that kernel's lane j, vector i), so a
    // row normalises to the same bits whichever kernel the row count selects (batch invariance)
    const int lane = tid & 63;
    sp[0][tid] = act ? a.x + a.y + a.z + a.w : 0.f;
    __syncthreads();
    const float mean = wave_sum(((sp[0][lane] + sp[0][64 + lane]) + sp[0][128 + lane]) + sp[0][192 + lane]) / (float)D;
    const float dx = a.x - mean, dy = a.y - mean, dz = a.z - mean, dw = a.w - mean;
    sp[1][tid] = act ? dx * dx + dy * dy + dz * dz + dw * dw : 0.f;
    __syncthreads();
    const float rstd =
        1.0f / sqrtf(wave_sum(((sp[1][lane] + sp[1][64 + lane]) + sp[1][128 + lane]) + sp[1][192 + lane]) / (float)D + eps);
    if (act) {
        float4 o;
        o.x = dx * rstd * g.x + be.x; o.y = dy * rstd * g.y + be.y; o.z = dz * rstd * g.z + be.z; o.w = dw * rstd * g.w + be.w;
        if (out_f) *(float4*)(out_f + (size_t)row * D + c) = o;
        if (out_t) store4(out_t + (size_t)row * D, c, o);
    }
}

// The same consumer for rows wider than 1024 (OPT-2.7b: 2560) when there are only a few of them (a decode step: 32 rows):
// one 256-thread workgroup per row, up to 3 vectors per thread, every slice load in flight at once.  (The wave-per-row
// kernel walks such a row with 64 lanes: 19 us per launch at 32 x 2560 x 4 slices, measured; this one is bound by the launch.)
template <typename T>
__global__ __launch_bounds__(256) void reduce_layernorm_wide_kernel(const float* __restrict__ part, int S,
                                                                     const float* __restrict__ bias,
                                                                     const float* resid,   // may alias y_out (in-place residual stream)
                                                                     const float* __restrict__ gamma,
                                                                     const float* __restrict__ beta, float eps, T* out_t,
                                                                     float* out_f, float* y_out, int M, int D) {
#pragma clang fp contract(off)
    __shared__ float sp[2][4];
    const int row = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const float4 z4 = make_float4(0.f, 0.f, 0.f, 0.f);
    float4 a[3];
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        const int c = tid * 4 + i * 1024;
        a[i] = z4;
        if (c < D) {
            float4 pz[4];                          // the first four slices are loaded before the first add
#pragma unroll
            for (int z = 0; z < 4; ++z) pz[z] = z < S ? *(const float4*)(part + ((size_t)z * M + row) * D + c) : z4;
            a[i] = pz[0];
#pragma unroll
            for (int z = 1; z < 4; ++z)
                if (z < S) { a[i].x += pz[z].x; a[i].y += pz[z].y; a[i].z += pz[z].z; a[i].w += pz[z].w; }
            for (int z = 4; z < S; ++z) {
                const float4 b = *(const float4*)(part + ((size_t)z * M + row) * D + c);
                a[i].x += b.x; a[i].y += b.y; a[i].z += b.z; a[i].w += b.w;
            }
            if (bias) { const float4 b = *(const float4*)(bias + c); a[i].x += b.x; a[i].y += b.y; a[i].z += b.z; a[i].w += b.w; }
            if (resid) { const float4 b = *(const float4*)(resid + (size_t)row * D + c); a[i].x += b.x; a[i].y += b.y; a[i].z += b.z; a[i].w += b.w; }
            if (y_out) *(float4*)(y_out + (size_t)row * D + c) = a[i];
            s += (a[i].x + a[i].y) + (a[i].z + a[i].w);
        }
    }
    s = wave_sum(s);
    if (lane == 0) sp[0][wv] = s;
    __syncthreads();
    const float mean = (((sp[0][0] + sp[0][1]) + sp[0][2]) + sp[0][3]) / (float)D;
    float q = 0.f;
#pragma unroll
    for (int i = 0; i < 3; ++i)
        if (tid * 4 + i * 1024 < D) {
            const float dx = a[i].x - mean, dy = a[i].y - mean, dz = a[i].z - mean, dw = a[i].w - mean;
            q += (dx * dx + dy * dy) + (dz * dz + dw * dw);
        }
    q = wave_sum(q);
    if (lane == 0) sp[1][wv] = q;
    __syncthreads();
    const float rstd = 1.0f / sqrtf((((sp[1][0] + sp[1][1]) + sp[1][2]) + sp[1][3]) / (float)D + eps);
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        const int c = tid * 4 + i * 1024;
        if (c < D) {
            const float4 g = *(const float4*)(gamma + c), be = *(const float4*)(beta + c);
            float4 o;
            o.x = (a[i].x - mean) * rstd * g.x + be.x; o.y = (a[i].y - mean) * rstd * g.y + be.y;
            o.z = (a[i].z - mean) * rstd * g.z + be.z; o.w = (a[i].w - mean) * rstd * g.w + be.w;
            if (out_f) *(float4*)(out_f + (size_t)row * D + c) = o;
            if (out_t) store4(out_t + (size_t)row * D, c, o);
        }
    }
}

template <typename T>
__global__ __launch_bounds__(256) void embed_kernel(const int* __restrict__ seq, int seq_ld, int t,
                                                    const float* __restrict__ word, const float* __restrict__ pos,
                                                    const float* __restrict__ gamma, const float* __restrict__ beta,
                                                    float eps, T* out_t, float* out_f, float* y_out, int R, int D, RowMap map) {
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;      // (compact) output row
    if (row >= R || (map.n && row >= *map.n)) return;
    const int tok = seq[(size_t)(map.live ? map.live[row] : row) * seq_ld + t];
    const int nv = (D + 255) / 256;
    float4 v[LN_MAXV];
    const float* w = word + (size_t)tok * D;
    const float* p = pos + (size_t)t * D;
#pragma unroll
    for (int i = 0; i < LN_MAXV; ++i) {
        const int c = lane * 4 + i * 256;
        if (i < nv && c < D) {
            float4 a = *(const float4*)(w + c), b = *(const float4*)(p + c);
            v[i].x = a.x + b.x; v[i].y = a.y + b.y; v[i].z = a.z + b.z; v[i].w = a.w + b.w;
            if (y_out) *(float4*)(y_out + (size_t)row * D + c) = v[i];      // raw sum = pre-LN residual stream (CoCa)
        }
    }
    ln_row<T>(v, nv, lane, D, gamma, beta, eps, out_t ? out_t + (size_t)row * D : nullptr,
              out_f ? out_f + (size_t)row * D : nullptr);
}

// Sentence-encoder input embeddings (HF BertEmbeddings, HF:models/bert/modeling_bert.py `BertEmbeddings.forward`):
// (word[id] + token_type[0]) + position[l], then LayerNorm.  One wave per token row; row = b * L + l.
template <typename T>
__global__ __launch_bounds__(256) void embed_tokens_kernel(const int* __restrict__ ids, int L,
                                                           const float* __restrict__ word, const float* __restrict__ pos,
                                                           const float* __restrict__ type0,
                                                           const float* __restrict__ gamma, const float* __restrict__ beta,
                                                           float eps, T* out_t, float* out_f, int R, int D, int V) {
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (row >= R) return;
    const int tok = min(max(ids[row], 0), V - 1), l = row % L;      // ids are caller data: never index outside the table
    const int nv = (D + 255) / 256;
    float4 v[LN_MAXV];
    const float* w = word + (size_t)tok * D;
    const float* p = pos + (size_t)l * D;
#pragma unroll
    for (int i = 0; i < LN_MAXV; ++i) {
        const int c = lane * 4 + i * 256;
        if (i < nv && c < D) {
            const float4 a = *(const float4*)(w + c), t = *(const float4*)(type0 + c), b = *(const float4*)(p + c);
            v[i].x = (a.x + t.x) + b.x; v[i].y = (a.y + t.y) + b.y; v[i].z = (a.z + t.z) + b.z; v[i].w = (a.w + t.w) + b.w;
        }
    }
    ln_row<T>(v, nv, lane, D, gamma, beta, eps, out_t + (size_t)row * D, out_f + (size_t)row * D);
}

// sentence-transformers Pooling(mean over the attention mask) + Normalize (F.normalize, eps 1e-12): one block per
// sentence, one thread per 4 columns.
__global__ __launch_bounds__(256) void mean_pool_normalize_kernel(const float* __restrict__ x, const int* __restrict__ lens,
                                                                  int L, int D, float* __restrict__ out) {
    __shared__ float red[4];
    const int b = blockIdx.x, tid = threadIdx.x, c = tid * 4;
    const int n = min(max(lens[b], 1), L);
    float4 a = make_float4(0.f, 0.f, 0.f, 0.f);
    if (c < D)
        for (int l = 0; l < n; ++l) {
            const float4 v = *(const float4*)(x + ((size_t)b * L + l) * D + c);
            a.x += v.x; a.y += v.y; a.z += v.z; a.w += v.w;
        }
    const float inv = 1.0f / (float)n;
    a.x *= inv; a.y *= inv; a.z *= inv; a.w *= inv;
    float q = wave_sum(c < D ? a.x * a.x + a.y * a.y + a.z * a.z + a.w * a.w : 0.f);
    if ((tid & 63) == 0) red[tid >> 6] = q;
    __syncthreads();
    const float nrm = fmaxf(sqrtf(((red[0] + red[1]) + red[2]) + red[3]), 1e-12f);
    if (c < D) {
        a.x /= nrm; a.y /= nrm; a.z /= nrm; a.w /= nrm;
        *(float4*)(out + (size_t)b * D + c) = a;
    }
}

// ------------------------------------------------------------------------------------------------
// Greedy step (HF:generation/utils.py:2925-2937): next = argmax (first maximal index, like torch.argmax);
// finished rows emit pad; a row finishes when it emits EOS or reaches max_len.
__global__ __launch_bounds__(256) void greedy_select_kernel(const float* __restrict__ logits, int ld, int V,
                                                            int* __restrict__ seq, int seq_ld, int t, int max_len,
                                                            int eos, int pad, int* __restrict__ finished,
                                                            int* __restrict__ out_len, int min_len, int force_eos, RowMap map) {
    // min_len > 0: EOS cannot win while the row has fewer than min_len tokens (HF MinLengthLogitsProcessor, used by the
    // reference's CoCa loop coca_model.py:235-240); force_eos (= "this is the CoCa loop"): the last position is EOS
    // (coca_model.py:317-318) and a sampled pad id ends the row (:305)
    const int crow = blockIdx.x, tid = threadIdx.x;      // logits row; the caption it belongs to is row map.live[crow]
    if (map.n && crow >= *map.n) return;                 // (uniform over the block: before any barrier)
    const int row = map.live ? map.live[crow] : crow;
    const bool mask_eos = min_len > 0 && t + 1 < min_len;
    const float* x = logits + (size_t)crow * ld;
    float best = -INFINITY;
    int bi = 0x7fffffff;
    // 16-byte loads, 4 independent chunks in flight per thread; ascending index order inside a thread keeps "first
    // maximal index" with a strict > compare
    const int V4 = V >> 2;
    for (int c0 = tid; c0 < V4; c0 += 1024) {
        float4 v[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int c = c0 + u * 256;
            v[u] = c < V4 ? *(const float4*)(x + 4 * c) : make_float4(-INFINITY, -INFINITY, -INFINITY, -INFINITY);
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int i = 4 * (c0 + u * 256);
            if (mask_eos) {          // component-wise: a dynamic index would push v[] into scratch
                if (eos == i) v[u].x = -INFINITY;
                if (eos == i + 1) v[u].y = -INFINITY;
                if (eos == i + 2) v[u].z = -INFINITY;
                if (eos == i + 3) v[u].w = -INFINITY;
            }
            if (v[u].x > best) { best = v[u].x; bi = i; }
            if (v[u].y > best) { best = v[u].y; bi = i + 1; }
            if (v[u].z > best) { best = v[u].z; bi = i + 2; }
            if (v[u].w > best) { best = v[u].w; bi = i + 3; }
        }
    }
    for (int i = (V4 << 2) + tid; i < V; i += 256) {
        const float v = (mask_eos && i == eos) ? -INFINITY : x[i];
        if (v > best) { best = v; bi = i; }
    }
    __shared__ float sv[256];
    __shared__ int si[256];
    sv[tid] = best; si[tid] = bi;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
        if (tid < o) {
            float v2 = sv[tid + o]; int i2 = si[tid + o];
            if (v2 > sv[tid] || (v2 == sv[tid] && i2 < si[tid])) { sv[tid] = v2; si[tid] = i2; }
        }
        __syncthreads();
    }
    if (tid == 0) {
        int fin = finished[row];
        int tok = fin ? pad : si[0];
        if (!fin && force_eos && t + 2 >= max_len) tok = eos;
        seq[(size_t)row * seq_ld + t + 1] = tok;
        if (!fin) {
            if (tok == eos || t + 2 >= max_len) { finished[row] = 1; out_len[row] = t + 2; }
            // the reference's CoCa loop also stops a row whose newest token IS the pad id (coca_model.py:305: `mask =
            // last == eos | last == pad`; pad id 0 is an ordinary vocabulary entry there): it emits pad from then on and
            // gets no EOS.  The row's length then excludes that pad.
            else if (force_eos && tok == pad) { finished[row] = 1; out_len[row] = t + 1; }
        }
    }
}

__global__ void fill_i32_kernel(int* p, int v, size_t n) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) p[i] = v;
}
__global__ void fill_f32_kernel(float* p, float v, size_t n) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) p[i] = v;
}
__global__ void copy_f32_kernel(const float* s, float* d, size_t n) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) d[i] = s[i];
}

#define CAP_DISPATCH_T(dtype, MACRO)                                                                                  \
    do { if ((dtype) == CAP_DT_BF16) MACRO(bf16_t); else if ((dtype) == CAP_DT_G8) MACRO(g8_t); else MACRO(float); } while (0)

inline int grid_for(size_t n, int per_block) {
    size_t g = (n + per_block - 1) / per_block;
    return (int)(g < 1 ? 1 : (g > 2048 ? 2048 : g));
}

}  // namespace

int launch_convert(int dtype, const float* src, void* dst, size_t n, hipStream_t s) {
    if (dtype == CAP_DT_G8 && n % 8 != 0) { cap_set_error("convert: a G8 buffer holds whole groups of 8 elements (n=%zu)", n); return -1; }
#define CAP_CV(TT) hipLaunchKernelGGL(convert_kernel<TT>, dim3(grid_for(n, 1024)), dim3(256), 0, s, src, (TT*)dst, n)
    CAP_DISPATCH_T(dtype, CAP_CV);
#undef CAP_CV
    CAP_HIP_CHECK(hipGetLastError());
    return 0;
}

int launch_convert2d(int dtype, const float* src, void* dst, int rows, int cols, int dst_ld, hipStream_t s, float scale) {
    const size_t n = (size_t)rows * cols;
    if (dtype == CAP_DT_G8 && dst_ld % 8 != 0) { cap_set_error("convert2d: G8 rows need ld %% 8 == 0 (ld=%d)", dst_ld); return -1; }
#define CAP_CV(TT) hipLaunchKernelGGL(convert2d_kernel<TT>, dim3(grid_for(n, 256)), dim3(256), 0, s, src, (TT*)dst, rows, cols, dst_ld, scale)
    CAP_DISPATCH_T(dtype, CAP_CV);
#undef CAP_CV
    CAP_HIP_CHECK(hipGetLastError());
    return 0;
}

int launch_convert2d_t(int dtype, const float* src, void* dst, int rows, int cols, hipStream_t s, float scale) {
    const size_t n = (size_t)rows * cols;
    if (dtype == CAP_DT_G8 && rows % 8 != 0) { cap_set_error("convert2d_t: G8 rows need ld %% 8 == 0 (ld=%d)", rows); return -1; }
#define CAP_CV(TT) hipLaunchKernelGGL(convert2d_t_kernel<TT>, dim3(grid_for(n, 256)), dim3(256), 0, s, src, (TT*)dst, rows, cols, scale)
    CAP_DISPATCH_T(dtype, CAP_CV);
#undef CAP_CV
    CAP_HIP_CHECK(hipGetLastError());
    return 0;
}

// The same gather with one thread per 8 consecutive pixels of an image row (patch sizes that are multiples of 8: 16 for ViT-B/16):
// 8 consecutive k of one patch row = one 32-byte G8 group (16 bytes of bf16), written as whole pieces instead of one 2-byte store
// per half per element (155 us per 256 frames at 224 x 224 before; the values are those of patchify_kernel: same expression).
template <typename T>
__global__ void patchify8_kernel(const void* __restrict__ pixels, int fmt, int B, int img, int ps, int Kpad,
                                 T* __restrict__ out, float m0, float m1, float m2, float s0, float s1, float s2) {
    const int G = img / ps, W8 = img / 8;
    const size_t total = (size_t)B * 3 * img * W8;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const int x = (int)(i % W8) * 8;
        size_t r = i / W8;
        const int y = (int)(r % img); r /= img;
        const int c = (int)(r % 3);
        const int b = (int)(r / 3);
        float v[8];
        if (fmt == 0) {
            const float4* src = (const float4*)((const float*)pixels + (((size_t)b * 3 + c) * img + y) * img + x);
            const float4 a = src[0], d = src[1];
            v[0] = a.x; v[1] = a.y; v[2] = a.z; v[3] = a.w; v[4] = d.x; v[5] = d.y; v[6] = d.z; v[7] = d.w;
        } else {
            const unsigned char* src = (const unsigned char*)pixels + (((size_t)b * img + y) * img + x) * 3 + c;
            const float mean = c == 0 ? m0 : (c == 1 ? m1 : m2), sd = c == 0 ? s0 : (c == 1 ? s1 : s2);
#pragma unroll
            for (int j = 0; j < 8; ++j) v[j] = ((float)src[3 * j] * (1.0f / 255.0f) - mean) / sd;
        }
        const int py = y / ps, dy = y - py * ps, px = x / ps, dx = x - px * ps;
        T* row = out + ((size_t)b * G * G + (size_t)py * G + px) * Kpad;
        const int k = c * ps * ps + dy * ps + dx;                    // a multiple of 8
        store4(row, k, make_float4(v[0], v[1], v[2], v[3]));
        store4(row, k + 4, make_float4(v[4], v[5], v[6], v[7]));
    }
}

int launch_patchify(int dtype, const void* pixels, int fmt, int B, int img, int ps, int Kpad, void* out,
                    const float* mean, const float* stdv, hipStream_t s) {
    if (img % ps != 0 || 3 * ps * ps > Kpad) {
        cap_set_error("patchify: image %d not divisible by patch %d or Kpad %d too small", img, ps, Kpad);
        return -1;
    }
    const size_t n = (size_t)B * 3 * img * img;
    const float m0 = mean ? mean[0] : 0.f, m1 = mean ? mean[1] : 0.f, m2 = mean ? mean[2] : 0.f;
    const float s0 = stdv ? stdv[0] : 1.f, s1 = stdv ? stdv[1] : 1.f, s2 = stdv ? stdv[2] : 1.f;
    if (ps % 8 == 0 && ((uintptr_t)pixels & 15) == 0) {
#define CAP_PF8(TT) hipLaunchKernelGGL(patchify8_kernel<TT>, dim3(grid_for(n / 8, 256)), dim3(256), 0, s, pixels, fmt, B, img, ps, Kpad, (TT*)out, m0, m1, m2, s0, s1, s2)
        CAP_DISPATCH_T(dtype, CAP_PF8);
#undef CAP_PF8
        CAP_HIP_CHECK(hipGetLastError());
        return 0;
    }
#define CAP_PF(TT) hipLaunchKernelGGL(patchify_kernel<TT>, dim3(grid_for(n, 256)), dim3(256), 0, s, pixels, fmt, B, img, ps, Kpad, (TT*)out, m0, m1, m2, s0, s1, s2)
    CAP_DISPATCH_T(dtype, CAP_PF);
#undef CAP_PF
    CAP_HIP_CHECK(hipGetLastError());
    return 0;
}

int launch_cls_rows(const float* cls, const float* pos, float* X, int B, int tokens, int D, hipStream_t s) {
    hipLaunchKernelGGL(cls_rows_kernel, dim3((B * D + 255) / 256), dim3(256), 0, s, cls, pos, X, B, tokens, D);
    CAP_HIP_CHECK(hipGetLastError());
    return 0;
}

int launch_layernorm(int dtype, const float* in, int ld_in, const float* gamma, const float* beta, float eps,
                     void* out_t, float* out_f, int M, int D, hipStream_t s) {
    if (D % 4 != 0 || D > 256 * LN_MAXV || (ld_in % 4) != 0 || (dtype == CAP_DT_G8 && D % 8 != 0)) {
        cap_set_error("layernorm: unsupported width %d (ld %d)", D, ld_in);
        return -1;
    }
#define CAP_LN(TT, MV)                                                                                                  \
    hipLaunchKernelGGL((layernorm_kernel<TT, MV>), dim3((M + 3) / 4), dim3(256), 0, s, in, ld_in, gamma, beta, eps,    \
                       (TT*)out_t, out_f, M, D)
#define CAP_LN_T(TT) do { if (D <= 1024) CAP_LN(TT, 4); else if (D <= 2048) CAP_LN(TT, 8); else CAP_LN(TT, LN_MAXV); } while (0)
    CAP_DISPATCH_T(dtype, CAP_LN_T);
#undef CAP_LN_T
#undef CAP_LN
    CAP_HIP_CHECK(hipGetLastError());
    return 0;
}

int launch_reduce_layernorm(int dtype, const void* part, int S, const float* bias, const float* resid,
                            const float* gamma, const float* beta, float eps, void* out_t, float* out_f, float* y_out,
                            int M, int D, hipStream_t s, bool per_row_block, bool part_in_t, const int* n_rows) {
    if (n_rows && per_row_block && D > 1024) { cap_set_error("reduce_layernorm: the wide block kernel takes no row count from the device"); return -1; }
    if (per_row_block && part_in_t) { cap_set_error("reduce_layernorm: the per-row-block kernel takes fp32 partial sums"); return -1; }
    if (D % 4 != 0 || D > 256 * LN_MAXV || S < 1 || (dtype == CAP_DT_G8 && D % 8 != 0)) { cap_set_error("reduce_layernorm: unsupported width %d / slices %d", D, S); return -1; }
    // per_row_block: the decoder's choice up to a few hundred rows (latency-bound).  For rows up to 1024 wide the block-per-row
    // kernel forms the wave-per-row kernel's sums in its order (slices, bias, residual; statistics: thread j + 64 i = lane j,
    // vector i) and the two agree bit for bit (tests/test_kernels_gpu.py::test_reduce_layernorm_kernels_agree_bit_for_bit), so a
    // caller may pick by row count there; the WIDE block kernel (rows beyond 1024) has its own order: by path only.
    if (per_row_block && D <= 1024) {
#define CAP_RR(TT) hipLaunchKernelGGL(reduce_layernorm_row_kernel<TT>, dim3(M), dim3(256), 0, s, (const float*)part, S, bias, resid, gamma, beta, eps, (TT*)out_t, out_f, y_out, M, D, n_rows)
        CAP_DISPATCH_T(dtype, CAP_RR);
#undef CAP_RR
        CAP_HIP_CHECK(hipGetLastError());
        return 0;
    }
    if (per_row_block && D <= 3072) {     // wide rows, decode-sized row count (the caller's choice, as above)
#define CAP_RW(TT) hipLaunchKernelGGL(reduce_layernorm_wide_kernel<TT>, dim3(M), dim3(256), 0, s, (const float*)part, S, bias, resid, gamma, beta, eps, (TT*)out_t, out_f, y_out, M, D)
        CAP_DISPATCH_T(dtype, CAP_RW);
#undef CAP_RW
        CAP_HIP_CHECK(hipGetLastError());
        return 0;
    }
    const int wpb = M >= 2048 ? 4 : 1;
    const dim3 grid((M + wpb - 1) / wpb), block(64 * wpb);
#define CAP_RLN(TT, PP, MV)                                                                                             \
    hipLaunchKernelGGL((reduce_layernorm_kernel<TT, PP, MV>), grid, block, 0, s, (const PP*)part, S, bias, resid, gamma,  \
                       beta, eps, (TT*)out_t, out_f, y_out, M, D, n_rows)
#define CAP_RLN_T(TT, PP) do { if (D <= 1024) CAP_RLN(TT, PP, 4); else if (D <= 2048) CAP_RLN(TT, PP, 8); else CAP_RLN(TT, PP, LN_MAXV); } while (0)
    if (dtype == CAP_DT_BF16 && part_in_t) CAP_RLN_T(bf16_t, bf16_t);
    else if (dtype == CAP_DT_BF16) CAP_RLN_T(bf16_t, float);
    else if (dtype == CAP_DT_G8) CAP_RLN_T(g8_t, float);
    else CAP_RLN_T(float, float);
#undef CAP_RLN_T
#undef CAP_RLN
    CAP_HIP_CHECK(hipGetLastError());
    return 0;
}

// Split-K consumer without LayerNorm: out[r][c] = act(sum_z part[z][r][c] + bias[c]) in the compute type (OPT decode: the
// fused q|k|v projection and fc1 + ReLU).  act: 0 none, 2 ReLU.
template <typename T>
__global__ __launch_bounds__(256) void reduce_bias_act_kernel(const float* __restrict__ part, int S, const float* __restrict__ bias,
                                                              T* __restrict__ out, int M, int N, int act) {
    const size_t n4 = (size_t)M * N / 4;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (size_t)gridDim.x * blockDim.x) {
        const size_t e = i * 4;
        const int c = e % N;
        float4 a = *(const float4*)(part + e);
        for (int z = 1; z < S; ++z) {
            const float4 b = *(const float4*)(part + (size_t)z * M * N + e);
            a.x += b.x; a.y += b.y; a.z += b.z; a.w += b.w;
        }
        if (bias) { const float4 b = *(const float4*)(bias + c); a.x += b.x; a.y += b.y; a.z += b.z; a.w += b.w; }
        if (act == 2) { a.x = fmaxf(a.x, 0.f); a.y = fmaxf(a.y, 0.f); a.z = fmaxf(a.z, 0.f); a.w = fmaxf(a.w, 0.f); }
        store4(out + (e - c), c, a);
    }
}

int launch_reduce_bias_act(int dtype, const float* part, int S, const float* bias, void* out, int M, int N, int act, hipStream_t s) {
    if (N % 4 != 0) { cap_set_error("reduce_bias_act: N must be a multiple of 4"); return -1; }
    const int grid = (int)std::min<size_t>(((size_t)M * N / 4 + 255) / 256, 2048);
#define CAP_RB(TT) hipLaunchKernelGGL(reduce_bias_act_kernel<TT>, dim3(grid), dim3(256), 0, s, part, S, bias, (TT*)out, M, N, act)
    CAP_DISPATCH_T(dtype, CAP_RB);
#undef CAP_RB
    CAP_HIP_CHECK(hipGetLastError());
    return 0;
}

int launch_embed_tokens(int dtype, const int* ids, int L, const float* word, const float* pos, const float* type0,
                        const float* gamma, const float* beta, float eps, void* out_t, float* out_f, int R, int D,
                        hipStream_t s, int V) {
    if (D % 4 != 0 || D > 256 * LN_MAXV || V < 1) { cap_set_error("embed_tokens: unsupported width %d / vocabulary %d", D, V); return -1; }
#define CAP_ET(TT) hipLaunchKernelGGL(embed_tokens_kernel<TT>, dim3((R + 3) / 4), dim3(256), 0, s, ids, L, word, pos, type0, gamma, beta, eps, (TT*)out_t, out_f, R, D, V)
    CAP_DISPATCH_T(dtype, CAP_ET);
#undef CAP_ET
    CAP_HIP_CHECK(hipGetLastError());
    return 0;
}

int launch_mean_pool_normalize(const float* x, const int* lens, int B, int L, int D, float* out, hipStream_t s) {
    if (D % 4 != 0 || D > 1024) { cap_set_error("mean_pool: unsupported width %d", D); return -1; }
    hipLaunchKernelGGL(mean_pool_normalize_kernel, dim3(B), dim3(256), 0, s, x, lens, L, D, out);
    CAP_HIP_CHECK(hipGetLastError());
    return 0;
}

int launch_embed(int dtype, const int* seq, int seq_ld, int t, const float* word, const float* pos,
                 const float* gamma, const float* beta, float eps, void* out_t, float* out_f, int R, int D,
                 hipStream_t s, float* y_out, RowMap map) {
    if (D % 4 != 0 || D > 256 * LN_MAXV) { cap_set_error("embed: unsupported width %d", D); return -1; }
#define CAP_EM(TT) hipLaunchKernelGGL(embed_kernel<TT>, dim3((R + 3) / 4), dim3(256), 0, s, seq, seq_ld, t, word, pos, gamma, beta, eps, (TT*)out_t, out_f, y_out, R, D, map)
    CAP_DISPATCH_T(dtype, CAP_EM);
#undef CAP_EM
    CAP_HIP_CHECK(hipGetLastError());
    return 0;
}

int launch_greedy_select(const float* logits, int ld, int V, int* seq, int seq_ld, int t, int max_len, int eos,
                         int pad, int* finished, int* out_len, int R, hipStream_t s, int min_len, int force_eos, RowMap map) {
    hipLaunchKernelGGL(greedy_select_kernel, dim3(R), dim3(256), 0, s, logits, ld, V, seq, seq_ld, t, max_len, eos, pad,
                       finished, out_len, min_len, force_eos, map);
    CAP_HIP_CHECK(hipGetLastError());
    return 0;
}

// The open rows in ascending order (stable: a caption's compact position only ever moves towards the front) and their count.
__global__ __launch_bounds__(1024) void compact_rows_kernel(const int* __restrict__ finished, int R, int* __restrict__ live,
                                                            int* __restrict__ n_live) {
    __shared__ int wsum[16];
    __shared__ int base;
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    if (tid == 0) base = 0;
    __syncthreads();
    for (int r0 = 0; r0 < R; r0 += 1024) {
        const int r = r0 + tid;
        const bool on = r < R && finished[r] == 0;
        const unsigned long long b = __ballot(on);
        if (lane == 0) wsum[w] = __popcll(b);
        __syncthreads();
        int off = base;
        for (int i = 0; i < w; ++i) off += wsum[i];
        if (on) live[off + __popcll(b & ((1ull << lane) - 1ull))] = r;
        __syncthreads();
        if (tid == 0) {
            int sum = 0;
            for (int i = 0; i < 16; ++i) sum += wsum[i];
            base += sum;
        }
        __syncthreads();
    }
    if (tid == 0) *n_live = base;
}
int launch_compact_rows(const int* finished, int R, int* live, int* n_live, hipStream_t s) {
    hipLaunchKernelGGL(compact_rows_kernel, dim3(1), dim3(1024), 0, s, finished, R, live, n_live);
    CAP_HIP_CHECK(hipGetLastError());
    return 0;
}

__global__ void absmax_f32_kernel(const float* __restrict__ src, size_t n, unsigned int* out_bits) {
    float m = 0.f;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        const float a = fabsf(src[i]);
        m = (a > m || a != a) ? a : m;                   // a NaN sticks (and compares above every bound below)
    }
    const float w = (m != m) ? m : wave_max(m);
    // non-negative floats order like their bit patterns; a NaN's pattern is above +inf's
    if ((threadIdx.x & 63) == 0 || m != m) atomicMax(out_bits, __float_as_uint(w));
}
int launch_absmax_f32(const float* src, size_t n, unsigned int* out_bits, hipStream_t s) {
    hipLaunchKernelGGL(absmax_f32_kernel, dim3(grid_for(n, 256)), dim3(256), 0, s, src, n, out_bits);
    CAP_HIP_CHECK(hipGetLastError());
    return 0;
}

int launch_fill_i32(int* p, int v, size_t n, hipStream_t s) {
    hipLaunchKernelGGL(fill_i32_kernel, dim3(grid_for(n, 256)), dim3(256), 0, s, p, v, n);
    CAP_HIP_CHECK(hipGetLastError());
    return 0;
}
int launch_fill_f32(float* p, float v, size_t n, hipStream_t s) {
    hipLaunchKernelGGL(fill_f32_kernel, dim3(grid_for(n, 256)), dim3(256), 0, s, p, v, n);
    CAP_HIP_CHECK(hipGetLastError());
    return 0;
}
int launch_copy_f32(const float* src, float* dst, size_t n, hipStream_t s) {
    hipLaunchKernelGGL(copy_f32_kernel, dim3(grid_for(n, 256)), dim3(256), 0, s, src, dst, n);
    CAP_HIP_CHECK(hipGetLastError());
    return 0;
}

CAP_DEFINE_G8_CLAMP_READER(cap_g8_clamped_elementwise)
