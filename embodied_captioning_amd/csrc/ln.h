// LayerNorm building blocks shared by elementwise.hip and the fused split-K consumer in gemm.hip.
#pragma once
#include "common.h"

// LayerNorm, one wave per row, row held in registers (D <= 64*4*MAXV).  Two-pass mean/variance in fp32,
// biased variance, eps inside the sqrt (torch.nn.functional.layer_norm).
constexpr int LN_MAXV = 12;   // rows up to 3072 wide (OPT-2.7b: 2560)

// MAXV: vectors per lane the caller's row array holds (the launchers pick 4 / 8 / 12 by row width: a 768-wide row in a
// 12-vector kernel pays for the predicated tail - measured +21 % on the encoder LayerNorm).
template <typename T, int MAXV = LN_MAXV>
__device__ __forceinline__ void ln_row(const float4 (&v)[MAXV], int nv, int lane, int D, const float* gamma,
                                       const float* beta, float eps, T* out_t, float* out_f) {
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < MAXV; ++i)
        if (i < nv && lane * 4 + i * 256 < D) s += v[i].x + v[i].y + v[i].z + v[i].w;
    const float mean = wave_sum(s) / (float)D;
    float q = 0.f;
#pragma unroll
    for (int i = 0; i < MAXV; ++i)
        if (i < nv && lane * 4 + i * 256 < D) {
            float a = v[i].x - mean, b = v[i].y - mean, c = v[i].z - mean, d = v[i].w - mean;
            q += a * a + b * b + c * c + d * d;
        }
    const float rstd = 1.0f / sqrtf(wave_sum(q) / (float)D + eps);
#pragma unroll
    for (int i = 0; i < MAXV; ++i) {
        const int c = lane * 4 + i * 256;
        if (i < nv && c < D) {
            float4 g = *(const float4*)(gamma + c), bb = *(const float4*)(beta + c);
            float4 o;
            o.x = (v[i].x - mean) * rstd * g.x + bb.x; o.y = (v[i].y - mean) * rstd * g.y + bb.y;
            o.z = (v[i].z - mean) * rstd * g.z + bb.z; o.w = (v[i].w - mean) * rstd * g.w + bb.w;
            if (out_f) *(float4*)(out_f + c) = o;
            if (out_t) store4(out_t, c, o);
        }
    }
}


// 16 bytes of a split-K slab.  COHERENT: agent-scope loads (the slab was written by blocks of the same kernel, maybe on
// another XCD: must not be served from this XCD's L2).
template <bool COHERENT>
__device__ __forceinline__ float4 load_slab4(const float* p) {
    if constexpr (COHERENT) {
        float4 r;
        r.x = __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        r.y = __hip_atomic_load(p + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        r.z = __hip_atomic_load(p + 2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        r.w = __hip_atomic_load(p + 3, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        return r;
    } else {
        return *(const float4*)p;
    }
}

// One wave reduces row `row` of S fp32 split-K slices (+ bias + residual, summed in slice order), optionally writes the
// sum (y_out) and LayerNorms it.  Exactly the arithmetic of reduce_layernorm_kernel<T, float>.
template <typename T, bool COHERENT = false, int MAXV = LN_MAXV>
__device__ __forceinline__ void reduce_ln_row_wave(const float* __restrict__ part, int S, int M, int D, int row, int lane,
                                                   const float* __restrict__ bias, const float* resid,
                                                   const float* __restrict__ gamma, const float* __restrict__ beta,
                                                   float eps, T* out_t, float* out_f, float* y_out) {
    const int nv = (D + 255) / 256;
    float4 v[MAXV];
#pragma unroll
    for (int i = 0; i < MAXV; ++i) {
        const int c = lane * 4 + i * 256;
        if (i < nv && c < D) {
            float4 a = load_slab4<COHERENT>(part + (size_t)row * D + c);
            if (S == 4) {
                const float4 b1 = load_slab4<COHERENT>(part + ((size_t)1 * M + row) * D + c);
                const float4 b2 = load_slab4<COHERENT>(part + ((size_t)2 * M + row) * D + c);
                const float4 b3 = load_slab4<COHERENT>(part + ((size_t)3 * M + row) * D + c);
                a.x = ((a.x + b1.x) + b2.x) + b3.x; a.y = ((a.y + b1.y) + b2.y) + b3.y;
                a.z = ((a.z + b1.z) + b2.z) + b3.z; a.w = ((a.w + b1.w) + b2.w) + b3.w;
            } else {
#pragma unroll 4
                for (int z = 1; z < S; ++z) {
                    const float4 b = load_slab4<COHERENT>(part + ((size_t)z * M + row) * D + c);
                    a.x += b.x; a.y += b.y; a.z += b.z; a.w += b.w;
                }
            }
            if (bias) { const float4 b = *(const float4*)(bias + c); a.x += b.x; a.y += b.y; a.z += b.z; a.w += b.w; }
            if (resid) {
                const float4 b = *(const float4*)(resid + (size_t)row * D + c);
                a.x += b.x; a.y += b.y; a.z += b.z; a.w += b.w;
            }
            v[i] = a;
            if (y_out) *(float4*)(y_out + (size_t)row * D + c) = a;
        }
    }
    ln_row<T, MAXV>(v, nv, lane, D, gamma, beta, eps, out_t ? out_t + (size_t)row * D : nullptr,
              out_f ? out_f + (size_t)row * D : nullptr);
}
