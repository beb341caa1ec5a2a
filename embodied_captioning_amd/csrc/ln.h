// LayerNorm building block of elementwise.hip (one wave per row).
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
#pragma clang fp contract(off)      // as written, wherever it is inlined: the block-per-row consumer must give the same bits
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

// ln_row with gamma / beta already in registers (g[i], be[i] = the float4 at column lane * 4 + i * 256): the same expressions in
// the same order - a row normalises to the same bits - for callers that put the two loads in flight with the row's own loads
// instead of behind the statistics (the small-batch decode kernels: every dependent memory round trip is 2-3 us there).
template <typename T, int MAXV>
__device__ __forceinline__ void ln_row_regs(const float4 (&v)[MAXV], int nv, int lane, int D, const float4 (&g)[MAXV],
                                            const float4 (&be)[MAXV], float eps, T* out_t, float* out_f) {
#pragma clang fp contract(off)
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
            float4 o;
            o.x = (v[i].x - mean) * rstd * g[i].x + be[i].x; o.y = (v[i].y - mean) * rstd * g[i].y + be[i].y;
            o.z = (v[i].z - mean) * rstd * g[i].z + be[i].z; o.w = (v[i].w - mean) * rstd * g[i].w + be[i].w;
            if (out_f) *(float4*)(out_f + c) = o;
            if (out_t) store4(out_t, c, o);
        }
    }
}
