// Single-query decode attention as per-(row, head) UNIT functions: one wave computes one unit.  The kernels of attention.hip
// (one unit per wave of a grid) and the small-batch decode kernels of decode_small.hip (units inside a GEMM's prologue) call the
// SAME functions, so a row's context has the same bits whichever path its batch size selects.
// Everything is in an anonymous namespace: include from a translation unit that defines decode kernels.
#pragma once
#include "common.h"

namespace {

template <typename T> __device__ __forceinline__ void load8(const T* p, float (&v)[8]);
template <> __device__ __forceinline__ void load8<float>(const float* p, float (&v)[8]) {
    float4 a = *(const float4*)p, b = *(const float4*)(p + 4);
    v[0] = a.x; v[1] = a.y; v[2] = a.z; v[3] = a.w; v[4] = b.x; v[5] = b.y; v[6] = b.z; v[7] = b.w;
}
template <> __device__ __forceinline__ void load8<bf16_t>(const bf16_t* p, float (&v)[8]) {
    bf16x8 a = *(const bf16x8*)p;
#pragma unroll
    for (int i = 0; i < 8; ++i) v[i] = (float)a[i];
}

// Optional fused producer of the decode attention kernels: the q (and, for self-attention, the new k/v) projection arrives
// as split-K partial sums `part` fp32 [S][R][part_ld] (+ bias); the unit finishes the reduction, appends k/v of position
// n_keys-1 to the cache of its own row and attends - saving the GEMM's second round trip and its scatter epilogue.
struct QSource {
    const float* part;    // nullptr: q comes from the `q` tensor
    const float* bias;
    int S, part_ld, col0;  // q columns start at col0; k at col0 + Dh, v at col0 + 2*Dh when append_kv
    int append_kv;
};

// Split-K partials of 8 consecutive columns: issue() puts every slice's loads (and the bias) in flight without a
// dependent add between them; finish() sums them in slice order.  Up to 4 slices are unrolled (what the decode GEMMs produce);
// a slice beyond S enters as slice 0 with weight 0 (S is uniform: one scalar branch per slice, no load).
struct Part8 {
    f32x4 a[4], b[4], ba, bb;
    const float* p;
    size_t zs;
    __device__ __forceinline__ void issue(const QSource& qs, int R, int row, int col) {
        p = qs.part + (size_t)row * qs.part_ld + col;
        zs = (size_t)R * qs.part_ld;
#pragma unroll
        for (int z = 0; z < 4; ++z) {
            if (z < qs.S) {
                const float* pz = p + (size_t)z * zs;
                a[z] = *(const f32x4*)pz; b[z] = *(const f32x4*)(pz + 4);
            } else {
                a[z] = a[0]; b[z] = b[0];
            }
        }
        ba = *(const f32x4*)(qs.bias + col); bb = *(const f32x4*)(qs.bias + col + 4);
    }
    template <typename T>
    __device__ __forceinline__ void finish(const QSource& qs, float (&v)[8]) {
#pragma clang fp contract(off)
        f32x4 sa = a[0], sb = b[0];
#pragma unroll
        for (int z = 1; z < 4; ++z) {
            const float w = z < qs.S ? 1.f : 0.f;
            sa += a[z] * w; sb += b[z] * w;
        }
        for (int z = 4; z < qs.S; ++z) { sa += *(const f32x4*)(p + z * zs); sb += *(const f32x4*)(p + z * zs + 4); }
        sa += ba; sb += bb;
        // round through the compute dtype exactly like the unfused GEMM epilogue would have
#pragma unroll
        for (int i = 0; i < 4; ++i) { v[i] = to_f32(from_f32<T>(sa[i])); v[4 + i] = to_f32(from_f32<T>(sb[i])); }
    }
};

template <typename T> struct Raw8;
template <> struct Raw8<bf16_t> {
    bf16x8 r;
    __device__ __forceinline__ void load(const bf16_t* p) { r = *(const bf16x8*)p; }
    __device__ __forceinline__ void load_nt(const bf16_t* p) { r = __builtin_nontemporal_load((const bf16x8*)p); }
    __device__ __forceinline__ void load_row(const void* base, size_t ri, int dch) { load((const bf16_t*)base + ri * 64 + dch * 8); }
    __device__ __forceinline__ void load_row_nt(const void* base, size_t ri, int dch) { load_nt((const bf16_t*)base + ri * 64 + dch * 8); }
    __device__ __forceinline__ void zero() { for (int i = 0; i < 8; ++i) r[i] = (bf16_t)0.f; }
    __device__ __forceinline__ float get(int i) const { return (float)r[i]; }
    __device__ __forceinline__ void set(int i, float x) { r[i] = (bf16_t)x; }
    static constexpr bool scaled = false;
    __device__ __forceinline__ float scale() const { return 1.f; }
};
// KV16 (common.h): eight head dimensions dch * 8 .. + 7 of a row = 16 bytes of int16 + the row's fp32 scale (the same word for
// the 8 lanes of a key).  get(i) is the INTEGER as a float (one SDWA convert); the kernels apply scale() once per key to the
// score and to the probability instead of once per element.  Raw8<T>::load_row(base, ri, dch) reads those eight dimensions of
// the 64-wide KV row `ri` of a cache of element type T (the other specialisations: plain typed rows, scale() == 1).
typedef unsigned int u32x4_t __attribute__((ext_vector_type(4)));
template <> struct Raw8<kv16_t> {
    u32x4_t q; float sc;
    static constexpr bool scaled = true;
    __device__ __forceinline__ void load_row(const void* base, size_t ri, int dch) {
        q = *(const u32x4_t*)((const char*)base + kv16_row_off(ri) + dch * 16);
        sc = *(const float*)((const char*)base + kv16_scale_off(ri));
    }
    __device__ __forceinline__ void load_row_nt(const void* base, size_t ri, int dch) {
        q = __builtin_nontemporal_load((const u32x4_t*)((const char*)base + kv16_row_off(ri) + dch * 16));
        sc = __builtin_nontemporal_load((const float*)((const char*)base + kv16_scale_off(ri)));
    }
    __device__ __forceinline__ void zero() { q = 0u; sc = 0.f; }
    __device__ __forceinline__ float get(int i) const {
        const unsigned int w = q[i >> 1];
        return (float)((i & 1) ? (short)(w >> 16) : (short)(w & 0xFFFFu));
    }
    __device__ __forceinline__ float scale() const { return sc; }
};
template <> struct Raw8<float> {
    f32x4 a, b;
    __device__ __forceinline__ void load(const float* p) { a = *(const f32x4*)p; b = *(const f32x4*)(p + 4); }
    __device__ __forceinline__ void load_nt(const float* p) { load(p); }
    __device__ __forceinline__ void load_row(const void* base, size_t ri, int dch) { load((const float*)base + ri * 64 + dch * 8); }
    __device__ __forceinline__ void load_row_nt(const void* base, size_t ri, int dch) { load_row(base, ri, dch); }
    __device__ __forceinline__ void zero() { a = 0.f; b = 0.f; }
    __device__ __forceinline__ float get(int i) const { return i < 4 ? a[i] : b[i - 4]; }
    __device__ __forceinline__ void set(int i, float x) { if (i < 4) a[i] = x; else b[i - 4] = x; }
    static constexpr bool scaled = false;
    __device__ __forceinline__ float scale() const { return 1.f; }
};

// ---- short history (n_keys <= 8*NI): no LDS, no barriers.  All K and V loads of the wave are issued before any arithmetic so
// the whole history is one memory round trip.  out_row: element 0 of this row's context (columns h * 64 .. are written by the
// 8 lanes with ksub == 0).  write_kv: append the new position's k / v to the cache (the small-batch path computes a unit in
// several workgroups; one of them writes).  q_ready: see decode_attention_online_unit.
template <typename T, int NI, typename TO>
__device__ __forceinline__ void decode_attention_wave_unit(const T* __restrict__ q, T* __restrict__ kbase, T* __restrict__ vbase,
                                                           const int* __restrict__ anc, int anc_ld, int rows_per_kv, int kv_ld,
                                                           int n_keys, TO* out_row, int R, int H, const QSource& qs, int row, int h,
                                                           int lane, bool write_kv, const float* q_ready = nullptr, int prow = -1) {
#pragma clang fp contract(off)      // as written, wherever it is inlined (batch kernel / small-batch prologue): same bits
    // prow >= 0 (compacted decode loop, ops.h RowMap): the row of the split-K partial sums / the `q` tensor, when it is not `row` -
    // `row` stays the row that owns the caches and the ancestry
    const int pr = prow >= 0 ? prow : row;
    const int Dh = H * 64;
    const int ksub = lane >> 3, dch = lane & 7;
    const bool fused_kv = qs.part != nullptr && qs.append_kv;
    // program order = issue order: ancestry indices, then the split-K partials, then the history (which waits on the
    // indices only), then arithmetic
    int srcs[NI];
#pragma unroll
    for (int i = 0; i < NI; ++i) {
        const int key = i * 8 + ksub;
        srcs[i] = (anc && key < n_keys) ? anc[(size_t)row * anc_ld + key] : row / rows_per_kv;
    }
    Part8 pq, pk, pv;
    if (qs.part) pq.issue(qs, R, pr, qs.col0 + h * 64 + dch * 8);
    if (fused_kv) {
        pk.issue(qs, R, pr, qs.col0 + Dh + h * 64 + dch * 8);
        pv.issue(qs, R, pr, qs.col0 + 2 * Dh + h * 64 + dch * 8);
    }
    Raw8<T> kk[NI], vv[NI];
#pragma unroll
    for (int i = 0; i < NI; ++i) {
        const int key = i * 8 + ksub;
        if (key < n_keys && !(fused_kv && key == n_keys - 1)) {
            const size_t o = (((size_t)srcs[i] * H + h) * kv_ld + key) * 64 + dch * 8;
            kk[i].load(kbase + o);
            vv[i].load(vbase + o);
        } else {
            kk[i].zero(); vv[i].zero();
        }
    }
    float qv[8];
    if (q_ready) {
#pragma unroll
        for (int e = 0; e < 8; ++e) qv[e] = q_ready[dch * 8 + e];
    } else if (qs.part) {
        pq.finish<T>(qs, qv);
    } else {
        load8<T>(q + (size_t)pr * Dh + h * 64 + dch * 8, qv);
    }
    if (fused_kv) {
        // every lane finishes the newest position's k/v for its 8 columns (same addresses across the 8 key sub-lanes);
        // the 8 lanes that own that position append them to this row's cache and use them
        float kn[8], vn[8];
        pk.finish<T>(qs, kn);
        pv.finish<T>(qs, vn);
        const int t = n_keys - 1;
#pragma unroll
        for (int i = 0; i < NI; ++i) {
            if (i * 8 + ksub == t) {
                const size_t o = (((size_t)row * H + h) * kv_ld + t) * 64 + dch * 8;
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    kk[i].set(e, kn[e]); vv[i].set(e, vn[e]);
                    if (write_kv) { kbase[o + e] = from_f32<T>(kn[e]); vbase[o + e] = from_f32<T>(vn[e]); }
                }
            }
        }
    }
    float sc[NI], m = -INFINITY;
#pragma unroll
    for (int i = 0; i < NI; ++i) {
        float s = 0.f;
#pragma unroll
        for (int e = 0; e < 8; ++e) s = fmaf(qv[e] * 0.125f, kk[i].get(e), s);
        s += __shfl_xor(s, 1, 64); s += __shfl_xor(s, 2, 64); s += __shfl_xor(s, 4, 64);
        sc[i] = (i * 8 + ksub < n_keys) ? s : -INFINITY;
        m = fmaxf(m, sc[i]);
    }
    m = fmaxf(m, __shfl_xor(m, 8, 64)); m = fmaxf(m, __shfl_xor(m, 16, 64)); m = fmaxf(m, __shfl_xor(m, 32, 64));
    float l = 0.f, o[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) o[e] = 0.f;
#pragma unroll
    for (int i = 0; i < NI; ++i) {
        const float p = expf(sc[i] - m);
        l += p;
#pragma unroll
        for (int e = 0; e < 8; ++e) o[e] = fmaf(p, vv[i].get(e), o[e]);
    }
    l += __shfl_xor(l, 8, 64); l += __shfl_xor(l, 16, 64); l += __shfl_xor(l, 32, 64);
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        o[e] += __shfl_xor(o[e], 8, 64); o[e] += __shfl_xor(o[e], 16, 64); o[e] += __shfl_xor(o[e], 32, 64);
    }
    if (ksub == 0) {
        const float inv = 1.0f / l;
        store4(out_row, h * 64 + dch * 8, make_float4(o[0] * inv, o[1] * inv, o[2] * inv, o[3] * inv));
        store4(out_row, h * 64 + dch * 8 + 4, make_float4(o[4] * inv, o[5] * inv, o[6] * inv, o[7] * inv));
    }
}

// score of one key for the 8 lanes that hold its row: q . k over the lane's 8 dimensions (q already scaled), the cache's row scale,
// then the sum over the 8 lanes - every lane of the key ends with the same value.  The one place this arithmetic is written.
template <typename TKV>
__device__ __forceinline__ float decode_key_score(const float (&qv)[8], const Raw8<TKV>& kr) {
#pragma clang fp contract(off)      // as written, wherever it is inlined: callers on different paths must agree bit for bit
    float s = 0.f;
#pragma unroll
    for (int e = 0; e < 8; ++e) s = fmaf(qv[e], kr.get(e), s);
    if constexpr (Raw8<TKV>::scaled) s *= kr.scale();
    s += __shfl_xor(s, 1, 64); s += __shfl_xor(s, 2, 64); s += __shfl_xor(s, 4, 64);
    return s;
}

// Scores of the key groups g_first, g_first + g_stride, ... (8 keys each) of one (row, head) into sc_out[key]: the part of the
// online unit below that does not depend on the running softmax state, so several waves can share it (small-batch path: one
// (row, head) per workgroup, the chain of shuffles per key group is what the single wave would otherwise walk alone).
template <typename TKV>
__device__ __forceinline__ void decode_attention_scores(const void* kbase, size_t ri_base, int n_keys, const float* q_ready,
                                                        float* sc_out, int lane, int g_first, int g_stride) {
#pragma clang fp contract(off)
    const int ksub = lane >> 3, dch = lane & 7;
    float qv[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) qv[e] = q_ready[dch * 8 + e];
#pragma unroll
    for (int e = 0; e < 8; ++e) qv[e] *= 0.125f;
    const int ng = (n_keys + 7) / 8;
    for (int g0 = g_first; g0 < ng; g0 += 4 * g_stride) {
        Raw8<TKV> kr[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int key = (g0 + i * g_stride) * 8 + ksub;
            if (key < n_keys) kr[i].load_row(kbase, ri_base + key, dch); else kr[i].zero();
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int key = (g0 + i * g_stride) * 8 + ksub;
            const float s = decode_key_score<TKV>(qv, kr[i]);
            if (dch == 0 && key < n_keys) sc_out[key] = s;
        }
    }
}

// ---- long history (cross-attention over the image tokens): the history walked in chunks of 8 G keys.  A chunk's K loads and
// V loads (16 B per lane, one 128-byte key row per 8 lanes) are all issued into raw registers before any arithmetic.  Online
// softmax across chunks (fp32).  DB: two register buffers, the next chunk's loads are in flight while the current one is consumed.
// ri_base: K/V row index of key 0 of this (row, head) when there is no ancestry (kbase / vbase + row index -> Raw8::load_row);
// with ancestry the rows are ri0 + (anc[row][key] * H + h) * kv_ld + key.
// q_ready: when non-null the 8 query values of this lane (already summed over the split-K partials, bias added, NOT yet scaled)
// are read from q_ready[dch * 8 ..] instead of qs / q.  sc_ready: when non-null the scores come from sc_ready[key]
// (decode_attention_scores) and K is not read here.
template <typename T, int G, bool DB, bool NT, typename TO, typename TKV, bool SCR = false>
__device__ __forceinline__ void decode_attention_online_unit(const T* __restrict__ q, const void* kbase, const void* vbase,
                                                             const int* __restrict__ anc, int anc_ld, int kv_ld, int n_keys,
                                                             TO* out_row, int R, int H, const QSource& qs, int row, int h, int lane,
                                                             size_t ri0, size_t ri_base, const float* q_ready,
                                                             const float* sc_ready = nullptr, int prow = -1) {
#pragma clang fp contract(off)      // as written, wherever it is inlined: same bits on every path
    const int pr = prow >= 0 ? prow : row;              // row of the partial sums / `q` (see decode_attention_wave_unit)
    // SCR (compile time): the scores come from sc_ready
    constexpr int CH = 8 * G;
    const int Dh = H * 64;
    const int ksub = lane >> 3, dch = lane & 7;
    float m = -INFINITY, l = 0.f, o[8], qv[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) o[e] = 0.f;

    auto issue = [&](Raw8<TKV>(&kr)[G], Raw8<TKV>(&vr)[G], int k0) {
#pragma unroll
        for (int g = 0; g < G; ++g) {
            const int key = k0 + g * 8 + ksub;
            if (key < n_keys) {
                const size_t ri = anc ? ri0 + ((size_t)anc[(size_t)row * anc_ld + key] * H + h) * kv_ld + key : ri_base + key;
                if constexpr (NT) { kr[g].load_row_nt(kbase, ri, dch); vr[g].load_row_nt(vbase, ri, dch); }
                else { if constexpr (!SCR) kr[g].load_row(kbase, ri, dch); else kr[g].zero(); vr[g].load_row(vbase, ri, dch); }
            } else {
                kr[g].zero(); vr[g].zero();
            }
        }
    };
    auto consume = [&](const Raw8<TKV>(&kr)[G], const Raw8<TKV>(&vr)[G], int k0) {
        float sc[G], cm = -INFINITY;
#pragma unroll
        for (int g = 0; g < G; ++g) {
            const int key = k0 + g * 8 + ksub;
            float s;
            if constexpr (SCR) s = sc_ready[min(key, n_keys - 1)];
            else s = decode_key_score<TKV>(qv, kr[g]);
            sc[g] = (key < n_keys) ? s : -INFINITY;
            cm = fmaxf(cm, sc[g]);
        }
        cm = fmaxf(cm, __shfl_xor(cm, 8, 64)); cm = fmaxf(cm, __shfl_xor(cm, 16, 64)); cm = fmaxf(cm, __shfl_xor(cm, 32, 64));
        const float mn = fmaxf(m, cm);
        const float c = expf(m - mn);                   // first chunk: exp(-inf) = 0 and l, o are 0
        l *= c;
#pragma unroll
        for (int e = 0; e < 8; ++e) o[e] *= c;
#pragma unroll
        for (int g = 0; g < G; ++g) {
            const float p = expf(sc[g] - mn);
            l += p;
            const float pv = Raw8<TKV>::scaled ? p * vr[g].scale() : p;
#pragma unroll
            for (int e = 0; e < 8; ++e) o[e] = fmaf(pv, vr[g].get(e), o[e]);
        }
        m = mn;
    };
    auto get_q = [&](Part8& pq) {
        if (q_ready) {
#pragma unroll
            for (int e = 0; e < 8; ++e) qv[e] = q_ready[dch * 8 + e];
        } else if (qs.part) {
            pq.finish<T>(qs, qv);
        } else {
            load8<T>(q + (size_t)pr * Dh + h * 64 + dch * 8, qv);
        }
#pragma unroll
        for (int e = 0; e < 8; ++e) qv[e] *= 0.125f;
    };

    Part8 pq;
    if (!q_ready && qs.part) pq.issue(qs, R, pr, qs.col0 + h * 64 + dch * 8);
    Raw8<TKV> ka[G], va[G];
    issue(ka, va, 0);
    if constexpr (DB) {
        Raw8<TKV> kb[G], vb[G];
        if (CH < n_keys) issue(kb, vb, CH);
        get_q(pq);
        for (int k0 = 0;;) {
            consume(ka, va, k0);
            if (k0 + 2 * CH < n_keys) issue(ka, va, k0 + 2 * CH);
            if (k0 + CH >= n_keys) break;
            consume(kb, vb, k0 + CH);
            if (k0 + 3 * CH < n_keys) issue(kb, vb, k0 + 3 * CH);
            k0 += 2 * CH;
            if (k0 >= n_keys) break;
        }
    } else {
        get_q(pq);
        for (int k0 = 0;;) {
            consume(ka, va, k0);
            k0 += CH;
            if (k0 >= n_keys) break;
            issue(ka, va, k0);
        }
    }
    l += __shfl_xor(l, 8, 64); l += __shfl_xor(l, 16, 64); l += __shfl_xor(l, 32, 64);
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        o[e] += __shfl_xor(o[e], 8, 64); o[e] += __shfl_xor(o[e], 16, 64); o[e] += __shfl_xor(o[e], 32, 64);
    }
    if (ksub == 0) {
        const float inv = 1.0f / l;
        store4(out_row, h * 64 + dch * 8, make_float4(o[0] * inv, o[1] * inv, o[2] * inv, o[3] * inv));
        store4(out_row, h * 64 + dch * 8 + 4, make_float4(o[4] * inv, o[5] * inv, o[6] * inv, o[7] * inv));
    }
}

}  // namespace
