// Building blocks shared by the fused decode kernels (decode_small.hip: up to 16 rows):
// MFMA operand fragments read straight from global memory / an LDS row image, the product sequence of one K slab, and the
// split-K consumer + LayerNorm of a row in registers.  Everything is in an anonymous namespace: include from a translation unit
// that defines such kernels, after gemm_tile.h, ln.h, decode_attn.h and decode_small.h.
#pragma once

namespace {

// Pointers that arrive inside a by-value struct are generic to the compiler (flat_load: both counters, no global addressing
// modes); every one of them points to global memory, and a round trip through address space 1 tells it so.
template <typename P> __device__ __forceinline__ P* glob(P* p) {
    return (P*)(__attribute__((address_space(1))) P*)p;
}
__device__ __forceinline__ SmallLN glob_ln(SmallLN ln) {
    ln.part = glob(ln.part); ln.bias = glob(ln.bias); ln.resid = glob(ln.resid); ln.gamma = glob(ln.gamma); ln.beta = glob(ln.beta);
    ln.x_out = glob(ln.x_out);
    return ln;
}

template <typename T> struct AttT { using type = T; };
template <> struct AttT<g8_t> { using type = float; };        // split mode: q|k|v, the K/V caches are fp32

// one lane's two 16-byte pieces of a 128-byte K-slab row: G8 (hi, lo) halves of its 8 k values; bf16 k-steps 0 and 1
struct Frag { u32x4 x[2]; };
template <typename T> __device__ __forceinline__ int frag_off(int kg, int i) {
    if constexpr (is_g8<T>) return kg * 32 + i * 16;
    else return i * 64 + kg * 16;
}
template <typename T> __device__ __forceinline__ Frag load_frag(const char* row_slab, int kg) {
    Frag f;
    f.x[0] = *(const u32x4*)(row_slab + frag_off<T>(kg, 0));
    f.x[1] = *(const u32x4*)(row_slab + frag_off<T>(kg, 1));
    return f;
}
__device__ __forceinline__ Frag zero_frag() { Frag f; f.x[0] = 0u; f.x[1] = 0u; return f; }

// one K-slab of one 16 x 16 block: the product sequence every G8 / bf16 GEMM kernel of the library uses (gemm_tile.h)
template <typename T> __device__ __forceinline__ void mma_slab(f32x4& acc, const Frag& w, const Frag& a) {
    if constexpr (is_g8<T>) {
        const f16x8 wh = __builtin_bit_cast(f16x8, w.x[0]), wl = __builtin_bit_cast(f16x8, w.x[1]);
        const f16x8 ah = __builtin_bit_cast(f16x8, a.x[0]), al = __builtin_bit_cast(f16x8, a.x[1]);
        acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(wl, ah, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(wh, al, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(wh, ah, acc, 0, 0, 0);
    } else {
        acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, w.x[0]), __builtin_bit_cast(bf16x8, a.x[0]), acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, w.x[1]), __builtin_bit_cast(bf16x8, a.x[1]), acc, 0, 0, 0);
    }
}

// ---- split-K consumer + LayerNorm of up to RPW rows per wave (rows wave, wave + NWV, ...): y = sum_z part[z] + bias + resid in
// the order of reduce_layernorm_row_kernel, LayerNorm in ln_row's order; the operand-type row goes to the LDS image `img`
// (row pitch `pitch` bytes), the fp32 row to x_out when this workgroup is the designated writer.  EVERY load - the slabs and
// residual rows of all the wave's rows, bias, gamma, beta - is issued before the first add: the slabs were written by the
// previous kernel on other XCDs, a dependent round trip to them costs 2-3 us, and this way there is one.
// NV: float4 per lane and row (3 for rows up to 768 wide, 4 up to 1024).
// (Loads are unconditional 16-byte vector loads from clamped addresses under wave-uniform branches only: a per-lane
// `cond ? *p : 0` becomes a select between a global and a private address, i.e. flat loads in dword pieces.)
struct LnCols { f32x4 g, be, bb; };
template <int NV>
__device__ __forceinline__ void ln_load_cols(const SmallLN& ln, int D, int lane, LnCols (&k)[NV]) {
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        const int cc = min(lane * 4 + i * 256, D - 4);
        k[i].g = *(const f32x4*)(ln.gamma + cc);
        k[i].be = *(const f32x4*)(ln.beta + cc);
        k[i].bb = 0.f;
        if (ln.bias) k[i].bb = *(const f32x4*)(ln.bias + cc);
    }
}
struct LnRow { f32x4 pz[4], rs; };
template <int NV>
__device__ __forceinline__ void ln_load_row(const SmallLN& ln, int R, int D, int row, int lane, LnRow (&v)[NV]) {
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        const int cc = min(lane * 4 + i * 256, D - 4);
#pragma unroll
        for (int z = 0; z < 4; ++z) {
            v[i].pz[z] = 0.f;
            if (z < ln.S) v[i].pz[z] = *(const f32x4*)(ln.part + ((size_t)z * R + row) * D + cc);
        }
        v[i].rs = 0.f;
        if (ln.resid) v[i].rs = *(const f32x4*)(ln.resid + (size_t)row * D + cc);
    }
}
// y = slabs in order, + bias, + residual (reduce_layernorm_row_kernel's order), then the LayerNorm
template <typename T, int NV>
__device__ __forceinline__ void ln_finish_row(const SmallLN& ln, int R, int D, int row, int lane, const LnRow (&v)[NV],
                                              const LnCols (&k)[NV], T* out_t, float* out_f) {
    float4 a[NV], g[NV], be[NV];
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        const int cc = min(lane * 4 + i * 256, D - 4);
        f32x4 s = v[i].pz[0];
#pragma unroll
        for (int z = 1; z < 4; ++z)
            if (z < ln.S) s += v[i].pz[z];
        for (int z = 4; z < ln.S; ++z) s += *(const f32x4*)(ln.part + ((size_t)z * R + row) * D + cc);
        if (ln.bias) s += k[i].bb;
        if (ln.resid) s += v[i].rs;
        a[i] = make_float4(s[0], s[1], s[2], s[3]);
        g[i] = make_float4(k[i].g[0], k[i].g[1], k[i].g[2], k[i].g[3]);
        be[i] = make_float4(k[i].be[0], k[i].be[1], k[i].be[2], k[i].be[3]);
        if (out_f && ln.x_is_sum && lane * 4 + i * 256 < D) *(f32x4*)(out_f + lane * 4 + i * 256) = s;   // pre-LN: the stream is y
    }
    ln_row_regs<T, NV>(a, NV, lane, D, g, be, ln.eps, out_t, ln.x_is_sum ? nullptr : out_f);
}

template <typename T, int RPW, int NV>
__device__ __forceinline__ void ln_rows_prologue(const SmallLN& ln, int R, int D, char* img, int pitch, int wave, int nwv, int lane,
                                                 bool write_x) {
    constexpr int RB = RPW < 2 ? RPW : 2;                   // rows of a wave in flight at once (registers: 20 NV per row)
    LnCols k[NV];
#pragma unroll
    for (int r0 = 0; r0 < RPW; r0 += RB) {
        if (wave + r0 * nwv >= R) break;
        LnRow v[RB][NV];
#pragma unroll
        for (int rr = 0; rr < RB; ++rr) {
            const int row = wave + (r0 + rr) * nwv;
            ln_load_row<NV>(ln, R, D, min(row, R - 1), lane, v[rr]);
        }
        if (r0 == 0) ln_load_cols<NV>(ln, D, lane, k);
#pragma unroll
        for (int rr = 0; rr < RB; ++rr) {
            const int row = wave + (r0 + rr) * nwv;
            if (row >= R) break;
            ln_finish_row<T, NV>(ln, R, D, row, lane, v[rr], k, (T*)(img + (size_t)row * pitch),
                                 write_x && ln.x_out ? ln.x_out + (size_t)row * D : nullptr);
        }
    }
}

}  // namespace
