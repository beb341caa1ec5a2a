// Shared device/host helpers for the gfx950 captioner kernels.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

typedef __bf16 bf16_t;
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

#define CAP_WAVE 64

typedef _Float16 f16_t;
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));

// compute dtype tags (match include/captioner_hip.h)
enum { CAP_DT_F32 = 0, CAP_DT_BF16 = 1, CAP_DT_G8 = 2,
       CAP_DT_I8W = 3 };   // storage only: row-quantised int8 weights in MFMA fragment order (gemm_skinny.hip), never a compute type

// ---- G8: the GEMM-operand layout of the split-fp16 mode (CAP_F32_SPLIT) -------------------------------------------
// An fp32 value x travels as two fp16 halves, hi = rn16(x) and lo = rn16(x - hi) (x - hi is exact in fp32), so
// hi + lo = x (1 + e), |e| <= 2^-23 inside fp16's normal range.  A GEMM forms a.w as a_hi.w_hi + a_hi.w_lo + a_lo.w_hi
// on the fp16 MFMA pipe with fp32 accumulation (the dropped a_lo.w_lo term is <= 2^-22 |a w|): fp32-grade products at
// 3/16 of the cost of the fp32 MFMA.  Storage is 4 bytes per element like fp32, rows of ld elements = 4 ld bytes, but
// inside a row every group of 8 consecutive elements is 32 bytes = [8 hi halves | 8 lo halves]: a lane's MFMA operand
// (8 consecutive k) is one 16-byte LDS read per half, with no unpacking.  Rows therefore need ld % 8 == 0.
// Weights are stored scaled by G8_WSCALE (a power of two: exact) so that the lo halves of typical weights (|w| ~ 1e-2)
// stay in fp16's normal range; the GEMM epilogue multiplies the accumulator by 1 / G8_WSCALE.  Range: a weight tensor with
// max |w| * G8_WSCALE > G8_AMAX (|w| > 15.87) is REJECTED at load (cap_load_weight names it; use CAP_F32 or CAP_BF16 for such
// a checkpoint).  Activations are unscaled; a value beyond +-G8_AMAX (fp16's range) is clamped AND counted: every G8 store
// bumps a device counter when it clamps, read through cap_g8_saturations() - the envelope inside which the
// mode is fp32-grade is |activation| <= 65000 at every GEMM input, and leaving it is never silent.
struct g8_t { unsigned int w; };                         // never dereferenced as a scalar: see store4 / g8_put
constexpr float G8_WSCALE = 4096.0f;
constexpr float G8_AMAX = 65000.0f;

// ---- KV16: the cross-attention K/V cache of the split mode.  A 64-wide head row (one token, one head) is stored as 64 int16
// and ONE fp32 scale: x ~ q * s with s = max|x| / 32767 over the row, q = rint(x / s) - 15 value bits relative to the row's
// largest element.  Rows come in groups of 32: [32 x 128 bytes of int16][32 x fp32 scale] = 4224 bytes = 33 whole cache lines,
// 132 bytes per row, every row's 128 bytes line-aligned.  The cache is written once per image by the cross-K/V GEMM's epilogue
// (gemm_pp.hip) and streamed by every decode step of every layer: it is the HBM stream of the decode side.  fp32 rows are 256
// bytes, the 24-bit rounding this format replaces (round 3, first half) 192.  Measured before the layout was built, on the
// CPU restatement with the cache values quantised in place (48 golden rows, every step): tokens identical, largest logit move
// 2.3e-5 - 24-bit rounding 1.6e-5, fp16 1.6e-4, 12-bit blocks 3.1e-4, bf16 1.7e-3; the bar is 1e-3 and the oracle's own
// summation-order noise ~1e-5.  A (layer, k | v) block of the cache is padded to whole groups (kv16_block_bytes).
struct kv16_t { short q; };                              // tag type; rows are addressed through the helpers below
constexpr int KV16_GROUP_ROWS = 32, KV16_GROUP_BYTES = 32 * 128 + 32 * 4;
__host__ __device__ __forceinline__ size_t kv16_block_bytes(size_t rows) { return (rows + 31) / 32 * (size_t)KV16_GROUP_BYTES; }
__device__ __forceinline__ size_t kv16_row_off(size_t ri) { return (ri >> 5) * KV16_GROUP_BYTES + (ri & 31) * 128; }
__device__ __forceinline__ size_t kv16_scale_off(size_t ri) { return (ri >> 5) * KV16_GROUP_BYTES + 4096 + (ri & 31) * 4; }
// the row's scale pair from its largest magnitude: s (stored) and 1 / s (applied); an all-zero row quantises to zeros
__device__ __forceinline__ void kv16_scales(float amax, float& s, float& inv) {
    s = amax * (1.0f / 32767.0f);
    inv = amax > 0.f ? 32767.0f / amax : 0.f;
}
__device__ __forceinline__ int kv16_quant(float x, float inv) { return (int)__builtin_rintf(x * inv); }   // |x| <= amax: inside int16
__device__ __forceinline__ unsigned int kv16_pack2(float a, float b, float inv) {
    return ((unsigned)kv16_quant(a, inv) & 0xFFFFu) | ((unsigned)kv16_quant(b, inv) << 16);
}

template <typename T> struct DT;
template <> struct DT<float> { static constexpr int tag = CAP_DT_F32; static constexpr int per16B = 4; };
template <> struct DT<bf16_t> { static constexpr int tag = CAP_DT_BF16; static constexpr int per16B = 8; };
template <> struct DT<g8_t> { static constexpr int tag = CAP_DT_G8; static constexpr int per16B = 4; };

__device__ __forceinline__ float to_f32(float x) { return x; }
__device__ __forceinline__ float to_f32(bf16_t x) { return (float)x; }
template <typename T> __device__ __forceinline__ T from_f32(float x);
template <> __device__ __forceinline__ float from_f32<float>(float x) { return x; }
template <> __device__ __forceinline__ bf16_t from_f32<bf16_t>(float x) { return (bf16_t)x; }

// Values the G8 stores of THIS translation unit clamped (|x| > G8_AMAX) since the last reset (groups of 4 count once).  One instance per
// .hip file (static linkage); CAP_DEFINE_G8_CLAMP_READER(name) below gives a file its host-side reader and captioner.hip sums
// them in cap_g8_saturations().
static __device__ unsigned int g_g8_clamped;
__device__ __forceinline__ void g8_note_range(float absmax) {
    if (!(absmax <= G8_AMAX)) atomicAdd(&g_g8_clamped, 1u);          // rare by construction: one compare per store otherwise
}

// ---- typed stores of GEMM operands: `row` points at element 0 of a row (a multiple of 8 elements from the buffer start
// for g8_t), c is the column.  store4: c % 4 == 0, four consecutive columns.
__device__ __forceinline__ void g8_split(float x, f16_t& hi, f16_t& lo) {
    x = __builtin_amdgcn_fmed3f(x, -G8_AMAX, G8_AMAX);
    hi = (f16_t)x;
    lo = (f16_t)(x - (float)hi);
}
__device__ __forceinline__ void store4(float* row, int c, float4 v) { *(float4*)(row + c) = v; }
__device__ __forceinline__ void store4(bf16_t* row, int c, float4 v) {
    bf16x4 w;
    w[0] = (bf16_t)v.x; w[1] = (bf16_t)v.y; w[2] = (bf16_t)v.z; w[3] = (bf16_t)v.w;
    *(bf16x4*)(row + c) = w;
}
__device__ __forceinline__ void store4(g8_t* row, int c, float4 v) {
    f16x4 hi, lo;
    const float x[4] = {v.x, v.y, v.z, v.w};
    g8_note_range(fmaxf(fmaxf(fabsf(v.x), fabsf(v.y)), fmaxf(fabsf(v.z), fabsf(v.w))));
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        f16_t h, l;
        g8_split(x[i], h, l);
        hi[i] = h; lo[i] = l;
    }
    char* g = (char*)row + (c >> 3) * 32 + (c & 7) * 2;
    *(f16x4*)g = hi;
    *(f16x4*)(g + 16) = lo;
}
__device__ __forceinline__ void store1(float* row, int c, float v) { row[c] = v; }
__device__ __forceinline__ void store1(bf16_t* row, int c, float v) { row[c] = (bf16_t)v; }
__device__ __forceinline__ void store1(g8_t* row, int c, float v) {
    f16_t hi, lo;
    g8_note_range(fabsf(v));
    g8_split(v, hi, lo);
    char* g = (char*)row + (c >> 3) * 32 + (c & 7) * 2;
    *(f16_t*)g = hi;
    *(f16_t*)(g + 16) = lo;
}
__device__ __forceinline__ float g8_get(const g8_t* row, int c) {       // tests / slow paths only
    const char* g = (const char*)row + (c >> 3) * 32 + (c & 7) * 2;
    return (float)*(const f16_t*)g + (float)*(const f16_t*)(g + 16);
}

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
    return v;
}

// exact-erf GELU (torch.nn.functional.gelu default; HF ACT2FN["gelu"])
__device__ __forceinline__ float gelu_erf(float x) { return 0.5f * x * (1.0f + erff(x * 0.70710678118654752440f)); }
// Same function with erf from Abramowitz-Stegun 7.1.26 (|abs err| <= 1.5e-7, i.e. fp32-level, far below bf16
// rounding): ~12 VALU ops + one exp instead of libm's branchy erff.  Used when the result is rounded to bf16.
__device__ __forceinline__ float gelu_erf_fast(float x) {
    const float z = fabsf(x) * 0.70710678118654752440f;
    const float t = __builtin_amdgcn_rcpf(fmaf(0.3275911f, z, 1.0f));   // v_rcp_f32 (1 ulp), not the IEEE divide sequence
    float p = fmaf(1.061405429f, t, -1.453152027f);
    p = fmaf(p, t, 1.421413741f);
    p = fmaf(p, t, -0.284496736f);
    p = fmaf(p, t, 0.254829592f);
    const float e = 1.0f - p * t * __expf(-z * z);      // erf(|x|/sqrt2)
    return 0.5f * x * (1.0f + copysignf(e, x));
}
template <typename T> __device__ __forceinline__ float gelu_for(float x);
template <> __device__ __forceinline__ float gelu_for<float>(float x) { return gelu_erf(x); }
template <> __device__ __forceinline__ float gelu_for<__bf16>(float x) { return gelu_erf_fast(x); }
// split mode: the A&S 7.1.26 form (|erf error| <= 1.5e-7, i.e. a few fp32 ulps - the same order as the split product's own
// 2^-21) instead of libm's branchy erff: the fc1 epilogue covers 155 M elements per layer at batch 256 and measured 1.4 ms
// per batch slower with erff; every golden stays token-identical (tests/test_parity_gpu.py)
template <> __device__ __forceinline__ float gelu_for<g8_t>(float x) { return gelu_erf_fast(x); }

// One-time per (kernel, device) setup shared by the launchers: raises the kernel's dynamic-LDS limit when `lds_bytes`
// exceeds the 64 KiB default and returns the device's CU count in *n_cu (may be null).  Thread-safe; a handle per GPU in
// one process works.  Returns 0, or -1 with cap_set_error.
int cap_kernel_setup(const void* kernel, int lds_bytes, int* n_cu);

// Host-side reader of a translation unit's g_g8_clamped: adds it to *total and optionally clears it (current device).
#define CAP_DEFINE_G8_CLAMP_READER(name)                                                                              \
    int name(unsigned long long* total, int reset) {                                                                  \
        unsigned int v = 0, z = 0;                                                                                    \
        CAP_HIP_CHECK(hipMemcpyFromSymbol(&v, HIP_SYMBOL(g_g8_clamped), sizeof(v)));                                  \
        *total += v;                                                                                                  \
        if (reset) CAP_HIP_CHECK(hipMemcpyToSymbol(HIP_SYMBOL(g_g8_clamped), &z, sizeof(z)));                         \
        return 0;                                                                                                     \
    }

// Host-side error plumbing (captioner.cpp owns the storage).
void cap_set_error(const char* fmt, ...);
#define CAP_HIP_CHECK(expr)                                                                   \
    do {                                                                                      \
        hipError_t _e = (expr);                                                               \
        if (_e != hipSuccess) {                                                               \
            cap_set_error("%s failed: %s (%s:%d)", #expr, hipGetErrorString(_e), __FILE__, __LINE__); \
            return -1;                                                                        \
        }                                                                                     \
    } while (0)
