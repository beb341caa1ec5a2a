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

// compute dtype tags (match include/captioner_hip.h)
enum { CAP_DT_F32 = 0, CAP_DT_BF16 = 1 };

template <typename T> struct DT;
template <> struct DT<float> { static constexpr int tag = CAP_DT_F32; static constexpr int per16B = 4; };
template <> struct DT<bf16_t> { static constexpr int tag = CAP_DT_BF16; static constexpr int per16B = 8; };

__device__ __forceinline__ float to_f32(float x) { return x; }
__device__ __forceinline__ float to_f32(bf16_t x) { return (float)x; }
template <typename T> __device__ __forceinline__ T from_f32(float x);
template <> __device__ __forceinline__ float from_f32<float>(float x) { return x; }
template <> __device__ __forceinline__ bf16_t from_f32<bf16_t>(float x) { return (bf16_t)x; }

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

// One-time per (kernel, device) setup shared by the launchers: raises the kernel's dynamic-LDS limit when `lds_bytes`
// exceeds the 64 KiB default and returns the device's CU count in *n_cu (may be null).  Thread-safe; a handle per GPU in
// one process works.  Returns 0, or -1 with cap_set_error.
int cap_kernel_setup(const void* kernel, int lds_bytes, int* n_cu);

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
