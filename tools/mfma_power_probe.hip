// What the matrix pipe sustains under the board's power limit: a register-only loop of v_mfma_f32_16x16x32_f16 (no LDS, no
// memory), 8 waves per CU on every CU, for ~6 seconds; prints the achieved TFLOP/s per second of run time (the shader clock
// settles as the DVFS reacts).  Power is read next to it with rocm-smi.
//   hipcc -O3 --offload-arch=gfx950 tools/mfma_power_probe.hip -o /tmp/mfma_power_probe && /tmp/mfma_power_probe [random]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <chrono>
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

// RANDOM: eight A and four B fragments of pseudo-random fp16 values (all mantissa / exponent bits toggling, as real operands do),
// every MFMA with another pair - against the same loop on ONE constant pair, which prices the pipe with hardly any data switching.
template <bool RANDOM>
__global__ __launch_bounds__(512) void mfma_loop(float* out, int iters) {
    f16x8 a[8], b[4];
    unsigned x = 0x9E3779B9u * (threadIdx.x + 1) + blockIdx.x;
    for (int r = 0; r < 8; ++r)
        for (int i = 0; i < 8; ++i) {
            x = x * 1664525u + 1013904223u;
            const float v = RANDOM ? ((int)(x >> 8) - (1 << 23)) * (1.0f / (1 << 23)) : 0.001f * (threadIdx.x + i);
            a[r][i] = (_Float16)v;
            if (r < 4) b[r][i] = (_Float16)(RANDOM ? 0.37f * v - 0.11f * (float)a[r][(i + 3) & 7] : 0.002f * (threadIdx.x - i));
        }
    f32x4 acc[8];
    for (int j = 0; j < 8; ++j) acc[j] = 0.f;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 4; ++u)
#pragma unroll
            for (int j = 0; j < 8; ++j)
                acc[j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a[RANDOM ? j : 0], b[RANDOM ? u : 0], acc[j], 0, 0, 0);
    }
    float s = 0.f;
    for (int j = 0; j < 8; ++j) s += acc[j][0] + acc[j][1] + acc[j][2] + acc[j][3];
    if (s == 12345.678f) out[0] = s;
}

// The same loop on v_mfma_f32_32x32x16_f16 (round 4: does the larger block - 16 MACs per operand element read from the register
// file against 8 - sustain more under the power limit?): four 32x32 accumulators, 4 x 2 random fragments, 32768 flop per MFMA.
typedef float f32x16 __attribute__((ext_vector_type(16)));
template <bool RANDOM>
__global__ __launch_bounds__(512) void mfma_loop_32(float* out, int iters) {
    f16x8 a[4], b[2];
    unsigned x = 0x9E3779B9u * (threadIdx.x + 1) + blockIdx.x;
    for (int r = 0; r < 4; ++r)
        for (int i = 0; i < 8; ++i) {
            x = x * 1664525u + 1013904223u;
            const float v = RANDOM ? ((int)(x >> 8) - (1 << 23)) * (1.0f / (1 << 23)) : 0.001f * (threadIdx.x + i);
            a[r][i] = (_Float16)v;
            if (r < 2) b[r][i] = (_Float16)(RANDOM ? 0.37f * v - 0.11f * (float)a[r][(i + 3) & 7] : 0.002f * (threadIdx.x - i));
        }
    f32x16 acc[4];
    for (int j = 0; j < 4; ++j) acc[j] = 0.f;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 4; ++u)
#pragma unroll
            for (int j = 0; j < 4; ++j)
                acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[RANDOM ? j : 0], b[RANDOM ? (u & 1) : 0], acc[j], 0, 0, 0);
    }
    float s = 0.f;
    for (int j = 0; j < 4; ++j)
        for (int e = 0; e < 16; ++e) s += acc[j][e];
    if (s == 12345.678f) out[0] = s;
}

int main(int argc, char** argv) {
    const bool rnd = argc > 1 && argv[1][0] == 'r';
    const bool big = argc > 2 && argv[2][0] == '3';                    // "32": the 32x32x16 shape (same flops per iteration)
    auto kern = big ? (rnd ? mfma_loop_32<true> : mfma_loop_32<false>) : (rnd ? mfma_loop<true> : mfma_loop<false>);
    printf("shape: %s, operands: %s\n", big ? "v_mfma_f32_32x32x16_f16" : "v_mfma_f32_16x16x32_f16", rnd ? "random fp16" : "one constant pair");
    int n_cu = 0;
    hipDeviceGetAttribute(&n_cu, hipDeviceAttributeMultiprocessorCount, 0);
    float* out; hipMalloc(&out, 4);
    const int iters = 20000;                                   // 32 MFMAs of 16x16x32 (16 of 32x32x16) per iteration per wave: same flops
    const double flop = (double)n_cu * 8 * iters * 32 * 16384.0;
    hipLaunchKernelGGL(kern, dim3(n_cu), dim3(512), 0, 0, out, iters);
    hipDeviceSynchronize();
    auto t00 = std::chrono::steady_clock::now();
    for (int rep = 0; rep < 400; ++rep) {
        auto t0 = std::chrono::steady_clock::now();
        hipLaunchKernelGGL(kern, dim3(n_cu), dim3(512), 0, 0, out, iters);
        hipDeviceSynchronize();
        auto t1 = std::chrono::steady_clock::now();
        const double dt = std::chrono::duration<double>(t1 - t0).count();
        if (rep % 25 == 0) printf("t=%5.2fs  %7.1f TFLOP/s\n", std::chrono::duration<double>(t1 - t00).count(), flop / dt / 1e12);
        if (std::chrono::duration<double>(t1 - t00).count() > 8.0) break;
    }
    return 0;
}
