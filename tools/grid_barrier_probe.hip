// What does a device-wide barrier INSIDE a kernel cost on this part, next to the ~5.5 us of a dependent kernel launch?  (Round 4:
// the small-batch decode path is 76 dependent launches per step; a persistent step kernel would replace 72 of them by barriers.)
// G workgroups (one per CU at most), N barriers each followed by a read of what ANOTHER workgroup wrote before the barrier (the
// hand-over a decode phase needs, checked), three barrier variants:
//   flat   every workgroup: release fence, atomicAdd on ONE counter, spin on it, acquire fence
//   xcd    two levels: a counter per XCD (blockIdx % 8), the last arriver of an XCD bumps the global counter, all spin on that
//   nofence  flat without the fences (what the atomics alone cost; the hand-over check may then fail and is not made)
// Every spin is bounded: a workgroup that waits longer than ~50 ms sets an abort flag and leaves (the kernel always ends).
//   hipcc -O3 --offload-arch=gfx950 tools/grid_barrier_probe.hip -o /tmp/grid_barrier_probe && /tmp/grid_barrier_probe
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <vector>

struct Bar { unsigned long long count; unsigned long long xcd[8]; unsigned int abort_flag; unsigned int pad[13]; };

__device__ __forceinline__ bool spin_until(volatile unsigned long long* p, unsigned long long target, volatile unsigned int* abort_flag) {
    for (unsigned it = 0; it < 300000u; ++it) {
        if (__hip_atomic_load((unsigned long long*)p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) >= target) return true;
        if ((it & 1023u) == 1023u && *abort_flag) return false;
        __builtin_amdgcn_s_sleep(1);
    }
    *abort_flag = 1u;
    return false;
}

template <int MODE>   // 0 flat, 1 xcd, 2 nofence
__global__ __launch_bounds__(256) void barrier_loop(Bar* bar, float* data, int n, unsigned int* errors) {
    const int G = gridDim.x, b = blockIdx.x;
    __shared__ int ok;
    float mine = (float)b;
    for (int k = 0; k < n; ++k) {
        if (threadIdx.x == 0) data[(size_t)(k & 1) * G + b] = mine + (float)k;       // what the next phase reads
        __syncthreads();
        if (threadIdx.x == 0) {
            ok = 1;
            if (MODE != 2) __threadfence();
            const unsigned long long target = (unsigned long long)(k + 1) * G;
            if (MODE == 1) {
                const int x = b & 7;
                const int members = (G - x + 7) / 8;
                const unsigned long long t = __hip_atomic_fetch_add(&bar->xcd[x], 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) + 1;
                if (t == (unsigned long long)(k + 1) * members)
                    __hip_atomic_fetch_add(&bar->count, (unsigned long long)members, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            } else {
                __hip_atomic_fetch_add(&bar->count, 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
            if (!spin_until(&bar->count, target, &bar->abort_flag)) ok = 0;
            if (MODE != 2) __threadfence();
        }
        __syncthreads();
        if (!ok) return;
        if (MODE != 2 && threadIdx.x == 0) {
            const int other = (b + G / 2 + 1) % G;                                    // a workgroup on another XCD, usually
            const float v = __builtin_nontemporal_load(&data[(size_t)(k & 1) * G + other]);
            if (v != (float)other + (float)k) atomicAdd(errors, 1u);
            mine = (float)b + 0.0f * v;
        }
    }
}

template <int MODE>
static void run(const char* name, int G, int n) {
    Bar* bar; float* data; unsigned int* err;
    hipMalloc(&bar, sizeof(Bar)); hipMalloc(&data, sizeof(float) * 2 * G); hipMalloc(&err, 4);
    for (int rep = 0; rep < 3; ++rep) {
        hipMemset(bar, 0, sizeof(Bar)); hipMemset(err, 0, 4);
        hipDeviceSynchronize();
        auto t0 = std::chrono::steady_clock::now();
        hipLaunchKernelGGL(barrier_loop<MODE>, dim3(G), dim3(256), 0, 0, bar, data, n, err);
        hipDeviceSynchronize();
        auto t1 = std::chrono::steady_clock::now();
        Bar h; unsigned int e = 0;
        hipMemcpy(&h, bar, sizeof(Bar), hipMemcpyDeviceToHost); hipMemcpy(&e, err, 4, hipMemcpyDeviceToHost);
        const double us = std::chrono::duration<double, std::micro>(t1 - t0).count();
        if (rep == 2)
            printf("%-8s G=%3d  %d barriers: %8.1f us total, %6.2f us per barrier%s, hand-over errors %u\n", name, G, n, us, us / n,
                   h.abort_flag ? "  [ABORTED: a spin ran out]" : "", e);
    }
    hipFree(bar); hipFree(data); hipFree(err);
}

__global__ void empty_kernel(float* p) { if (p == nullptr && threadIdx.x == 12345) *p = 0.f; }

int main() {
    int n_cu = 0;
    hipDeviceGetAttribute(&n_cu, hipDeviceAttributeMultiprocessorCount, 0);
    printf("CUs: %d\n", n_cu);
    const int n = 2000;
    for (int G : {48, 96, 192, 256}) {
        if (G > n_cu) continue;
        run<0>("flat", G, n);
        run<1>("xcd", G, n);
        run<2>("nofence", G, n);
    }
    // the alternative: dependent launches of an empty kernel on one stream
    for (int G : {48, 256}) {
        hipDeviceSynchronize();
        auto t0 = std::chrono::steady_clock::now();
        for (int k = 0; k < n; ++k) hipLaunchKernelGGL(empty_kernel, dim3(G), dim3(256), 0, 0, (float*)nullptr + 1);
        hipDeviceSynchronize();
        auto t1 = std::chrono::steady_clock::now();
        printf("launches G=%3d  %d dependent empty kernels: %6.2f us each\n", G, n, std::chrono::duration<double, std::micro>(t1 - t0).count() / n);
    }
    return 0;
}
