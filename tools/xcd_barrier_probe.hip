// Probe: how much does a software barrier cost among (a) the 32 workgroups of one XCD, (b) all 256 workgroups, and is data
// written with plain stores by one CU visible to L1-bypassing loads of another CU after such a barrier (same XCD / other XCD)?
//   hipcc -O2 --offload-arch=gfx950 tools/xcd_barrier_probe.hip -o /tmp/xcd_probe && /tmp/xcd_probe
// Every spin is bounded: a broken assumption gives an error count, not a hung GPU.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>

#define CHECK(x) do { hipError_t err_ = (x); if (err_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(err_)); exit(1); } } while (0)

__device__ __forceinline__ int xcc_id() { return __builtin_amdgcn_s_getreg((3 << 11) | (0 << 6) | 20) & 15; }   // HW_REG_XCC_ID

// arrive + wait on a monotonically growing counter: `target` = value after everybody of the group arrived this round
__device__ __forceinline__ int group_barrier(int* ctr, int target, int* err) {
    __syncthreads();
    int spins = 0;
    if (threadIdx.x == 0) {
        __hip_atomic_fetch_add(ctr, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        while (__hip_atomic_load(ctr, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target) {
            if (++spins > (1 << 22)) { atomicAdd(err, 1); break; }
            __builtin_amdgcn_s_sleep(1);
        }
    }
    __syncthreads();
    return spins;
}

__global__ __launch_bounds__(256) void probe(int* ctrs, float* data, long long* out, int* err, int iters, int mode, int* xcc_of) {
    __shared__ char pad[100 * 1024];                 // one workgroup per CU
    pad[threadIdx.x] = 0;
    const int bid = blockIdx.x, xcd = bid & 7, local = bid >> 3, nb = gridDim.x;
    if (threadIdx.x == 0) xcc_of[bid] = xcc_id();
    int* ctr = mode != 1 ? ctrs + xcd * 64 : ctrs + 8 * 64;
    const int gsize = mode != 1 ? nb / 8 : nb;
    // partner: next workgroup of the same XCD (mode 0) / the next workgroup id = another XCD (mode 1)
    const int partner = mode != 1 ? (((local + 1) % (nb / 8)) << 3 | xcd) : (bid + 1) % nb;
    float* mine = data + (size_t)bid * 1024;
    const float* theirs = data + (size_t)partner * 1024;
    int round = 0, bad = 0;
    group_barrier(ctr, ++round * gsize, err);
    const long long t0 = wall_clock64();
    for (int it = 0; it < iters; ++it) {
        mine[threadIdx.x * 4 + (it & 3)] = (float)(it * 1000 + bid);            // plain store
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                          // acknowledged by L2
        group_barrier(ctr, ++round * gsize, err);
        float v;
        if (mode < 2) {
            v = __hip_atomic_load(theirs + threadIdx.x * 4 + (it & 3), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        } else if (mode == 3) {      // mode 3: 16-byte buffer load with sc1 (agent scope: the L1 misses, the XCD's L2 serves it)
            typedef unsigned int u32x4_t __attribute__((ext_vector_type(4)));
            const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)theirs, 0, 4096, 0x00020000);
            const u32x4_t r = __builtin_amdgcn_raw_buffer_load_b128(rs, threadIdx.x * 16, 0, 16);
            v = __builtin_bit_cast(float, r[it & 3]);
        } else {                     // mode 2: acquire fence (L1 invalidate) once per wave, then PLAIN vector loads
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
            const float4 q = *(const float4*)(theirs + threadIdx.x * 4);
            v = (it & 3) == 0 ? q.x : (it & 3) == 1 ? q.y : (it & 3) == 2 ? q.z : q.w;
        }
        bad += v != (float)(it * 1000 + partner);
        group_barrier(ctr, ++round * gsize, err);
    }
    const long long t1 = wall_clock64();
    if (bad) atomicAdd(err + 1, bad);
    if (threadIdx.x == 0) out[bid] = t1 - t0;
}

int main() {
    int *ctrs, *err, *xcc;
    float* data;
    long long* out;
    const int nb = 256, iters = 2000;
    CHECK(hipMalloc(&ctrs, 9 * 64 * 4)); CHECK(hipMalloc(&err, 16)); CHECK(hipMalloc(&xcc, nb * 4));
    CHECK(hipMalloc(&data, (size_t)nb * 1024 * 4)); CHECK(hipMalloc(&out, nb * 8));
    CHECK(hipFuncSetAttribute((const void*)probe, hipFuncAttributeMaxDynamicSharedMemorySize, 0));
    for (int mode = 0; mode < 4; ++mode) {
        CHECK(hipMemset(ctrs, 0, 9 * 64 * 4)); CHECK(hipMemset(err, 0, 16)); CHECK(hipMemset(data, 0, (size_t)nb * 1024 * 4));
        hipLaunchKernelGGL(probe, dim3(nb), dim3(256), 0, 0, ctrs, data, out, err, iters, mode, xcc);
        CHECK(hipDeviceSynchronize());
        std::vector<long long> t(nb); std::vector<int> x(nb); int e[4];
        CHECK(hipMemcpy(t.data(), out, nb * 8, hipMemcpyDeviceToHost)); CHECK(hipMemcpy(x.data(), xcc, nb * 4, hipMemcpyDeviceToHost));
        CHECK(hipMemcpy(e, err, 16, hipMemcpyDeviceToHost));
        long long mx = 0; for (auto v : t) mx = v > mx ? v : mx;
        int mism = 0; for (int i = 0; i < nb; ++i) mism += x[i] != (i & 7);
        printf("mode %d (%s): %.3f us per barrier (2 per iteration + store/load, 100 MHz wall clock), spin timeouts %d, stale reads %d, "
               "workgroups not on XCD blockIdx%%8: %d\n", mode, mode == 0 ? "32 workgroups of one XCD, L1-bypassing loads" : mode == 1 ? "all 256 workgroups" : mode == 2 ? "32 workgroups of one XCD, acquire fence + plain loads" : "32 workgroups of one XCD, 16-byte sc1 buffer loads",
               mx / 100.0 / iters / 2, e[0], e[1], mism);
    }
    return 0;
}
