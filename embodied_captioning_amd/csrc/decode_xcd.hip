// Persistent decode-step kernel of the BLIP text decoder: ALL layers of one decode step in ONE launch.
//
// Why: a decode step is a chain of ~140 short dependent kernels, and a dependent launch costs ~5.4 us on this system however
// small the kernel is - 19 steps x 140 launches put ~14 ms under every generate.  A software barrier
// across all 256 workgroups is no cheaper (3.9 us, and the eight L2s are not coherent with each other: tools/
// xcd_barrier_probe.hip), but a barrier among the 32 workgroups of ONE XCD costs 0.9 us and needs no cache maintenance: they
// share an L2, stores are written through to it, and a load with the sc1 bit (agent scope) misses the L1 and is served from it.
//
// So the rows of a step are partitioned by XCD: XCD x owns rows [x * RX, (x + 1) * RX) and runs every phase of every layer on
// them with its 32 workgroups, separated by XCD-local barriers; nothing is exchanged between XCDs inside the launch.  Which
// XCD a workgroup is on is READ from the hardware (HW_REG_XCC_ID), not inferred from blockIdx: alone on the GPU the
// dispatcher places workgroup i on XCD i % 8, but not when kernels of other streams are resident (measured: the stream
// pool).  The grid is one workgroup per CU and a CU holds at most one (98 KiB of LDS), so every XCD ends up with exactly
// n_cu / 8 of them whatever the placement order; a workgroup's index inside its XCD is its arrival order there.
// Every XCD streams all decoder weights itself (8 x the weight traffic, mostly served by the memory-side Infinity Cache:
// all XCDs read the same weights at about the same time) - the price for cutting ~140 launches to one.
//
// Phases of a layer (a "unit" is what one workgroup or wave works on):
//   GEMM       out[rows, N] = A[rows, K] . W[N, K]^T: a workgroup takes a contiguous range of 16-column units (N / 16 units
//              over the XCD's workgroups, <= 6 each), its 8 waves split K (wave w: k-steps w, w + 8, ...), MFMA operands go
//              straight from global memory to registers in fragment layout (a lane's 8 consecutive k = 16 / 32 bytes: rows
//              of W are streamed exactly once per XCD, coalesced), the 8 partial tiles are summed through LDS in wave order.
//   attention  one wave per (row, head): self-attention over the cached positions (+ the step's own k / v from the q|k|v
//              buffer, appended to the cache here), cross-attention over the image's beam-shared K / V (HBM-bound).
//   LayerNorm  one workgroup per row: y = x + branch, LayerNorm -> residual stream (fp32) and the next GEMM's operand.
// Activations that cross workgroups inside the launch are read with sc1 loads (ld_l2); weights, caches and everything written
// by earlier launches are read normally.
//
// Deadlock freedom: the barriers need every workgroup of the launch resident (grid = one per CU, 98 KiB of LDS each keeps it
// at one per CU).  Other kernels may hold CUs - they finish without waiting for us.  Two of THESE kernels at once (two
// streams) could each hold part of the GPU and wait for the rest forever, so the host serialises them on the GPU with an
// event chain (captioner.hip, xcd_launch_guard).  Every spin is bounded: a broken assumption sets an error word and gives a
// wrong answer that the host reports, never a hung GPU.
#include "decode_xcd.h"

namespace {

constexpr int NWAVE = 8, NTHREAD = NWAVE * 64, UMAX = 6, MBMAX = 2;

__device__ __forceinline__ int xcc_id() { return __builtin_amdgcn_s_getreg((3 << 11) | (0 << 6) | 20) & 15; }   // HW_REG_XCC_ID

// 8 / 16 bytes through the XCD's L2, not this CU's L1 (agent-scope atomic loads compile to global_load_dwordx2 sc1)
__device__ __forceinline__ unsigned long long ld8_l2(const void* p) {
    return __hip_atomic_load((const unsigned long long*)p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
struct B16 { unsigned long long a, b; };
__device__ __forceinline__ B16 ld16_l2(const void* p) { B16 r; r.a = ld8_l2(p); r.b = ld8_l2((const char*)p + 8); return r; }
__device__ __forceinline__ float4 ldf4_l2(const float* p) { return __builtin_bit_cast(float4, ld16_l2(p)); }

struct Ctx {
    int xcd, local, nl, tid, lane, wave;
    int* ctr;                 // this XCD's barrier counter
    unsigned target;          // value the counter reaches when everybody has arrived at the next barrier
    int* err;
    char* lds;
    int broken;               // (thread 0) a barrier of this workgroup timed out: the launch is lost, stop waiting at the others
    long long* dbg; int ndbg;
};

// XCD-local barrier.  Every thread first waits for its own stores to be acknowledged by the L2.
__device__ __forceinline__ void xcd_barrier(Ctx& c) {
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    __syncthreads();
    c.target += (unsigned)c.nl;
    if (c.tid == 0) {
        __hip_atomic_fetch_add(c.ctr, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        int spins = 0;
        while (!c.broken && (int)((unsigned)__hip_atomic_load(c.ctr, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) - c.target) < 0) {
            if (++spins > (1 << 18)) { atomicOr(c.err, 1); c.broken = 1; break; }      // ~30 ms: a healthy barrier takes ~1 us
            __builtin_amdgcn_s_sleep(1);
        }
        if (c.dbg) c.dbg[c.ndbg++] = wall_clock64();
    }
    __syncthreads();
}

// ---------------------------------------------------------------------------------------------------------------- GEMM
template <typename T> struct Frag;
template <> struct Frag<g8_t> {            // 32 k per step: hi chunk 2 kg, lo chunk 2 kg + 1 of a 128-byte row segment
    f16x8 hi, lo;
    static constexpr int STEP_BYTES = 128, REGS = 8;     // VGPRs per fragment
    __device__ __forceinline__ void load_w(const char* row_step, int kg) {
        hi = *(const f16x8*)(row_step + kg * 32); lo = *(const f16x8*)(row_step + kg * 32 + 16);
    }
    __device__ __forceinline__ void load_a(const char* row_step, int kg) {
        hi = __builtin_bit_cast(f16x8, ld16_l2(row_step + kg * 32)); lo = __builtin_bit_cast(f16x8, ld16_l2(row_step + kg * 32 + 16));
    }
    // acc += w . a  (W as the MFMA A operand: a lane's 4 accumulators are 4 consecutive output columns), order of common.h
    __device__ static __forceinline__ void mma(f32x4& acc, const Frag& w, const Frag& a) {
        acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(w.lo, a.hi, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(w.hi, a.lo, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(w.hi, a.hi, acc, 0, 0, 0);
    }
    static constexpr float OSCALE = 1.0f / G8_WSCALE;
};
template <> struct Frag<bf16_t> {          // 32 k per step = 64 bytes per row: chunk kg
    bf16x8 v;
    static constexpr int STEP_BYTES = 64, REGS = 4;
    __device__ __forceinline__ void load_w(const char* row_step, int kg) { v = *(const bf16x8*)(row_step + kg * 16); }
    __device__ __forceinline__ void load_a(const char* row_step, int kg) { v = __builtin_bit_cast(bf16x8, ld16_l2(row_step + kg * 16)); }
    __device__ static __forceinline__ void mma(f32x4& acc, const Frag& w, const Frag& a) {
        acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w.v, a.v, acc, 0, 0, 0);
    }
    static constexpr float OSCALE = 1.0f;
};

// One pass of a workgroup over NU 16-column units x 32 rows: wave w takes k-steps w, w + 8, ...; CH of them are in flight at a
// time (every load of a chunk is issued before its first MFMA: one memory round trip per chunk).  (NU, CH) is picked by the
// caller so that CH * (NU + 2) fragments fit the register file: narrow outputs (N = 768: 1-2 units per workgroup) run deep.
template <typename T, int NU, int CH>
__device__ __forceinline__ void gemm_units(Ctx& c, const char* const (&wrow)[UMAX], const char* const (&arow)[MBMAX], int nu, int nk,
                                           f32x4* red) {
    using F = Frag<T>;
    const int kg = c.lane >> 4;
    f32x4 acc[NU][MBMAX];
#pragma unroll
    for (int u = 0; u < NU; ++u)
#pragma unroll
        for (int mb = 0; mb < MBMAX; ++mb) acc[u][mb] = 0.f;
    for (int s0 = c.wave; s0 < nk; s0 += NWAVE * CH) {
        F wf[CH][NU], af[CH][MBMAX];
#pragma unroll
        for (int ch = 0; ch < CH; ++ch) {
            const int s = min(s0 + ch * NWAVE, nk - 1);                  // past the end: re-read a valid step, result unused
#pragma unroll
            for (int u = 0; u < NU; ++u) wf[ch][u].load_w(wrow[u] + (size_t)s * F::STEP_BYTES, kg);
#pragma unroll
            for (int mb = 0; mb < MBMAX; ++mb) af[ch][mb].load_a(arow[mb] + (size_t)s * F::STEP_BYTES, kg);
        }
#pragma unroll
        for (int ch = 0; ch < CH; ++ch) {
            if (s0 + ch * NWAVE < nk) {
#pragma unroll
                for (int u = 0; u < NU; ++u)
#pragma unroll
                    for (int mb = 0; mb < MBMAX; ++mb) F::mma(acc[u][mb], wf[ch][u], af[ch][mb]);
            }
        }
    }
    // K-split partial tiles -> LDS (units beyond nu are duplicates of the last one: not stored)
#pragma unroll
    for (int u = 0; u < NU; ++u)
#pragma unroll
        for (int mb = 0; mb < MBMAX; ++mb)
            if (u < nu) red[((c.wave * UMAX + u) * MBMAX + mb) * 64 + c.lane] = acc[u][mb];
}

// out[r0 + i][n] = act(sum_k A[r0 + i][k] W[n][k] * OSCALE + bias[n]) for i < rows, n < N.  A: T rows of K elements (produced
// by other workgroups in this launch), W: T [N, K]; out_f (fp32, ld N) or out_t (T, ld N).  act: 0 none, 1 exact-erf GELU.
template <typename T>
__device__ void gemm_phase(Ctx& c, const T* __restrict__ A, const T* __restrict__ W, const float* __restrict__ bias,
                           float* out_f, T* out_t, int r0, int rows, int N, int K, int act) {
    using F = Frag<T>;
    const int U = N >> 4;                                   // 16-column units
    const int u_beg = (int)((long)c.local * U / c.nl), u_end = (int)((long)(c.local + 1) * U / c.nl);
    const int r16 = c.lane & 15;
    const int nk = K >> 5;
    const size_t row_bytes = (size_t)K * sizeof(T);
    f32x4* red = (f32x4*)c.lds;                             // [NWAVE][UMAX][MBMAX][64]
    constexpr int BUDGET = (F::REGS == 8 ? 168 : 128) / F::REGS;      // fragments in flight per lane
    for (int rg = 0; rg < rows; rg += 16 * MBMAX) {
        for (int ub = u_beg; ub < u_end; ub += UMAX) {
            const int nu = min(UMAX, u_end - ub);
            const char* wrow[UMAX];
            const char* arow[MBMAX];
#pragma unroll
            for (int u = 0; u < UMAX; ++u) wrow[u] = (const char*)W + (size_t)((ub + min(u, nu - 1)) * 16 + r16) * row_bytes;
#pragma unroll
            for (int mb = 0; mb < MBMAX; ++mb) arow[mb] = (const char*)A + (size_t)(r0 + min(rg + mb * 16 + r16, rows - 1)) * row_bytes;
            if (nu <= 2) gemm_units<T, 2, BUDGET / 4>(c, wrow, arow, nu, nk, red);
            else if (nu <= 4) gemm_units<T, 4, BUDGET / 6>(c, wrow, arow, nu, nk, red);
            else gemm_units<T, 6, BUDGET / 8>(c, wrow, arow, nu, nk, red);
            if (c.dbg && c.tid == 0 && c.ndbg < 4000) c.dbg[2048 + c.ndbg * 2] = wall_clock64();      // loads + MFMAs done (wave 0)
            __syncthreads();
            // the 8 waves' partial tiles are summed in wave order (deterministic)
            for (int it = c.tid; it < nu * MBMAX * 64; it += NTHREAD) {
                const int l = it & 63, mb = (it >> 6) % MBMAX, u = it / (64 * MBMAX);
                f32x4 v = red[((0 * UMAX + u) * MBMAX + mb) * 64 + l];
#pragma unroll
                for (int w = 1; w < NWAVE; ++w) v += red[((w * UMAX + u) * MBMAX + mb) * 64 + l];
                const int row = rg + mb * 16 + (l & 15), col = (ub + u) * 16 + 4 * (l >> 4);
                if (row < rows) {
                    v *= F::OSCALE;
                    if (bias) v += *(const f32x4*)(bias + col);
                    if (act == 1) {
#pragma unroll
                        for (int e = 0; e < 4; ++e) v[e] = gelu_for<T>(v[e]);
                    }
                    if (out_f) *(f32x4*)(out_f + (size_t)(r0 + row) * N + col) = v;
                    else store4(out_t + (size_t)(r0 + row) * N, col, make_float4(v[0], v[1], v[2], v[3]));
                }
            }
            if (c.dbg && c.tid == 0 && c.ndbg < 4000) { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); c.dbg[2048 + c.ndbg * 2 + 1] = wall_clock64(); }
            __syncthreads();
        }
    }
}

// ----------------------------------------------------------------------------------------------------------- LayerNorm
// y = (branch ? x + branch : x) for one row per workgroup; LayerNorm(y) -> x_out (fp32) and xt_out (T).  Rows wider than
// 4 * NTHREAD do not occur on this path (checked by the launcher).  The statistics are two-pass in fp32 like ln_row.
template <typename T>
__device__ void ln_rows(Ctx& c, const float* x_in, const float* branch, const float* __restrict__ g, const float* __restrict__ b,
                        float eps, float* x_out, T* xt_out, int r0, int rows, int D, bool x_in_shared) {
    float* sp = (float*)c.lds;                              // [2][NWAVE]
    for (int r = c.local; r < rows; r += c.nl) {
        const int row = r0 + r, col = c.tid * 4;
        const bool on = col < D;
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (on) {
            v = x_in_shared ? ldf4_l2(x_in + (size_t)row * D + col) : *(const float4*)(x_in + (size_t)row * D + col);
            if (branch) { const float4 d = ldf4_l2(branch + (size_t)row * D + col); v.x += d.x; v.y += d.y; v.z += d.z; v.w += d.w; }
        }
        float s = wave_sum(on ? (v.x + v.y) + (v.z + v.w) : 0.f);
        if (c.lane == 0) sp[c.wave] = s;
        __syncthreads();
        float tot = 0.f;
#pragma unroll
        for (int w = 0; w < NWAVE; ++w) tot += sp[w];
        const float mean = tot / (float)D;
        const float dx = v.x - mean, dy = v.y - mean, dz = v.z - mean, dw = v.w - mean;
        float q = wave_sum(on ? (dx * dx + dy * dy) + (dz * dz + dw * dw) : 0.f);
        if (c.lane == 0) sp[NWAVE + c.wave] = q;
        __syncthreads();
        float qt = 0.f;
#pragma unroll
        for (int w = 0; w < NWAVE; ++w) qt += sp[NWAVE + w];
        const float rstd = 1.0f / sqrtf(qt / (float)D + eps);
        if (on) {
            const float4 gg = *(const float4*)(g + col), bb = *(const float4*)(b + col);
            float4 o;
            o.x = dx * rstd * gg.x + bb.x; o.y = dy * rstd * gg.y + bb.y; o.z = dz * rstd * gg.z + bb.z; o.w = dw * rstd * gg.w + bb.w;
            *(float4*)(x_out + (size_t)row * D + col) = o;
            store4(xt_out + (size_t)row * D, col, o);
        }
        __syncthreads();
    }
}

// ----------------------------------------------------------------------------------------------------------- attention
template <typename TC> struct Row8;           // 8 consecutive head dims of one cached key / value row
template <> struct Row8<float> {
    f32x4 a, b;
    __device__ __forceinline__ void load(const float* p) { a = *(const f32x4*)p; b = *(const f32x4*)(p + 4); }
    __device__ __forceinline__ void zero() { a = 0.f; b = 0.f; }
    __device__ __forceinline__ float get(int i) const { return i < 4 ? a[i] : b[i - 4]; }
    __device__ static __forceinline__ void store(float* p, const float (&v)[8]) {
        *(f32x4*)p = f32x4{v[0], v[1], v[2], v[3]}; *(f32x4*)(p + 4) = f32x4{v[4], v[5], v[6], v[7]};
    }
    __device__ static __forceinline__ float round(float x) { return x; }
};
template <> struct Row8<bf16_t> {
    bf16x8 r;
    __device__ __forceinline__ void load(const bf16_t* p) { r = *(const bf16x8*)p; }
    __device__ __forceinline__ void zero() { for (int i = 0; i < 8; ++i) r[i] = (bf16_t)0.f; }
    __device__ __forceinline__ float get(int i) const { return (float)r[i]; }
    __device__ static __forceinline__ void store(bf16_t* p, const float (&v)[8]) {
        bf16x8 w;
        for (int i = 0; i < 8; ++i) w[i] = (bf16_t)v[i];
        *(bf16x8*)p = w;
    }
    __device__ static __forceinline__ float round(float x) { return (float)(bf16_t)x; }
};

// One wave = one (row, head): 8 lanes x 8 dims cover a 64-wide key row, 8 keys per pass (ksub = lane >> 3).  Keys
// [0, n_cached) come from the cache (row `src(j)` of it), then optionally one more key / value held in registers (the
// step's own: self-attention).  Online softmax in fp32 over chunks of 8 G keys.  Result (all lanes): out[8] for dims
// dch * 8 .. + 7, already normalised.
template <typename TC, int G>
__device__ __forceinline__ void attend(const float (&qv)[8], const TC* __restrict__ kb, const TC* __restrict__ vb, size_t head_off,
                                       size_t row_stride, int kv_ld, const int* __restrict__ anc_row, int fixed_src, int n_cached,
                                       bool extra, const float (&kn)[8], const float (&vn)[8], int lane, float (&out)[8]) {
    const int ksub = lane >> 3, dch = lane & 7;
    float m = -INFINITY, l = 0.f, o[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) o[e] = 0.f;
    for (int k0 = 0; k0 < n_cached; k0 += 8 * G) {
        Row8<TC> kr[G], vr[G];
#pragma unroll
        for (int g = 0; g < G; ++g) {
            const int key = k0 + g * 8 + ksub;
            if (key < n_cached) {
                const int src = anc_row ? anc_row[key] : fixed_src;
                const size_t off = (size_t)src * row_stride + head_off + ((size_t)key * 64 + dch * 8);
                kr[g].load(kb + off); vr[g].load(vb + off);
            } else { kr[g].zero(); vr[g].zero(); }
        }
        float sc[G], cm = -INFINITY;
#pragma unroll
        for (int g = 0; g < G; ++g) {
            float s = 0.f;
#pragma unroll
            for (int e = 0; e < 8; ++e) s = fmaf(qv[e], kr[g].get(e), s);
            s += __shfl_xor(s, 1, 64); s += __shfl_xor(s, 2, 64); s += __shfl_xor(s, 4, 64);
            sc[g] = (k0 + g * 8 + ksub < n_cached) ? s : -INFINITY;
            cm = fmaxf(cm, sc[g]);
        }
        cm = fmaxf(cm, __shfl_xor(cm, 8, 64)); cm = fmaxf(cm, __shfl_xor(cm, 16, 64)); cm = fmaxf(cm, __shfl_xor(cm, 32, 64));
        const float mn = fmaxf(m, cm);
        const float cf = expf(m - mn);
        l *= cf;
#pragma unroll
        for (int e = 0; e < 8; ++e) o[e] *= cf;
#pragma unroll
        for (int g = 0; g < G; ++g) {
            const float p = expf(sc[g] - mn);
            l += p;
#pragma unroll
            for (int e = 0; e < 8; ++e) o[e] = fmaf(p, vr[g].get(e), o[e]);
        }
        m = mn;
    }
    // fold the 8 key sub-lanes: (l, o) of every sub-lane are relative to the same running maximum m
    l += __shfl_xor(l, 8, 64); l += __shfl_xor(l, 16, 64); l += __shfl_xor(l, 32, 64);
#pragma unroll
    for (int e = 0; e < 8; ++e) { o[e] += __shfl_xor(o[e], 8, 64); o[e] += __shfl_xor(o[e], 16, 64); o[e] += __shfl_xor(o[e], 32, 64); }
    if (extra) {                                            // the step's own key / value (registers, every sub-lane holds them)
        float s = 0.f;
#pragma unroll
        for (int e = 0; e < 8; ++e) s = fmaf(qv[e], kn[e], s);
        s += __shfl_xor(s, 1, 64); s += __shfl_xor(s, 2, 64); s += __shfl_xor(s, 4, 64);
        const float mn = fmaxf(m, s);
        const float cf = expf(m - mn), p = expf(s - mn);     // n_cached == 0: m = -inf, cf = 0, l = o = 0
        l = l * cf + p;
#pragma unroll
        for (int e = 0; e < 8; ++e) o[e] = fmaf(p, vn[e], o[e] * cf);
    }
    const float inv = 1.0f / l;
#pragma unroll
    for (int e = 0; e < 8; ++e) out[e] = o[e] * inv;
}

__device__ __forceinline__ void ld8f_l2(const float* p, float (&v)[8]) {
    const float4 a = ldf4_l2(p), b = ldf4_l2(p + 4);
    v[0] = a.x; v[1] = a.y; v[2] = a.z; v[3] = a.w; v[4] = b.x; v[5] = b.y; v[6] = b.z; v[7] = b.w;
}

// self-attention over the cache + this step's k / v (appended to the cache here).  qkv fp32 [rows, 3 Dh] (this launch).
template <typename T, typename TC>
__device__ void self_attention_phase(Ctx& c, const XParams& p, const float* qkv, TC* kc, TC* vc, T* ctx, int r0, int rows) {
    const int H = p.H, Dh = H * 64, dch = c.lane & 7, ksub = c.lane >> 3;
    const size_t row_stride = (size_t)H * p.Lm * 64;
    const int nunits = rows * H, nw = c.nl * NWAVE;
    for (int u = c.local * NWAVE + c.wave; u < nunits; u += nw) {
        const int r = u / H, h = u - r * H, row = r0 + r;
        if (p.skip && p.skip[row]) continue;
        float qv[8], kn[8], vn[8], out[8];
        const float* base = qkv + (size_t)row * 3 * Dh + h * 64 + dch * 8;
        ld8f_l2(base, qv); ld8f_l2(base + Dh, kn); ld8f_l2(base + 2 * Dh, vn);
#pragma unroll
        for (int e = 0; e < 8; ++e) { qv[e] = Row8<TC>::round(qv[e]) * 0.125f; kn[e] = Row8<TC>::round(kn[e]); vn[e] = Row8<TC>::round(vn[e]); }
        if (ksub == 0) {                                    // position t of this row's own cache row
            const size_t o = (size_t)row * row_stride + (size_t)h * p.Lm * 64 + ((size_t)p.t * 64 + dch * 8);
            Row8<TC>::store(kc + o, kn); Row8<TC>::store(vc + o, vn);
        }
        attend<TC, 4>(qv, kc, vc, (size_t)h * p.Lm * 64, row_stride, p.Lm, p.anc ? p.anc + (size_t)row * p.anc_ld : nullptr, row, p.t, true,
                      kn, vn, c.lane, out);
        if (ksub == 0) {
            T* op = ctx + (size_t)row * Dh;
            store4(op, h * 64 + dch * 8, make_float4(out[0], out[1], out[2], out[3]));
            store4(op, h * 64 + dch * 8 + 4, make_float4(out[4], out[5], out[6], out[7]));
        }
    }
}

// cross-attention over the image's NT tokens (beam-shared K / V: rows r / K of an image read the same block).  q fp32 [rows, Dh].
template <typename T, typename TC>
__device__ void cross_attention_phase(Ctx& c, const XParams& p, const float* q, const TC* ck, const TC* cv, T* ctx, int r0, int rows) {
    const int H = p.H, Dh = H * 64, dch = c.lane & 7, ksub = c.lane >> 3;
    const size_t row_stride = (size_t)H * p.NT * 64;
    const int nunits = rows * H, nw = c.nl * NWAVE;
    for (int u = c.local * NWAVE + c.wave; u < nunits; u += nw) {
        const int r = u / H, h = u - r * H, row = r0 + r;
        if (p.skip && p.skip[row]) continue;
        float qv[8], out[8], dummy[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
        ld8f_l2(q + (size_t)row * Dh + h * 64 + dch * 8, qv);
#pragma unroll
        for (int e = 0; e < 8; ++e) qv[e] = Row8<TC>::round(qv[e]) * 0.125f;
        attend<TC, 7>(qv, ck, cv, (size_t)h * p.NT * 64, row_stride, p.NT, nullptr, row / p.K, p.NT, false, dummy, dummy, c.lane, out);
        if (ksub == 0) {
            T* op = ctx + (size_t)row * Dh;
            store4(op, h * 64 + dch * 8, make_float4(out[0], out[1], out[2], out[3]));
            store4(op, h * 64 + dch * 8 + 4, make_float4(out[4], out[5], out[6], out[7]));
        }
    }
}

// ------------------------------------------------------------------------------------------------------------- kernel
template <typename T, typename TC>
__global__ __launch_bounds__(NTHREAD, 1) void decode_step_xcd_kernel(XParams p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    Ctx c;
    c.tid = threadIdx.x; c.lane = c.tid & 63; c.wave = __builtin_amdgcn_readfirstlane(c.tid >> 6);
    c.nl = gridDim.x >> 3;
    // where am I?  XCD from the hardware register, index inside the XCD = arrival order (registration counter)
    __shared__ int where[2];
    if (c.tid == 0) {
        const int x = xcc_id() & 7;
        const unsigned slot = (unsigned)__hip_atomic_fetch_add(p.reg + x * 64, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) - p.reg_base[x];
        where[0] = x; where[1] = (int)slot;
        if (slot >= (unsigned)c.nl) atomicOr(p.err, 2);               // more workgroups on this XCD than CUs: not one per CU
    }
    __syncthreads();
    c.xcd = where[0]; c.local = where[1];
    if (c.local >= c.nl) return;
    c.ctr = p.bar + c.xcd * 64; c.target = p.bar_base[c.xcd]; c.err = p.err; c.lds = smem; c.broken = 0;
    c.dbg = (p.dbg && c.xcd == 0 && c.local == 0) ? p.dbg : nullptr; c.ndbg = 0;
    if (c.dbg && c.tid == 0) c.dbg[c.ndbg++] = wall_clock64();
    const int RX = (p.R + 7) / 8, r0 = c.xcd * RX, rows = min(RX, p.R - r0);
    if (rows <= 0) return;                                             // an XCD without rows: none of its workgroups waits
    const int Tw = p.T, Fw = p.F;
    T* xt = (T*)p.xt; T* ctx = (T*)p.ctx; T* hbuf = (T*)p.h;

    // embeddings: x = LayerNorm(word[token] + position[t])   (HF BlipTextEmbeddings)
    {
        float* sp = (float*)c.lds;
        for (int r = c.local; r < rows; r += c.nl) {
            const int row = r0 + r, col = c.tid * 4;
            const bool on = col < Tw;
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            if (on) {
                const int tok = p.tokens[(size_t)row * p.tok_ld + p.t];
                const float4 a = *(const float4*)(p.word + (size_t)tok * Tw + col), b = *(const float4*)(p.pos + (size_t)p.t * Tw + col);
                v.x = a.x + b.x; v.y = a.y + b.y; v.z = a.z + b.z; v.w = a.w + b.w;
            }
            float s = wave_sum(on ? (v.x + v.y) + (v.z + v.w) : 0.f);
            if (c.lane == 0) sp[c.wave] = s;
            __syncthreads();
            float tot = 0.f;
            for (int w = 0; w < NWAVE; ++w) tot += sp[w];
            const float mean = tot / (float)Tw;
            const float dx = v.x - mean, dy = v.y - mean, dz = v.z - mean, dw = v.w - mean;
            float qq = wave_sum(on ? (dx * dx + dy * dy) + (dz * dz + dw * dw) : 0.f);
            if (c.lane == 0) sp[NWAVE + c.wave] = qq;
            __syncthreads();
            float qt = 0.f;
            for (int w = 0; w < NWAVE; ++w) qt += sp[NWAVE + w];
            const float rstd = 1.0f / sqrtf(qt / (float)Tw + p.eps);
            if (on) {
                const float4 gg = *(const float4*)(p.emb_g + col), bb = *(const float4*)(p.emb_b + col);
                float4 o;
                o.x = dx * rstd * gg.x + bb.x; o.y = dy * rstd * gg.y + bb.y; o.z = dz * rstd * gg.z + bb.z; o.w = dw * rstd * gg.w + bb.w;
                *(float4*)(p.x + (size_t)row * Tw + col) = o;
                store4(xt + (size_t)row * Tw, col, o);
            }
            __syncthreads();
        }
    }
    xcd_barrier(c);
    for (int li = 0; li < p.n_layers; ++li) {
        const XLayer& L = p.layers[li];
        // self-attention block (post-LN): x = LN(x + dense(attn(q, k, v)))
        gemm_phase<T>(c, xt, (const T*)L.w_qkv, L.b_qkv, p.qkv, (T*)nullptr, r0, rows, 3 * Tw, Tw, 0);
        xcd_barrier(c);
        self_attention_phase<T, TC>(c, p, p.qkv, (TC*)L.kc, (TC*)L.vc, ctx, r0, rows);
        xcd_barrier(c);
        gemm_phase<T>(c, ctx, (const T*)L.w_so, L.b_so, p.tmp, (T*)nullptr, r0, rows, Tw, Tw, 0);
        xcd_barrier(c);
        ln_rows<T>(c, p.x, p.tmp, L.so_g, L.so_b, p.eps, p.x, xt, r0, rows, Tw, true);
        xcd_barrier(c);
        // cross-attention block
        gemm_phase<T>(c, xt, (const T*)L.w_cq, L.b_cq, p.q, (T*)nullptr, r0, rows, Tw, Tw, 0);
        xcd_barrier(c);
        cross_attention_phase<T, TC>(c, p, p.q, (const TC*)L.ck, (const TC*)L.cv, ctx, r0, rows);
        xcd_barrier(c);
        gemm_phase<T>(c, ctx, (const T*)L.w_co, L.b_co, p.tmp, (T*)nullptr, r0, rows, Tw, Tw, 0);
        xcd_barrier(c);
        ln_rows<T>(c, p.x, p.tmp, L.co_g, L.co_b, p.eps, p.x, xt, r0, rows, Tw, true);
        xcd_barrier(c);
        // feed-forward block
        gemm_phase<T>(c, xt, (const T*)L.w_f1, L.b_f1, (float*)nullptr, hbuf, r0, rows, Fw, Tw, 1);
        xcd_barrier(c);
        gemm_phase<T>(c, hbuf, (const T*)L.w_f2, L.b_f2, p.tmp, (T*)nullptr, r0, rows, Tw, Fw, 0);
        xcd_barrier(c);
        ln_rows<T>(c, p.x, p.tmp, L.f_g, L.f_b, p.eps, p.x, xt, r0, rows, Tw, true);
        xcd_barrier(c);
    }
}

}  // namespace

int xcd_barriers_per_launch(int n_layers) { return 1 + 11 * n_layers; }

int xcd_decode_supported(int gdt, int cache_dt, int T, int F, int H, int n_cu) {
    const bool types = (gdt == CAP_DT_G8 && cache_dt == CAP_DT_F32) || (gdt == CAP_DT_BF16 && cache_dt == CAP_DT_BF16);
    return types && T == H * 64 && T % 32 == 0 && F % 32 == 0 && T <= 4 * NTHREAD && n_cu >= 8 && n_cu % 8 == 0;
}

int launch_decode_step_xcd(int gdt, const XParams& p, int n_cu, hipStream_t s) {
    constexpr int LDS = NWAVE * UMAX * MBMAX * 64 * 16;                // 98 304 B: also what keeps it at one workgroup per CU
    const int grid = n_cu / 8 * 8;
    if (gdt == CAP_DT_G8) {
        auto kern = decode_step_xcd_kernel<g8_t, float>;
        if (cap_kernel_setup((const void*)kern, LDS, nullptr) != 0) return -1;
        hipLaunchKernelGGL(kern, dim3(grid), dim3(NTHREAD), LDS, s, p);
    } else if (gdt == CAP_DT_BF16) {
        auto kern = decode_step_xcd_kernel<bf16_t, bf16_t>;
        if (cap_kernel_setup((const void*)kern, LDS, nullptr) != 0) return -1;
        hipLaunchKernelGGL(kern, dim3(grid), dim3(NTHREAD), LDS, s, p);
    } else {
        cap_set_error("launch_decode_step_xcd: unsupported operand type %d", gdt);
        return -1;
    }
    CAP_HIP_CHECK(hipGetLastError());
    return 0;
}
