// Device-side beam search bookkeeping with HuggingFace v5 semantics (HF:generation/utils.py:3010-3204, 3316-3523):
// 2*K candidates per item, -1e9 masking arithmetic in fp32, finished pool with the length penalty
// (cur_len+1-prompt)^lp, early-stop heuristic on cur_len-prompt.
// BEAM_LEGACY_RAW: the reference's CoCa loop instead (coca_model.py:335-482: HF's pre-5.x BeamSearchScorer, one beam group) -
// the same 2K-candidate step with three differences: (1) candidates are scored with RAW logits + the running score (no
// log-softmax; MinLength's -inf on EOS) - coca_model.py:418-425; (2) the start token counts in every length denominator
// (decoder_prompt_len is never passed): finished score = sum / (cur_len + 1)^lp; (3) `BeamHypotheses.is_done` compares the
// pool's worst score with the step's best CANDIDATE (EOS ones included) / (cur_len + 1)^lp, not with the best running beam.
// `finalize` adding the open beams at seq_len is the max-length rule of the last step here (every top-K candidate finishes),
// which selects the same best hypothesis.  The self-attention KV cache is never copied on a
// beam reorder: each row carries a table anc[row][pos] = physical row that wrote position `pos` of its history.
//
//   beam_rows_kernel   one block per (item, beam) row: log-softmax statistics + the row's top-2K candidates
//   beam_merge_kernel  one 64-thread block per item: merge K*2K candidates, update running beams / finished pool /
//                      ancestry (ranking sorts in LDS); the global "loop still running" flag is an atomic OR.
#include "ops.h"

namespace {

constexpr int MAXK = 8;           // beams
constexpr int MAXC = 2 * MAXK;    // candidates kept per item

struct BeamLayout {
    size_t run_seq[2], pool_seq[2], run_score[2], pool_score[2], pool_fin[2], pool_len[2];
    size_t open, active, cand_val, cand_idx, total;
};

inline BeamLayout beam_layout(int B, int K, int L) {
    BeamLayout o;
    size_t off = 0;
    auto take = [&](size_t bytes) { size_t r = off; off += (bytes + 255) & ~(size_t)255; return r; };
    for (int p = 0; p < 2; ++p) {
        o.run_seq[p] = take((size_t)B * K * L * 4);
        o.pool_seq[p] = take((size_t)B * K * L * 4);
        o.run_score[p] = take((size_t)B * K * 4);
        o.pool_score[p] = take((size_t)B * K * 4);
        o.pool_fin[p] = take((size_t)B * K * 4);
        o.pool_len[p] = take((size_t)B * K * 4);
    }
    o.open = take((size_t)B * 4);
    o.active = take(256);
    o.cand_val = take((size_t)B * K * 2 * K * 4);
    o.cand_idx = take((size_t)B * K * 2 * K * 4);
    o.total = off;
    return o;
}

__global__ void beam_init_kernel(char* st, BeamLayout lo, int B, int K, int L, int bos, int fill) {
    const int n = B * K;
    int* rs0 = (int*)(st + lo.run_seq[0]); int* rs1 = (int*)(st + lo.run_seq[1]);
    int* ps0 = (int*)(st + lo.pool_seq[0]); int* ps1 = (int*)(st + lo.pool_seq[1]);
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n * L; i += gridDim.x * blockDim.x) {
        const int v = (i % L == 0) ? bos : fill;
        rs0[i] = v; rs1[i] = v; ps0[i] = v; ps1[i] = v;
    }
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
        const float rsc = (i % K == 0) ? 0.f : -1.0e9f;
        for (int p = 0; p < 2; ++p) {
            ((float*)(st + lo.run_score[p]))[i] = rsc;
            ((float*)(st + lo.pool_score[p]))[i] = -1.0e9f;
            ((int*)(st + lo.pool_fin[p]))[i] = 0;
            ((int*)(st + lo.pool_len[p]))[i] = 0;
        }
    }
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < B; i += gridDim.x * blockDim.x) ((int*)(st + lo.open))[i] = 1;
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        int* flags = (int*)(st + lo.active);
        flags[0] = 1; flags[1] = 0; flags[2] = 0; flags[3] = 0; flags[4] = 0;
    }
}

// (value desc, index asc) strict ordering: is (v,i) after (pv,pi)?
__device__ __forceinline__ bool after(float v, int i, float pv, int pi) { return v < pv || (v == pv && i > pi); }
__device__ __forceinline__ bool better(float v, int i, float bv, int bi) { return v > bv || (v == bv && i < bi); }

__global__ __launch_bounds__(256) void beam_rows_kernel(char* st, BeamLayout lo, const float* __restrict__ logits,
                                                        int ld, int V, int K, int par, int raw, int eos_mask) {
    if (*(const int*)(st + lo.active) == 0) return;
    const int row = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const float* x = logits + (size_t)row * ld;
    __shared__ float redf[8];
    __shared__ int redi[8];
    __shared__ float bc[2];
    // log-softmax statistics, same association order as torch: (x - max) - log(sum(exp(x - max)))
    float m = -INFINITY;
    for (int i = tid; i < V; i += 256) m = fmaxf(m, x[i]);
    m = wave_max(m);
    if (lane == 0) redf[wave] = m;
    __syncthreads();
    m = fmaxf(fmaxf(redf[0], redf[1]), fmaxf(redf[2], redf[3]));
    __syncthreads();
    float sum = 0.f;
    for (int i = tid; i < V; i += 256) sum += expf(x[i] - m);
    sum = wave_sum(sum);
    if (lane == 0) redf[wave] = sum;
    __syncthreads();
    const float lsum = logf(redf[0] + redf[1] + redf[2] + redf[3]);
    const float run = ((const float*)(st + lo.run_score[par]))[row];
    __syncthreads();
    float pv = INFINITY; int pi = -1;
    float* cv = (float*)(st + lo.cand_val) + (size_t)row * 2 * K;
    int* ci = (int*)(st + lo.cand_idx) + (size_t)row * 2 * K;
    for (int c = 0; c < 2 * K; ++c) {
        float bv = -INFINITY; int bi = 0x7fffffff;
        for (int i = tid; i < V; i += 256) {
            float v = raw ? x[i] + run : ((x[i] - m) - lsum) + run;
            if (i == eos_mask) v = -INFINITY;
            if (after(v, i, pv, pi) && better(v, i, bv, bi)) { bv = v; bi = i; }
        }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
            const float v2 = __shfl_xor(bv, o, 64); const int i2 = __shfl_xor(bi, o, 64);
            if (better(v2, i2, bv, bi)) { bv = v2; bi = i2; }
        }
        if (lane == 0) { redf[wave] = bv; redi[wave] = bi; }
        __syncthreads();
        if (tid == 0) {
            float fv = redf[0]; int fi = redi[0];
            for (int w = 1; w < 4; ++w) if (better(redf[w], redi[w], fv, fi)) { fv = redf[w]; fi = redi[w]; }
            cv[c] = fv; ci[c] = fi; bc[0] = fv; ((int*)bc)[1] = fi;
        }
        __syncthreads();
        pv = bc[0]; pi = ((int*)bc)[1];
        __syncthreads();
    }
}

// Single-pass variant of beam_rows_kernel: the row is read from global memory once (16-byte loads) into LDS; max, sum-exp
// and the candidate scan run over the LDS copy; each thread keeps the CT best of ITS elements in a register list
// (unrolled insertion, static indices), and the block then merges the 256 sorted lists by 2K rounds of block arg-max.
// Same arithmetic and the same (value desc, index asc) order as beam_rows_kernel, which stays as the fallback for rows
// that do not fit in LDS.
// STAGE = false: vocabularies whose row does not fit in LDS next to the candidate lists (CoCa: 49408 x 4 B = 193 KiB): the
// same single-scan candidate selection reading the row from global memory (twice more for the log-softmax statistics; not at
// all for raw-logit scores) - the 2K-pass fallback kernel took 874 us per step at 128 x 5 rows there.
template <int CT, bool STAGE = true>
__global__ __launch_bounds__(256) void beam_rows_lds_kernel(char* st, BeamLayout lo, const float* __restrict__ logits,
                                                            int ld, int V, int K, int par, int raw, int eos_mask) {
    if (*(const int*)(st + lo.active) == 0) return;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int V4 = (V + 3) >> 2, C = 2 * K;
    float* xs = (float*)smem;                                  // STAGE: [V4 * 4], tail padded with -inf
    float* lv = xs + (STAGE ? (size_t)V4 * 4 : 0);             // [C][256] candidate values, c-major
    int* li = (int*)(lv + (size_t)C * 256);                    // [C][256] candidate indices
    __shared__ float redf[4];
    __shared__ int redi[4];
    const int row = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const float* x = logits + (size_t)row * ld;
    auto ldg = [&](int j) {                                     // 4 logits of the row, -inf beyond V
        float4 v;
        if (4 * j + 3 < V) v = *(const float4*)(x + 4 * j);
        else {
            v.x = 4 * j < V ? x[4 * j] : -INFINITY; v.y = 4 * j + 1 < V ? x[4 * j + 1] : -INFINITY;
            v.z = 4 * j + 2 < V ? x[4 * j + 2] : -INFINITY; v.w = -INFINITY;
        }
        return v;
    };
    auto ld4 = [&](int j) { return STAGE ? *(const float4*)(xs + 4 * j) : ldg(j); };
    float m = -INFINITY;
    if (STAGE || !raw)
        for (int j = tid; j < V4; j += 256) {
            const float4 v = ldg(j);
            if (STAGE) *(float4*)(xs + 4 * j) = v;
            m = fmaxf(fmaxf(m, fmaxf(v.x, v.y)), fmaxf(v.z, v.w));
        }
    m = wave_max(m);
    if (lane == 0) redf[wave] = m;
    __syncthreads();
    m = fmaxf(fmaxf(redf[0], redf[1]), fmaxf(redf[2], redf[3]));
    __syncthreads();
    // log-softmax statistics, same association order as torch: (x - max) - log(sum(exp(x - max))).  The per-thread
    // element order differs from beam_rows_kernel's (4 consecutive elements per step instead of stride 256), so the fp32
    // sum may differ from it in the last bits - as it does between any two reduction orders; both are tested against
    // the oracle's beam scores.
    float sum = 0.f;
    if (!raw)
        for (int j = tid; j < V4; j += 256) {
            const float4 v = ld4(j);
            sum += expf(v.x - m); sum += expf(v.y - m); sum += expf(v.z - m); sum += expf(v.w - m);   // exp(-inf) = 0 on the pad
        }
    sum = wave_sum(sum);
    if (lane == 0) redf[wave] = sum;
    __syncthreads();
    const float lsum = logf(redf[0] + redf[1] + redf[2] + redf[3]);
    const float run = ((const float*)(st + lo.run_score[par]))[row];
    __syncthreads();
    float* cv = (float*)(st + lo.cand_val) + (size_t)row * C;
    int* ci = (int*)(st + lo.cand_idx) + (size_t)row * C;
    auto score = [&](float e, int i) {
        const float v = raw ? e + run : ((e - m) - lsum) + run;
        return i == eos_mask ? -INFINITY : v;
    };
    // streaming from global memory (vocabulary larger than the LDS): 8 loads of 16 bytes per thread are in flight at a time
    constexpr int U = STAGE ? 1 : 8;
    // Selection by threshold.  (A) every thread finds the best element of its slice; the C-th best of those 256 is a bound tau
    // that at least C elements of the row reach, so the row's best C all lie at or above it.  (B) a second scan appends what
    // reaches tau to a list in LDS (typically C .. 3C entries), which is ranked directly.  A per-thread sorted list of C
    // instead (the fallback below) makes every lane of a wave walk the insertion whenever ONE lane inserts: 500 us per step
    // at 640 rows x 49408 (CoCa, 5 beams).
    constexpr int CAP = 1024;
    __shared__ float sbv[256];
    __shared__ int sbi[256];
    __shared__ float listv[CAP];
    __shared__ int listi[CAP];
    __shared__ int cnt, taui;
    __shared__ float tauv;
    {
        float bv = -INFINITY; int bi = 0x7fffffff;
        for (int j0 = tid; j0 < V4; j0 += 256 * U) {
            float4 qq[U];
#pragma unroll
            for (int w = 0; w < U; ++w) {
                const int j = j0 + w * 256;
                qq[w] = j < V4 ? ld4(j) : make_float4(-INFINITY, -INFINITY, -INFINITY, -INFINITY);
            }
#pragma unroll
            for (int w = 0; w < U; ++w) {
                const int j = j0 + w * 256;
                const float e[4] = {qq[w].x, qq[w].y, qq[w].z, qq[w].w};
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const int i = 4 * j + u;
                    const float v = score(e[u], i);
                    if (i < V && better(v, i, bv, bi)) { bv = v; bi = i; }
                }
            }
        }
        sbv[tid] = bv; sbi[tid] = bi;
        if (tid == 0) { cnt = 0; tauv = -INFINITY; taui = 0x7fffffff; }      // fewer than C slices with data: everything passes
        __syncthreads();
        int rank = 0;
        for (int t = 0; t < 256; ++t) rank += better(sbv[t], sbi[t], bv, bi) ? 1 : 0;
        if (rank == C - 1 && bi != 0x7fffffff) { tauv = bv; taui = bi; }    // indices are unique: exactly one such thread
        __syncthreads();
        const float tv0 = tauv; const int ti0 = taui;
        for (int j0 = tid; j0 < V4; j0 += 256 * U) {
            float4 qq[U];
#pragma unroll
            for (int w = 0; w < U; ++w) {
                const int j = j0 + w * 256;
                qq[w] = j < V4 ? ld4(j) : make_float4(-INFINITY, -INFINITY, -INFINITY, -INFINITY);
            }
#pragma unroll
            for (int w = 0; w < U; ++w) {
                const int j = j0 + w * 256;
                const float e[4] = {qq[w].x, qq[w].y, qq[w].z, qq[w].w};
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const int i = 4 * j + u;
                    const float v = score(e[u], i);
                    if (i < V && !better(tv0, ti0, v, i)) {                   // at or above tau
                        const int pos = atomicAdd(&cnt, 1);
                        if (pos < CAP) { listv[pos] = v; listi[pos] = i; }
                    }
                }
            }
        }
        __syncthreads();
        const int n = cnt;
        if (n <= CAP) {
            for (int e = tid; e < n; e += 256) {
                const float v = listv[e]; const int i = listi[e];
                int r = 0;
                for (int t = 0; t < n; ++t) r += better(listv[t], listi[t], v, i) ? 1 : 0;
                if (r < C) { cv[r] = v; ci[r] = i; }
            }
            return;
        }
        __syncthreads();
    }
    // fallback (more than CAP elements at or above tau: long runs of equal logits): per-thread sorted lists, merged by the block
    float tv[CT]; int ti[CT];
#pragma unroll
    for (int c = 0; c < CT; ++c) { tv[c] = -INFINITY; ti[c] = 0x7fffffff; }
    for (int j0 = tid; j0 < V4; j0 += 256 * U) {
        float4 qq[U];
#pragma unroll
        for (int w = 0; w < U; ++w) {
            const int j = j0 + w * 256;
            qq[w] = j < V4 ? ld4(j) : make_float4(-INFINITY, -INFINITY, -INFINITY, -INFINITY);
        }
#pragma unroll
        for (int w = 0; w < U; ++w) {
            const int j = j0 + w * 256;
            const float e[4] = {qq[w].x, qq[w].y, qq[w].z, qq[w].w};
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int i = 4 * j + u;
                float v = raw ? e[u] + run : ((e[u] - m) - lsum) + run;
                if (i == eos_mask) v = -INFINITY;
                if (i < V && better(v, i, tv[CT - 1], ti[CT - 1])) {
                    tv[CT - 1] = v; ti[CT - 1] = i;
#pragma unroll
                    for (int c = CT - 1; c > 0; --c)
                        if (better(tv[c], ti[c], tv[c - 1], ti[c - 1])) {
                            const float fv = tv[c]; tv[c] = tv[c - 1]; tv[c - 1] = fv;
                            const int fi = ti[c]; ti[c] = ti[c - 1]; ti[c - 1] = fi;
                        }
                }
            }
        }
    }
#pragma unroll
    for (int c = 0; c < CT; ++c)
        if (c < C) { lv[c * 256 + tid] = tv[c]; li[c * 256 + tid] = ti[c]; }
    int h = 0;                                                  // head of this thread's list
    for (int c = 0; c < C; ++c) {
        float bv = h < C ? lv[h * 256 + tid] : -INFINITY;
        int bi = h < C ? li[h * 256 + tid] : 0x7fffffff;
        const int mine = bi;
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
            const float v2 = __shfl_xor(bv, o, 64); const int i2 = __shfl_xor(bi, o, 64);
            if (better(v2, i2, bv, bi)) { bv = v2; bi = i2; }
        }
        if (lane == 0) { redf[wave] = bv; redi[wave] = bi; }
        __syncthreads();
        float fv = redf[0]; int fi = redi[0];
#pragma unroll
        for (int w = 1; w < 4; ++w) if (better(redf[w], redi[w], fv, fi)) { fv = redf[w]; fi = redi[w]; }
        if (mine == fi && fi != 0x7fffffff) ++h;                // indices are unique within a row
        if (tid == 0) { cv[c] = fv; ci[c] = fi; }
        __syncthreads();
    }
}

// One 64-thread block per item.  The three selections of a step (top-2K continuations over the K candidate lists,
// next running beams, finished pool) are RANKING sorts over <= 128 / 16 / 24 entries held in LDS - every lane ranks its
// entries by counting the entries that beat them under exactly the comparators of the serial formulation (value
// descending, then flat index / slot ascending, i.e. "first maximal" of a repeated arg-max) - and the sequence / ancestry
// rows are copied by all lanes.  The loop-still-running flag is an OR over the items: atomics on the state block, the last
// block to finish publishes it and clears the accumulators.
__global__ __launch_bounds__(64) void beam_merge_kernel(char* st, BeamLayout lo, int B, int K, int L, int V,
                                                        int cur_len, int eos, float denom_fin, float denom_run,
                                                        const int* __restrict__ anc_old, int* __restrict__ anc_new,
                                                        int anc_ld, int legacy) {
    int* flags = (int*)(st + lo.active);          // [0] active, [1] parity of the final state, [2] any_open, [3] some_miss, [4] done
    if (flags[0] == 0) return;
    const int b = blockIdx.x, lane = threadIdx.x;
    const int par = cur_len & 1, nxt = par ^ 1, C = 2 * K, NC = K * C;
    const int* rs_old = (const int*)(st + lo.run_seq[par]);   int* rs_new = (int*)(st + lo.run_seq[nxt]);
    const int* ps_old = (const int*)(st + lo.pool_seq[par]);  int* ps_new = (int*)(st + lo.pool_seq[nxt]);
    float* rsc_new = (float*)(st + lo.run_score[nxt]);
    const float* psc_old = (const float*)(st + lo.pool_score[par]); float* psc_new = (float*)(st + lo.pool_score[nxt]);
    const int* pf_old = (const int*)(st + lo.pool_fin[par]);  int* pf_new = (int*)(st + lo.pool_fin[nxt]);
    const int* pl_old = (const int*)(st + lo.pool_len[par]);  int* pl_new = (int*)(st + lo.pool_len[nxt]);
    int* open = (int*)(st + lo.open);

    __shared__ float cvS[MAXK * MAXC];            // candidate lists of the item's K rows
    __shared__ int cfS[MAXK * MAXC];              // flat index k * V + token
    __shared__ float val[MAXC], run_lp[MAXC], msc[MAXK + MAXC], new_sc[MAXK];
    __shared__ int src[MAXC], tok[MAXC], hit[MAXC], run_pick[MAXK], pool_pick[MAXK], new_fin[MAXK];

    // ---- c. top-2K continuations = the C best of the K*C candidates by (value desc, flat index asc)
    const float* cv = (const float*)(st + lo.cand_val) + (size_t)b * NC;
    const int* ci = (const int*)(st + lo.cand_idx) + (size_t)b * NC;
    for (int i = lane; i < NC; i += 64) { cvS[i] = cv[i]; cfS[i] = (i / C) * V + ci[i]; }
    __syncthreads();
    for (int i = lane; i < NC; i += 64) {
        const float v = cvS[i]; const int f = cfS[i];
        int rank = 0;
        for (int j = 0; j < NC; ++j) rank += better(cvS[j], cfS[j], v, f) ? 1 : 0;
        if (rank < C) {
            const int k = i / C, t = f - k * V;
            val[rank] = v; src[rank] = k; tok[rank] = t;
            hit[rank] = (t == eos || cur_len + 1 >= L) ? 1 : 0;
        }
    }
    __syncthreads();
    // ---- e / f inputs
    const bool is_open = open[b] != 0;
    if (lane < C) {
        run_lp[lane] = val[lane] + (hit[lane] ? 1.0f : 0.0f) * -1.0e9f;
        const bool just = hit[lane] && lane < K;
        float f = val[lane] / denom_fin;
        f = f + 0.0f * -1.0e9f;                              // early_stopping is False on this path
        f = f + (is_open ? 0.0f : 1.0f) * -1.0e9f;
        f = f + (just ? 0.0f : 1.0f) * -1.0e9f;
        msc[K + lane] = f;
    }
    if (lane < K) msc[lane] = psc_old[b * K + lane];
    __syncthreads();
    // ---- e. running beams: K best of run_lp, first maximal slot on ties
    if (lane < C) {
        int rank = 0;
        for (int j = 0; j < C; ++j) rank += (run_lp[j] > run_lp[lane] || (run_lp[j] == run_lp[lane] && j < lane)) ? 1 : 0;
        if (rank < K) run_pick[rank] = lane;
    }
    // ---- f. finished pool: K best of the K old entries and the C candidates, first maximal slot on ties
    if (lane < K + C) {
        int rank = 0;
        for (int j = 0; j < K + C; ++j) rank += (msc[j] > msc[lane] || (msc[j] == msc[lane] && j < lane)) ? 1 : 0;
        if (rank < K) pool_pick[rank] = lane;
    }
    __syncthreads();
    for (int k = 0; k < K; ++k) {
        const int r_new = b * K + k;
        {
            const int c = run_pick[k], r_src = b * K + src[c];
            for (int j = lane; j < L; j += 64) rs_new[(size_t)r_new * L + j] = j == cur_len ? tok[c] : rs_old[(size_t)r_src * L + j];
            if (anc_new)
                for (int j = lane; j <= cur_len && j < anc_ld; j += 64)
                    anc_new[(size_t)r_new * anc_ld + j] = j == cur_len ? r_new : anc_old[(size_t)r_src * anc_ld + j];
            if (lane == 0) rsc_new[r_new] = run_lp[c];
        }
        {
            const int bi = pool_pick[k];
            if (bi < K) {
                const int r_old = b * K + bi;
                for (int j = lane; j < L; j += 64) ps_new[(size_t)r_new * L + j] = ps_old[(size_t)r_old * L + j];
                if (lane == 0) { new_fin[k] = pf_old[r_old]; pl_new[r_new] = pl_old[r_old]; }
            } else {
                const int c = bi - K, r_src = b * K + src[c];
                for (int j = lane; j < L; j += 64) ps_new[(size_t)r_new * L + j] = j == cur_len ? tok[c] : rs_old[(size_t)r_src * L + j];
                if (lane == 0) { new_fin[k] = (hit[c] && c < K) ? 1 : 0; pl_new[r_new] = cur_len + 1; }
            }
            if (lane == 0) { new_sc[k] = msc[bi]; psc_new[r_new] = msc[bi]; }
        }
    }
    __syncthreads();
    if (lane == 0) {
        for (int k = 0; k < K; ++k) pf_new[b * K + k] = new_fin[k];
        // ---- g. early-stop heuristic (cur_len already advanced by one)
        float mn = new_sc[0];
        for (int k = 1; k < K; ++k) mn = fminf(mn, new_sc[k]);
        const float best_run = (legacy ? val[0] : run_lp[run_pick[0]]) / denom_run;
        bool any = false, all_hits = true;
        for (int k = 0; k < K; ++k) {
            const float worst = new_fin[k] ? mn : -1.0e9f;
            any = any || (best_run > worst);
        }
        for (int c = 0; c < C; ++c) all_hits = all_hits && hit[c];
        const int o2 = (is_open && any) ? 1 : 0;
        open[b] = o2;
        if (o2) atomicOr(&flags[2], 1);
        if (!all_hits) atomicOr(&flags[3], 1);
        __threadfence();
        if (atomicAdd(&flags[4], 1) == B - 1) {
            __threadfence();
            const int any_open = atomicOr(&flags[2], 0), some_miss = atomicOr(&flags[3], 0);
            flags[1] = (cur_len & 1) ^ 1;          // a stopped loop leaves its final state in this parity: finalize reads it
            flags[2] = 0; flags[3] = 0; flags[4] = 0;
            __threadfence();
            flags[0] = (any_open != 0 && some_miss != 0) ? 1 : 0;
        }
    }
}

__global__ void beam_finalize_kernel(char* st, BeamLayout lo, int B, int K, int L, int* out_ids, int* out_len,
                                     float* out_scores) {
    const int par = ((const int*)(st + lo.active))[1];
    const int* ps = (const int*)(st + lo.pool_seq[par]);
    const float* psc = (const float*)(st + lo.pool_score[par]);
    const int* pl = (const int*)(st + lo.pool_len[par]);
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < B * L; i += gridDim.x * blockDim.x) {
        const int b = i / L, j = i - b * L;
        out_ids[i] = ps[(size_t)b * K * L + j];
    }
    for (int b = blockIdx.x * blockDim.x + threadIdx.x; b < B; b += gridDim.x * blockDim.x) {
        if (out_len) out_len[b] = pl[b * K];
        if (out_scores) out_scores[b] = psc[b * K];
    }
}

}  // namespace

size_t beam_state_bytes(int B, int K, int max_len) { return beam_layout(B, K, max_len).total; }

int launch_beam_init(void* state, int B, int K, int max_len, int bos, int pad, int eos, hipStream_t s, int mode) {
    if (K < 1 || K > MAXK) { cap_set_error("beam search supports 1..%d beams (got %d)", MAXK, K); return -1; }
    // HF v5: `output_fill_value = pad_token_id or eos_token_id[0]` - pad id 0 falls through to EOS.  The legacy scorer's
    // finalize fills with pad_token_id as it is (CoCa: 0).
    const int fill = mode == BEAM_LEGACY_RAW ? pad : (pad ? pad : eos);
    hipLaunchKernelGGL(beam_init_kernel, dim3(64), dim3(256), 0, s, (char*)state, beam_layout(B, K, max_len), B, K,
                       max_len, bos, fill);
    CAP_HIP_CHECK(hipGetLastError());
    return 0;
}

// The 2K best continuations of every running beam -> cand_val / cand_idx of the state block.
static int launch_beam_rows(void* state, const BeamLayout& lo, const float* logits, int ld, int V, int B, int K, int par, int raw,
                            int eos_mask, hipStream_t s) {
    const size_t lds = (size_t)((V + 3) / 4) * 16 + (size_t)2 * K * 256 * 8;
    if (lds <= 148 * 1024 && (ld & 3) == 0 && V > 2 * K) {
        if (cap_kernel_setup((const void*)beam_rows_lds_kernel<4>, 148 * 1024, nullptr) != 0 ||
            cap_kernel_setup((const void*)beam_rows_lds_kernel<8>, 148 * 1024, nullptr) != 0 ||
            cap_kernel_setup((const void*)beam_rows_lds_kernel<16>, 148 * 1024, nullptr) != 0)
            return -1;
        if (2 * K <= 4)
            hipLaunchKernelGGL(beam_rows_lds_kernel<4>, dim3(B * K), dim3(256), lds, s, (char*)state, lo, logits, ld, V, K, par, raw, eos_mask);
        else if (2 * K <= 8)
            hipLaunchKernelGGL(beam_rows_lds_kernel<8>, dim3(B * K), dim3(256), lds, s, (char*)state, lo, logits, ld, V, K, par, raw, eos_mask);
        else
            hipLaunchKernelGGL(beam_rows_lds_kernel<16>, dim3(B * K), dim3(256), lds, s, (char*)state, lo, logits, ld, V, K, par, raw, eos_mask);
    } else if ((ld & 3) == 0 && V > 2 * K) {
        const size_t lds2 = (size_t)2 * K * 256 * 8;                 // candidate lists only
        if (2 * K <= 4)
            hipLaunchKernelGGL((beam_rows_lds_kernel<4, false>), dim3(B * K), dim3(256), lds2, s, (char*)state, lo, logits, ld, V, K, par, raw, eos_mask);
        else if (2 * K <= 8)
            hipLaunchKernelGGL((beam_rows_lds_kernel<8, false>), dim3(B * K), dim3(256), lds2, s, (char*)state, lo, logits, ld, V, K, par, raw, eos_mask);
        else
            hipLaunchKernelGGL((beam_rows_lds_kernel<16, false>), dim3(B * K), dim3(256), lds2, s, (char*)state, lo, logits, ld, V, K, par, raw, eos_mask);
    } else {
        hipLaunchKernelGGL(beam_rows_kernel, dim3(B * K), dim3(256), 0, s, (char*)state, lo, logits, ld, V, K, par, raw, eos_mask);
    }
    CAP_HIP_CHECK(hipGetLastError());
    return 0;
}

int launch_beam_step(void* state, const float* logits, int ld, int V, int B, int K, int max_len, int cur_len,
                     int eos, float length_penalty, int* anc, int anc_ld, hipStream_t s, int mode, int min_len) {
    const BeamLayout lo = beam_layout(B, K, max_len);
    const int par = cur_len & 1;
    const int raw = mode == BEAM_LEGACY_RAW ? 1 : 0;
    const int eos_mask = (raw && cur_len < min_len) ? eos : -1;       // MinLengthLogitsProcessor(min_len, eos)
    if (launch_beam_rows(state, lo, logits, ld, V, B, K, par, raw, eos_mask, s) != 0) return -1;
    // v5: prompt length is 1 ([BOS]); python computes the float power in double, torch divides in fp32.  Legacy scorer: the
    // prompt is not subtracted (generated_len = cur_len + 1 with decoder_prompt_len = 0)
    const int glen = raw ? cur_len + 1 : cur_len + 1 - 1;
    const float denom_fin = (float)pow((double)glen, (double)length_penalty);
    const float denom_run = (float)pow((double)glen, (double)length_penalty);
    const int* anc_old = anc ? anc + (size_t)par * B * K * anc_ld : nullptr;
    int* anc_new = anc ? anc + (size_t)(par ^ 1) * B * K * anc_ld : nullptr;
    hipLaunchKernelGGL(beam_merge_kernel, dim3(B), dim3(64), 0, s, (char*)state, lo, B, K, max_len, V, cur_len, eos,
                       denom_fin, denom_run, anc_old, anc_new, anc_ld, raw);
    CAP_HIP_CHECK(hipGetLastError());
    return 0;
}

int launch_beam_finalize(void* state, int B, int K, int max_len, int* out_ids, int* out_len, float* out_scores,
                         hipStream_t s) {
    hipLaunchKernelGGL(beam_finalize_kernel, dim3(64), dim3(256), 0, s, (char*)state, beam_layout(B, K, max_len), B, K,
                       max_len, out_ids, out_len, out_scores);
    CAP_HIP_CHECK(hipGetLastError());
    return 0;
}

const int* beam_active_flag_p(void* state, int B, int K, int max_len) {
    return (const int*)((char*)state + beam_layout(B, K, max_len).active);
}
const int* beam_running_tokens_p(void* state, int B, int K, int max_len, int parity) {
    return (const int*)((char*)state + beam_layout(B, K, max_len).run_seq[parity]);
}

// test hook (cap_op_beam_candidates): the candidate selection alone on a freshly initialised state (running scores 0 for beam
// 0, -1e9 for the others); the lists land in out_val / out_idx [B*K][2K], best first
int beam_candidates_only(void* state, const float* logits, int ld, int V, int B, int K, int mode, int eos_mask, float* out_val,
                         int* out_idx, hipStream_t s) {
    const BeamLayout lo = beam_layout(B, K, 4);
    if (launch_beam_init(state, B, K, 4, 1, 0, 2, s, mode) != 0) return -1;
    if (launch_beam_rows(state, lo, logits, ld, V, B, K, 0, mode == BEAM_LEGACY_RAW ? 1 : 0, eos_mask, s) != 0) return -1;
    CAP_HIP_CHECK(hipMemcpyAsync(out_val, (char*)state + lo.cand_val, (size_t)B * K * 2 * K * 4, hipMemcpyDeviceToDevice, s));
    CAP_HIP_CHECK(hipMemcpyAsync(out_idx, (char*)state + lo.cand_idx, (size_t)B * K * 2 * K * 4, hipMemcpyDeviceToDevice, s));
    return 0;
}
