// Device-side beam search bookkeeping with HuggingFace v5 semantics (HF:generation/utils.py:3010-3204, 3316-3523):
// 2*K candidates per item, -1e9 masking arithmetic in fp32, finished pool with the length penalty
// (cur_len+1-prompt)^lp, early-stop heuristic on cur_len-prompt.  The self-attention KV cache is never copied on a
// beam reorder: each row carries a table anc[row][pos] = physical row that wrote position `pos` of its history.
//
//   beam_rows_kernel   one block per (item, beam) row: log-softmax statistics + the row's top-2K candidates
//   beam_merge_kernel  one thread per item: merge K*2K candidates, update running beams / finished pool / ancestry;
//                      then one reduction for the global "loop still running" flag.
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
    if (blockIdx.x == 0 && threadIdx.x == 0) *(int*)(st + lo.active) = 1;
}

// (value desc, index asc) strict ordering: is (v,i) after (pv,pi)?
__device__ __forceinline__ bool after(float v, int i, float pv, int pi) { return v < pv || (v == pv && i > pi); }
__device__ __forceinline__ bool better(float v, int i, float bv, int bi) { return v > bv || (v == bv && i < bi); }

__global__ __launch_bounds__(256) void beam_rows_kernel(char* st, BeamLayout lo, const float* __restrict__ logits,
                                                        int ld, int V, int K, int par) {
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
            const float v = ((x[i] - m) - lsum) + run;
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

__global__ __launch_bounds__(256) void beam_merge_kernel(char* st, BeamLayout lo, int B, int K, int L, int V,
                                                         int cur_len, int eos, float denom_fin, float denom_run,
                                                         const int* __restrict__ anc_old, int* __restrict__ anc_new,
                                                         int anc_ld) {
    __shared__ int s_any_open, s_all_hits;
    const int active = *(const int*)(st + lo.active);
    if (threadIdx.x == 0) { s_any_open = 0; s_all_hits = 1; }
    __syncthreads();
    if (active) {
        const int par = cur_len & 1, nxt = par ^ 1, C = 2 * K;
        const int* rs_old = (const int*)(st + lo.run_seq[par]);   int* rs_new = (int*)(st + lo.run_seq[nxt]);
        const int* ps_old = (const int*)(st + lo.pool_seq[par]);  int* ps_new = (int*)(st + lo.pool_seq[nxt]);
        const float* rsc_old = (const float*)(st + lo.run_score[par]); float* rsc_new = (float*)(st + lo.run_score[nxt]);
        const float* psc_old = (const float*)(st + lo.pool_score[par]); float* psc_new = (float*)(st + lo.pool_score[nxt]);
        const int* pf_old = (const int*)(st + lo.pool_fin[par]);  int* pf_new = (int*)(st + lo.pool_fin[nxt]);
        const int* pl_old = (const int*)(st + lo.pool_len[par]);  int* pl_new = (int*)(st + lo.pool_len[nxt]);
        int* open = (int*)(st + lo.open);
        (void)rsc_old;
        for (int b = threadIdx.x; b < B; b += blockDim.x) {
            // ---- c. top-2K continuations over the K rows' candidate lists (each already sorted)
            const float* cv = (const float*)(st + lo.cand_val) + (size_t)b * K * C;
            const int* ci = (const int*)(st + lo.cand_idx) + (size_t)b * K * C;
            int head[MAXK];
            for (int k = 0; k < K; ++k) head[k] = 0;
            float val[MAXC]; int src[MAXC], tok[MAXC]; bool hit[MAXC];
            bool all_hits = true;
            for (int c = 0; c < C; ++c) {
                float bv = -INFINITY; int bk = -1, bflat = 0x7fffffff;
                for (int k = 0; k < K; ++k) {
                    if (head[k] >= C) continue;
                    const float v = cv[k * C + head[k]];
                    const int flat = k * V + ci[k * C + head[k]];
                    if (bk < 0 || better(v, flat, bv, bflat)) { bv = v; bk = k; bflat = flat; }
                }
                val[c] = bv; src[c] = bk; tok[c] = ci[bk * C + head[bk]]; head[bk]++;
                hit[c] = (tok[c] == eos) || (cur_len + 1 >= L);
                all_hits = all_hits && hit[c];
            }
            // ---- e. running beams for the next iteration
            float run_lp[MAXC];
            for (int c = 0; c < C; ++c) run_lp[c] = val[c] + (hit[c] ? 1.0f : 0.0f) * -1.0e9f;
            bool used[MAXC];
            for (int c = 0; c < C; ++c) used[c] = false;
            for (int k = 0; k < K; ++k) {
                int bc_ = -1;
                for (int c = 0; c < C; ++c)
                    if (!used[c] && (bc_ < 0 || run_lp[c] > run_lp[bc_])) bc_ = c;
                used[bc_] = true;
                const int r_new = b * K + k, r_src = b * K + src[bc_];
                for (int j = 0; j < cur_len; ++j) rs_new[(size_t)r_new * L + j] = rs_old[(size_t)r_src * L + j];
                rs_new[(size_t)r_new * L + cur_len] = tok[bc_];
                for (int j = cur_len + 1; j < L; ++j) rs_new[(size_t)r_new * L + j] = rs_old[(size_t)r_src * L + j];
                rsc_new[r_new] = run_lp[bc_];
                if (anc_new) {
                    for (int j = 0; j < cur_len; ++j) anc_new[(size_t)r_new * anc_ld + j] = anc_old[(size_t)r_src * anc_ld + j];
                    if (cur_len < anc_ld) anc_new[(size_t)r_new * anc_ld + cur_len] = r_new;
                }
            }
            // ---- f. finished pool
            const bool is_open = open[b] != 0;
            float msc[MAXK + MAXC];
            for (int k = 0; k < K; ++k) msc[k] = psc_old[b * K + k];
            for (int c = 0; c < C; ++c) {
                const bool just = hit[c] && c < K;
                float f = val[c] / denom_fin;
                f = f + 0.0f * -1.0e9f;                              // early_stopping is False on this path
                f = f + (is_open ? 0.0f : 1.0f) * -1.0e9f;
                f = f + (just ? 0.0f : 1.0f) * -1.0e9f;
                msc[K + c] = f;
            }
            bool mused[MAXK + MAXC];
            for (int i = 0; i < K + C; ++i) mused[i] = false;
            float new_sc[MAXK]; int new_fin[MAXK];
            for (int k = 0; k < K; ++k) {
                int bi = -1;
                for (int i = 0; i < K + C; ++i)
                    if (!mused[i] && (bi < 0 || msc[i] > msc[bi])) bi = i;
                mused[bi] = true;
                const int r_new = b * K + k;
                if (bi < K) {
                    const int r_old = b * K + bi;
                    for (int j = 0; j < L; ++j) ps_new[(size_t)r_new * L + j] = ps_old[(size_t)r_old * L + j];
                    new_fin[k] = pf_old[r_old]; pl_new[r_new] = pl_old[r_old];
                } else {
                    const int c = bi - K, r_src = b * K + src[c];
                    for (int j = 0; j < L; ++j) ps_new[(size_t)r_new * L + j] = rs_old[(size_t)r_src * L + j];
                    ps_new[(size_t)r_new * L + cur_len] = tok[c];
                    new_fin[k] = (hit[c] && c < K) ? 1 : 0; pl_new[r_new] = cur_len + 1;
                }
                new_sc[k] = msc[bi]; psc_new[r_new] = msc[bi]; pf_new[r_new] = new_fin[k];
            }
            // ---- g. early-stop heuristic (cur_len already advanced by one)
            float mn = new_sc[0];
            for (int k = 1; k < K; ++k) mn = fminf(mn, new_sc[k]);
            const float best_run = rsc_new[b * K] / denom_run;
            bool any = false;
            for (int k = 0; k < K; ++k) {
                const float worst = new_fin[k] ? mn : -1.0e9f;
                any = any || (best_run > worst);
            }
            const int o2 = (is_open && any) ? 1 : 0;
            open[b] = o2;
            if (o2) atomicOr(&s_any_open, 1);
            if (!all_hits) atomicAnd(&s_all_hits, 0);
        }
    }
    __syncthreads();
    if (threadIdx.x == 0 && active) {
        const int go = (s_any_open != 0) && (s_all_hits == 0);
        *(int*)(st + lo.active) = go;
        // a stopped loop leaves its final state in parity `cur_len & 1 ^ 1`; remember it for finalize
        ((int*)(st + lo.active))[1] = (cur_len & 1) ^ 1;
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

int launch_beam_init(void* state, int B, int K, int max_len, int bos, int pad, int eos, hipStream_t s) {
    if (K < 1 || K > MAXK) { cap_set_error("beam search supports 1..%d beams (got %d)", MAXK, K); return -1; }
    // HF: `output_fill_value = pad_token_id or eos_token_id[0]` - pad id 0 falls through to EOS.
    const int fill = pad ? pad : eos;
    hipLaunchKernelGGL(beam_init_kernel, dim3(64), dim3(256), 0, s, (char*)state, beam_layout(B, K, max_len), B, K,
                       max_len, bos, fill);
    CAP_HIP_CHECK(hipGetLastError());
    return 0;
}

int launch_beam_step(void* state, const float* logits, int ld, int V, int B, int K, int max_len, int cur_len,
                     int eos, float length_penalty, int* anc, int anc_ld, hipStream_t s) {
    const BeamLayout lo = beam_layout(B, K, max_len);
    const int par = cur_len & 1;
    hipLaunchKernelGGL(beam_rows_kernel, dim3(B * K), dim3(256), 0, s, (char*)state, lo, logits, ld, V, K, par);
    CAP_HIP_CHECK(hipGetLastError());
    // prompt length is 1 ([BOS]); python computes the float power in double, torch divides in fp32
    const float denom_fin = (float)pow((double)(cur_len + 1 - 1), (double)length_penalty);
    const float denom_run = (float)pow((double)(cur_len + 1 - 1), (double)length_penalty);
    const int* anc_old = anc ? anc + (size_t)par * B * K * anc_ld : nullptr;
    int* anc_new = anc ? anc + (size_t)(par ^ 1) * B * K * anc_ld : nullptr;
    hipLaunchKernelGGL(beam_merge_kernel, dim3(1), dim3(256), 0, s, (char*)state, lo, B, K, max_len, V, cur_len, eos,
                       denom_fin, denom_run, anc_old, anc_new, anc_ld);
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

const int* beam_running_tokens_p(void* state, int B, int K, int max_len, int parity) {
    return (const int*)((char*)state + beam_layout(B, K, max_len).run_seq[parity]);
}
