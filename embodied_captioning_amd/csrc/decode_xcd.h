// Persistent decode-step kernel (decode_xcd.hip): all layers of one BLIP decode step in one launch, rows partitioned by XCD.
#pragma once
#include "ops.h"

struct XLayer {                       // device-side view of one text-decoder layer (built once per handle)
    const void *w_qkv, *w_so, *w_cq, *w_co, *w_f1, *w_f2;
    const float *b_qkv, *b_so, *so_g, *so_b, *b_cq, *b_co, *co_g, *co_b, *b_f1, *b_f2, *f_g, *f_b;
    void *kc, *vc;                    // self-attention cache of the slice: [row][head][max_len][64] each
    const void *ck, *cv;              // beam-shared cross-attention K / V of the slice's first image: [image][head][NT][64]
};

constexpr int XCD_MAX_LAYERS = 16;    // the parameter block travels by value (kernel arguments: < 4 KiB)

struct XParams {
    XLayer layers[XCD_MAX_LAYERS]; int n_layers;
    int R, T, F, H, NT, Lm, t, K;     // rows, hidden, ffn, heads, image tokens, cache positions, position of this step, rows per image
    float eps;
    float* x; void* xt;               // residual stream fp32 [R, T] / the same as GEMM operand
    float* qkv; void* ctx; float* q; float* tmp; void* h;
    const int* anc; int anc_ld;       // beam ancestry of the self-attention cache (nullptr: greedy)
    const int* skip;                  // rows whose caption has ended (nullptr: none)
    const int* tokens; int tok_ld;    // newest token of row r at tokens[r * tok_ld + t]
    const float *word, *pos, *emb_g, *emb_b;
    int* bar;                         // [8][64] XCD barrier counters (monotonic)
    unsigned bar_base[8];             // their values before this launch
    int* err;                         // bit 0: a barrier spin timed out, bit 1: an XCD did not get its share of workgroups
    int* reg;                         // [8][64] workgroup registration counters (monotonic) and their values before this launch
    unsigned reg_base[8];
    long long* dbg;                   // optional: 100 MHz timestamps after every barrier of XCD 0's first workgroup
};

// XCD-local barriers one launch executes on an XCD that owns rows
int xcd_barriers_per_launch(int n_layers);
// can this configuration run on the persistent kernel? (operand / cache types, widths, a CU count that is a multiple of 8)
int xcd_decode_supported(int gdt, int cache_dt, int T, int F, int H, int n_cu);
int launch_decode_step_xcd(int gdt, const XParams& p, int n_cu, hipStream_t s);
