// Launchers of the fused batch-path decode kernels (decode_tile.hip): 16-row tiles of a batch of any size.
#pragma once
#include "decode_small.h"

// Cross-attention block of a decoder layer for every row of the batch: split-K consumer + LayerNorm of the self-attention output
// projection, the query projection and the attention over the image's K/V in ONE launch (the batch path's reduce_layernorm +
// cq GEMM + decode_attention_online launches).  Same parameter block as the small-batch kernel (SmallCross), any R;
// rows_per_kv must be 1 (greedy: one row per image).  Same bits as the three launches it replaces.
int launch_tile_cross(int dtype, const SmallCross& p, hipStream_t s);
// true when launch_tile_cross takes the shape (the caller falls back to the three launches otherwise)
bool tile_cross_takes(int dtype, const SmallCross& p);
int cap_g8_clamped_decode_tile(unsigned long long* total, int reset);
