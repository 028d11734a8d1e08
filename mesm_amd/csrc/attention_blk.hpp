// 16 x 16-block matrix-core attention backward (attention_blk.hip); dispatched from attention.hip.
#pragma once
#include <hip/hip_runtime.h>

#include "mesm_gfx950.h"

bool mesm_attn_blk_bwd_ok(const MesmAttnArgs& a);
bool mesm_attn_blk_bwd_groupable(const MesmAttnArgs& a);  // dk = 32 only
int mesm_attn_blk_bwd(const MesmAttnArgs& a, hipStream_t s);
// n <= 8 problems for which mesm_attn_blk_bwd_ok() holds, one launch
int mesm_attn_blk_bwd_group(const MesmAttnArgs* list, int n, hipStream_t s);

// forward on the same blocks: dk = 32 packed heads (groupable) or dk = 64 split heads, dv = 32, Lk <= 128
bool mesm_attn_blk_fwd_ok(const MesmAttnArgs& a);
bool mesm_attn_blk_fwd_groupable(const MesmAttnArgs& a);
int mesm_attn_blk_fwd(const MesmAttnArgs& a, hipStream_t s);
int mesm_attn_blk_fwd_group(const MesmAttnArgs* list, int n, hipStream_t s);

// long ranges (the head does not fit in LDS): J / I workgroups streaming the other side in chunks
bool mesm_attn_blk_bwd_long_ok(const MesmAttnArgs& a);
int mesm_attn_blk_bwd_long(const MesmAttnArgs& a, hipStream_t s);
bool mesm_attn_blk_fwd_long_ok(const MesmAttnArgs& a);
int mesm_attn_blk_fwd_long(const MesmAttnArgs& a, hipStream_t s);
