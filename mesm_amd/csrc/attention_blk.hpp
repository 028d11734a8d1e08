// 16 x 16-block matrix-core attention backward (attention_blk.hip); dispatched from attention.hip.
#pragma once
#include <hip/hip_runtime.h>

#include "mesm_gfx950.h"

bool mesm_attn_blk_bwd_ok(const MesmAttnArgs& a);
int mesm_attn_blk_bwd(const MesmAttnArgs& a, hipStream_t s);
// n <= 8 problems for which mesm_attn_blk_bwd_ok() holds, one launch
int mesm_attn_blk_bwd_group(const MesmAttnArgs* list, int n, hipStream_t s);
