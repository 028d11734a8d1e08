// attention_mfma.hip: matrix-core attention for dk = dv = 32, Lk <= 128 (see the file header)
#pragma once
#include "common.hpp"

bool mesm_attn_mfma_ok(const MesmAttnArgs& a);
int mesm_attn_mfma_fwd(const MesmAttnArgs& a, hipStream_t s);
bool mesm_attn_mfma_bwd_ok(const MesmAttnArgs& a);
int mesm_attn_mfma_bwd(const MesmAttnArgs& a, hipStream_t s);
// grouped launch of n <= 8 problems for which mesm_attn_mfma_groupable() holds
bool mesm_attn_mfma_groupable(const MesmAttnArgs& a);
int mesm_attn_mfma_fwd_group(const MesmAttnArgs* list, int n, hipStream_t s);
