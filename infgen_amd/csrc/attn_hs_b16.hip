// The k_attn_hs kernels behind gemm_terms = 2 (BASELINE config C5's "bf16"): the same source with every operand rounded to bf16 precision
// before it enters the f16 matrix pipe (split.cuh: IG_BF16_OPERANDS) - bf16 products, fp32 accumulation; hi term only.
#define IG_BF16_OPERANDS 1
#define k_attn_hs k_attn_hs_b16
#include "attn_hs.hip"
