// The k_fourier_h kernels behind gemm_terms = 2 (BASELINE config C5's "bf16"): the same source with every operand rounded to bf16 precision
// before it enters the f16 matrix pipe (split.cuh: IG_BF16_OPERANDS) - bf16 products, fp32 accumulation; hi term only.
#define IG_BF16_OPERANDS 1
#define k_fourier_h k_fourier_h_b16
#define k_fourier_h_multi k_fourier_h_multi_b16
#define g_fh_trace g_fh_trace_b16
#include "fourier_h.hip"
