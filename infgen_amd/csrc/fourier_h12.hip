// k_fourier_h12: k_fourier_h with THREE wave groups (12 waves, 192-edge tiles, one workgroup per CU; the single-buffered fragment
// schedule of split.cuh: gemm_unit_sb keeps a wave at 168 registers).  The kernel is bound by its vector phases (sine / cosine
// features, LayerNorm, hi / lo splitting: 292 of 409 us with the matrix phases removed, tools/bench_fourier.py) and a third wave per
// SIMD fills the vector pipe better: -3 % of the kernel on large edge sets (+0.85 % per 1024-scene rollout).  Small sets lose (larger
// tiles, fewer of them: -1.4 % at 8 scenes), so the library launches this variant only for large ones (api.hip: fourier_embed_impl).
// Same arithmetic per row in the same order: rows are bitwise equal to k_fourier_h's (tests/test_ops_gpu.py).
#define IG_FH_WAVES 12
#define k_fourier_h k_fourier_h12
#define k_fourier_h_multi k_fourier_h12_multi
#include "fourier_h.hip"
