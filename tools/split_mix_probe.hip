// GPU probe: split_pair through v_fma_mixlo_f16 / v_fma_mixhi_f16 (split.cuh, IG_SPLIT_MIX = 1) against the five-instruction form
// (convert, convert back, subtract, convert) - every fp32 bit pattern of a stride-7 sweep (normal, subnormal, both signs, up to the
// fp16 range the kernels' pre-scaling keeps operands in) must give the same hi and lo bits.  The instruction ORDER matters: a
// consumer straight behind a partial (half-register) write may be served the old half (the dst_sel forwarding hazard hipcc pads for
// in its own code and cannot see inside an asm block).  Variants, each with a consumer (v_mov) as the very next instruction:
//   0  lo, hi, use        1  hi, lo, use (what split.cuh issues)        2  hi0, hi1, lo0, lo1, use1, use0
//   3  lo0, lo1, hi0, hi1, use1, use0  - WRONG on gfx950: v_fma_mixhi_f16 keeps the low half of its destination only straight behind
//      the instruction that wrote it (the first mismatch is printed: which half differs)
//   hipcc --offload-arch=gfx950 -O3 -ffp-contract=off tools/split_mix_probe.hip -o build/split_mix_probe && build/split_mix_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cmath>
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ void split_ref(float a, float b, unsigned& hi, unsigned& lo) {
  const f16x2 h = __builtin_convertvector(f32x2{a, b}, f16x2);
  const f32x2 r = f32x2{a, b} - __builtin_convertvector(h, f32x2);
  hi = __builtin_bit_cast(unsigned, h);
  lo = __builtin_bit_cast(unsigned, __builtin_convertvector(r, f16x2));
}
#define MIXLO(d, s, h) "v_fma_mixlo_f16 " d ", " s ", 1.0, -" h " op_sel:[0,0,0] op_sel_hi:[0,0,1]\n\t"
#define MIXHI(d, s, h) "v_fma_mixhi_f16 " d ", " s ", 1.0, -" h " op_sel:[0,0,1] op_sel_hi:[0,0,1]\n\t"
template <int V>
__device__ __forceinline__ void split_mix2(float a0, float b0, float a1, float b1, unsigned& h0, unsigned& l0, unsigned& h1, unsigned& l1) {
  h0 = __builtin_bit_cast(unsigned, __builtin_convertvector(f32x2{a0, b0}, f16x2));
  h1 = __builtin_bit_cast(unsigned, __builtin_convertvector(f32x2{a1, b1}, f16x2));
  unsigned t0 = 0xdeadbeefu, t1 = 0xdeadbeefu, u0, u1;
  if (V == 0)
    asm volatile(MIXLO("%0", "%4", "%8") MIXHI("%0", "%5", "%8") "v_mov_b32 %2, %0\n\t" MIXLO("%1", "%6", "%9") MIXHI("%1", "%7", "%9") "v_mov_b32 %3, %1"
                 : "+&v"(t0), "+&v"(t1), "=&v"(u0), "=&v"(u1) : "v"(a0), "v"(b0), "v"(a1), "v"(b1), "v"(h0), "v"(h1));
  else if (V == 1)
    asm volatile(MIXHI("%0", "%5", "%8") MIXLO("%0", "%4", "%8") "v_mov_b32 %2, %0\n\t" MIXHI("%1", "%7", "%9") MIXLO("%1", "%6", "%9") "v_mov_b32 %3, %1"
                 : "+&v"(t0), "+&v"(t1), "=&v"(u0), "=&v"(u1) : "v"(a0), "v"(b0), "v"(a1), "v"(b1), "v"(h0), "v"(h1));
  else if (V == 3)
    asm volatile(MIXLO("%0", "%4", "%8") MIXLO("%1", "%6", "%9") MIXHI("%0", "%5", "%8") MIXHI("%1", "%7", "%9") "v_mov_b32 %3, %1\n\tv_mov_b32 %2, %0"
                 : "+&v"(t0), "+&v"(t1), "=&v"(u0), "=&v"(u1) : "v"(a0), "v"(b0), "v"(a1), "v"(b1), "v"(h0), "v"(h1));
  else
    asm volatile(MIXHI("%0", "%5", "%8") MIXHI("%1", "%7", "%9") MIXLO("%0", "%4", "%8") MIXLO("%1", "%6", "%9") "v_mov_b32 %3, %1\n\tv_mov_b32 %2, %0"
                 : "+&v"(t0), "+&v"(t1), "=&v"(u0), "=&v"(u1) : "v"(a0), "v"(b0), "v"(a1), "v"(b1), "v"(h0), "v"(h1));
  l0 = u0; l1 = u1;
}
template <int V>
__global__ void k(unsigned long long* bad, unsigned* first, unsigned long long* seen) {
  const unsigned long long n = 0x100000000ull / 7;
  unsigned long long mism = 0, cnt = 0;
  for (unsigned long long i = blockIdx.x * (unsigned long long)blockDim.x + threadIdx.x; i < n; i += (unsigned long long)gridDim.x * blockDim.x) {
    const unsigned ua = (unsigned)(i * 7), ub = ua ^ 0x80000000u ^ (unsigned)(i * 2654435761u >> 9);
    const float a = __uint_as_float(ua), b = __uint_as_float(ub);
    const float c = a * 0.37f, d = b * -1.7f;
    if (!(fabsf(a) <= 30000.f) || !(fabsf(b) <= 30000.f)) continue;     // (also skips NaN)
    unsigned h0, l0, h1, l1, g0, m0, g1, m1;
    split_ref(a, b, h0, l0);
    split_ref(c, d, h1, l1);
    split_mix2<V>(a, b, c, d, g0, m0, g1, m1);
    cnt += 2;
    const int w = (h0 != g0 || l0 != m0) + (h1 != g1 || l1 != m1);
    if (w) { if (!mism) { const bool p0 = l0 != m0; first[0] = p0 ? ua : __float_as_uint(c); first[1] = p0 ? ub : __float_as_uint(d); first[2] = p0 ? l0 : l1; first[3] = p0 ? m0 : m1; } mism += w; }
  }
  atomicAdd(bad, mism);
  atomicAdd(seen, cnt);
}
int main() {
  unsigned long long *bad, *seen; unsigned* first;
  (void)hipMalloc(&bad, 8); (void)hipMalloc(&seen, 8); (void)hipMalloc(&first, 16);
  int rc = 0;
  for (int v = 0; v < 4; ++v) {
    (void)hipMemset(bad, 0, 8); (void)hipMemset(seen, 0, 8); (void)hipMemset(first, 0, 16);
    if (v == 0) hipLaunchKernelGGL(k<0>, dim3(2048), dim3(256), 0, 0, bad, first, seen);
    if (v == 1) hipLaunchKernelGGL(k<1>, dim3(2048), dim3(256), 0, 0, bad, first, seen);
    if (v == 2) hipLaunchKernelGGL(k<2>, dim3(2048), dim3(256), 0, 0, bad, first, seen);
    if (v == 3) hipLaunchKernelGGL(k<3>, dim3(2048), dim3(256), 0, 0, bad, first, seen);
    unsigned long long hb = 0, hs = 0; unsigned hf[4];
    (void)hipMemcpy(&hb, bad, 8, hipMemcpyDeviceToHost); (void)hipMemcpy(&hs, seen, 8, hipMemcpyDeviceToHost); (void)hipMemcpy(hf, first, 16, hipMemcpyDeviceToHost);
    printf("variant %d: pairs checked %llu, mismatching %llu", v, hs, hb);
    if (hb) printf(" (first: a = %08x b = %08x lo %08x vs %08x)", hf[0], hf[1], hf[2], hf[3]);
    printf("\n");
    if (v != 3 && hb) rc = 1;
  }
  return rc;
}
