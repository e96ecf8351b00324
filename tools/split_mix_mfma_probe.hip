// GPU probe: how many wait states does an MFMA need behind v_fma_mixlo_f16 / v_fma_mixhi_f16 writes of its B operand?
// (hipcc pads for the hazards of its own code; inside an asm block the kernel author does.)  For NOPS = 0 .. 4 and both orders of the
// two half-register writes: four (lo, hi) pairs are written into v[100:103], NOPS x s_nop 0, then v_mfma_f32_16x16x32_f16 reads them;
// the result is compared with the same MFMA on operands built by plain C++.
//   hipcc --offload-arch=gfx950 -O3 -ffp-contract=off tools/split_mix_mfma_probe.hip -o build/split_mix_mfma_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <random>
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));
typedef _Float16 v8h __attribute__((ext_vector_type(8)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
#define LO(r, a, h) "v_fma_mixlo_f16 " r ", %[" a "], 1.0, -%[" h "] op_sel:[0,0,0] op_sel_hi:[0,0,1]\n\t"
#define HI(r, b, h) "v_fma_mixhi_f16 " r ", %[" b "], 1.0, -%[" h "] op_sel:[0,0,1] op_sel_hi:[0,0,1]\n\t"
template <int NOPS, bool HI_FIRST>
__global__ void k(const float* x, unsigned long long* bad, int iters) {
  const int lane = threadIdx.x & 63;
  unsigned long long mism = 0;
  for (int it = 0; it < iters; ++it) {
    float a[8];
    for (int q = 0; q < 8; ++q) a[q] = x[((size_t)(blockIdx.x * iters + it) * 64 + lane) * 8 + q];
    u32x4 hi, lo_ref;
    for (int p = 0; p < 4; ++p) {
      const f16x2 h = __builtin_convertvector(f32x2{a[2 * p], a[2 * p + 1]}, f16x2);
      const f32x2 r = f32x2{a[2 * p], a[2 * p + 1]} - __builtin_convertvector(h, f32x2);
      hi[p] = __builtin_bit_cast(unsigned, h);
      lo_ref[p] = __builtin_bit_cast(unsigned, __builtin_convertvector(r, f16x2));
    }
    v8h A;
    for (int q = 0; q < 8; ++q) A[q] = (_Float16)(1.0f + 0.125f * ((lane + q) & 7));
    const f32x4 ref = __builtin_amdgcn_mfma_f32_16x16x32_f16(A, __builtin_bit_cast(v8h, lo_ref), f32x4{0.f, 0.f, 0.f, 0.f}, 0, 0, 0);
    f32x4 d;
    const unsigned h0 = hi[0], h1 = hi[1], h2 = hi[2], h3 = hi[3];
    if (HI_FIRST)
      asm volatile("v_mov_b32 v100, 0x7c007c00\n\tv_mov_b32 v101, 0x7c007c00\n\tv_mov_b32 v102, 0x7c007c00\n\tv_mov_b32 v103, 0x7c007c00\n\ts_nop 4\n\t"
                   HI("v100", "a1", "h0") HI("v101", "a3", "h1") HI("v102", "a5", "h2") HI("v103", "a7", "h3")
                   LO("v100", "a0", "h0") LO("v101", "a2", "h1") LO("v102", "a4", "h2") LO("v103", "a6", "h3")
                   ".rept %[n]\n\ts_nop 0\n\t.endr\n\t"
                   "v_mfma_f32_16x16x32_f16 %[d], %[A], v[100:103], 0\n\ts_nop 7\n\ts_nop 7\n\ts_nop 7"
                   : [d] "=&v"(d)
                   : [A] "v"(A), [h0] "v"(h0), [h1] "v"(h1), [h2] "v"(h2), [h3] "v"(h3), [a0] "v"(a[0]), [a1] "v"(a[1]), [a2] "v"(a[2]),
                     [a3] "v"(a[3]), [a4] "v"(a[4]), [a5] "v"(a[5]), [a6] "v"(a[6]), [a7] "v"(a[7]), [n] "n"(NOPS)
                   : "v100", "v101", "v102", "v103");
    else
      asm volatile("v_mov_b32 v100, 0x7c007c00\n\tv_mov_b32 v101, 0x7c007c00\n\tv_mov_b32 v102, 0x7c007c00\n\tv_mov_b32 v103, 0x7c007c00\n\ts_nop 4\n\t"
                   LO("v100", "a0", "h0") HI("v100", "a1", "h0") LO("v101", "a2", "h1") HI("v101", "a3", "h1")
                   LO("v102", "a4", "h2") HI("v102", "a5", "h2") LO("v103", "a6", "h3") HI("v103", "a7", "h3")
                   ".rept %[n]\n\ts_nop 0\n\t.endr\n\t"
                   "v_mfma_f32_16x16x32_f16 %[d], %[A], v[100:103], 0\n\ts_nop 7\n\ts_nop 7\n\ts_nop 7"
                   : [d] "=&v"(d)
                   : [A] "v"(A), [h0] "v"(h0), [h1] "v"(h1), [h2] "v"(h2), [h3] "v"(h3), [a0] "v"(a[0]), [a1] "v"(a[1]), [a2] "v"(a[2]),
                     [a3] "v"(a[3]), [a4] "v"(a[4]), [a5] "v"(a[5]), [a6] "v"(a[6]), [a7] "v"(a[7]), [n] "n"(NOPS)
                   : "v100", "v101", "v102", "v103");
    for (int q = 0; q < 4; ++q) mism += __float_as_uint(d[q]) != __float_as_uint(ref[q]);
  }
  if (mism) atomicAdd(bad, mism);
}
template <int NOPS, bool HF>
static void run(const float* x, unsigned long long* bad, int blocks, int iters) {
  (void)hipMemset(bad, 0, 8);
  hipLaunchKernelGGL((k<NOPS, HF>), dim3(blocks), dim3(64), 0, 0, x, bad, iters);
  unsigned long long hb = 0;
  (void)hipMemcpy(&hb, bad, 8, hipMemcpyDeviceToHost);
  printf("%s first, %d x s_nop 0 in front of the MFMA: %llu of %llu results differ\n", HF ? "hi" : "lo", NOPS, hb, (unsigned long long)blocks * iters * 64 * 4);
}

// experiments (see main): instruction patterns around the partial writes
//   GAP    s_nop 7 repeated GAP times between a first MFMA (v[104:107] = A x hi) and the partial writes (0: in its shadow)
//   FIRST  0: no first MFMA (C = 0)
//   ORDER  0: lo0 hi0 lo1 hi1 ..   1: lo2 lo3 lo0 lo1 hi2 hi3 hi0 hi1 (what hipcc scheduled from two non-volatile asm statements)
#define PAIRS_ADJ LO("v100", "a0", "h0") HI("v100", "a1", "h0") LO("v101", "a2", "h1") HI("v101", "a3", "h1") LO("v102", "a4", "h2") HI("v102", "a5", "h2") LO("v103", "a6", "h3") HI("v103", "a7", "h3")
#define PAIRS_GRP LO("v102", "a4", "h2") LO("v103", "a6", "h3") LO("v100", "a0", "h0") LO("v101", "a2", "h1") HI("v102", "a5", "h2") HI("v103", "a7", "h3") HI("v100", "a1", "h0") HI("v101", "a3", "h1")
#define XBODY(FIRSTS, PAIRS, CSRC) \
      asm volatile("v_mov_b32 v100, 0x7c007c00\n\tv_mov_b32 v101, 0x7c007c00\n\tv_mov_b32 v102, 0x7c007c00\n\tv_mov_b32 v103, 0x7c007c00\n\ts_nop 4\n\t" \
                   FIRSTS ".rept %[gap]\n\ts_nop 7\n\t.endr\n\t" PAIRS \
                   ".rept %[wait]\n\ts_nop 7\n\t.endr\n\t" \
                   "s_nop 0\n\tv_mfma_f32_16x16x32_f16 v[108:111], %[A], v[100:103], " CSRC "\n\ts_nop 7\n\ts_nop 7\n\ts_nop 7\n\t" \
                   "v_mov_b32 %[d0], v108\n\tv_mov_b32 %[d1], v109\n\tv_mov_b32 %[d2], v110\n\tv_mov_b32 %[d3], v111" \
                   : [d0] "=&v"(d[0]), [d1] "=&v"(d[1]), [d2] "=&v"(d[2]), [d3] "=&v"(d[3]) \
                   : [A] "v"(A), [H] "v"(hi), [h0] "v"(h0), [h1] "v"(h1), [h2] "v"(h2), [h3] "v"(h3), [a0] "v"(a[0]), [a1] "v"(a[1]), [a2] "v"(a[2]), \
                     [a3] "v"(a[3]), [a4] "v"(a[4]), [a5] "v"(a[5]), [a6] "v"(a[6]), [a7] "v"(a[7]), [gap] "n"(FIRST ? GAP : 0), [wait] "n"(FIRST ? 0 : GAP) \
                   : "v100", "v101", "v102", "v103", "v104", "v105", "v106", "v107", "v108", "v109", "v110", "v111")
template <int GAP, int FIRST, int ORDER>
__global__ void kx(const float* x, unsigned long long* bad, int iters) {
  const int lane = threadIdx.x & 63;
  unsigned long long mism = 0;
  for (int it = 0; it < iters; ++it) {
    float a[8];
    for (int q = 0; q < 8; ++q) a[q] = x[((size_t)(blockIdx.x * iters + it) * 64 + lane) * 8 + q];
    u32x4 hi, lo_ref;
    for (int p = 0; p < 4; ++p) {
      const f16x2 h = __builtin_convertvector(f32x2{a[2 * p], a[2 * p + 1]}, f16x2);
      const f32x2 r = f32x2{a[2 * p], a[2 * p + 1]} - __builtin_convertvector(h, f32x2);
      hi[p] = __builtin_bit_cast(unsigned, h);
      lo_ref[p] = __builtin_bit_cast(unsigned, __builtin_convertvector(r, f16x2));
    }
    v8h A;
    for (int q = 0; q < 8; ++q) A[q] = (_Float16)(1.0f + 0.125f * ((lane + q) & 7));
    f32x4 c0 = {0.f, 0.f, 0.f, 0.f};
    if (FIRST) c0 = __builtin_amdgcn_mfma_f32_16x16x32_f16(A, __builtin_bit_cast(v8h, hi), c0, 0, 0, 0);
    const f32x4 ref = __builtin_amdgcn_mfma_f32_16x16x32_f16(A, __builtin_bit_cast(v8h, lo_ref), c0, 0, 0, 0);
    f32x4 d;
    const unsigned h0 = hi[0], h1 = hi[1], h2 = hi[2], h3 = hi[3];
    if (FIRST && ORDER == 0) XBODY("v_mfma_f32_16x16x32_f16 v[104:107], %[A], %[H], 0\n\t", PAIRS_ADJ, "v[104:107]");
    if (FIRST && ORDER == 1) XBODY("v_mfma_f32_16x16x32_f16 v[104:107], %[A], %[H], 0\n\t", PAIRS_GRP, "v[104:107]");
    if (!FIRST && ORDER == 0) XBODY("", PAIRS_ADJ, "0");
    if (!FIRST && ORDER == 1) XBODY("", PAIRS_GRP, "0");
    for (int q = 0; q < 4; ++q) mism += __float_as_uint(d[q]) != __float_as_uint(ref[q]);
  }
  if (mism) atomicAdd(bad, mism);
}
template <int GAP, int FIRST, int ORDER>
static void runx(const float* x, unsigned long long* bad, int blocks, int iters) {
  (void)hipMemset(bad, 0, 8);
  hipLaunchKernelGGL((kx<GAP, FIRST, ORDER>), dim3(blocks), dim3(64), 0, 0, x, bad, iters);
  unsigned long long hb = 0;
  (void)hipMemcpy(&hb, bad, 8, hipMemcpyDeviceToHost);
  printf("first MFMA %d, %s %d x s_nop 7, order %s: %llu of %llu results differ\n", FIRST, FIRST ? "gap behind it" : "wait in front of the consumer", GAP, ORDER ? "grouped" : "adjacent", hb,
         (unsigned long long)blocks * iters * 64 * 4);
}
int main() {
  const int blocks = 1024, iters = 64;
  std::vector<float> h((size_t)blocks * iters * 64 * 8);
  std::mt19937 g(1);
  std::normal_distribution<float> nd(0.f, 100.f);
  for (auto& v : h) v = nd(g);
  float* x; unsigned long long* bad;
  (void)hipMalloc(&x, h.size() * 4); (void)hipMalloc(&bad, 8);
  (void)hipMemcpy(x, h.data(), h.size() * 4, hipMemcpyHostToDevice);
  run<0, false>(x, bad, blocks, iters); run<1, false>(x, bad, blocks, iters); run<2, false>(x, bad, blocks, iters);
  run<3, false>(x, bad, blocks, iters); run<4, false>(x, bad, blocks, iters);
  run<0, true>(x, bad, blocks, iters); run<1, true>(x, bad, blocks, iters); run<2, true>(x, bad, blocks, iters);
  run<3, true>(x, bad, blocks, iters); run<4, true>(x, bad, blocks, iters);
  runx<0, 0, 0>(x, bad, blocks, iters); runx<0, 0, 1>(x, bad, blocks, iters);
  runx<1, 0, 1>(x, bad, blocks, iters); runx<4, 0, 1>(x, bad, blocks, iters); runx<16, 0, 1>(x, bad, blocks, iters);
  runx<0, 1, 0>(x, bad, blocks, iters); runx<0, 1, 1>(x, bad, blocks, iters);
  runx<2, 1, 0>(x, bad, blocks, iters); runx<2, 1, 1>(x, bad, blocks, iters);
  runx<8, 1, 0>(x, bad, blocks, iters); runx<8, 1, 1>(x, bad, blocks, iters);
  return 0;
}
