#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstring>
typedef _Float16 v2h __attribute__((ext_vector_type(2)));
typedef short v4s __attribute__((vector_size(8)));
__global__ void k(unsigned* o, const float* in) {
  __shared__ __attribute__((aligned(16))) unsigned short L[4096];
  for (int i = threadIdx.x; i < 4096; i += 64) L[i] = i;
  __syncthreads();
  // canonical: lane l reads row-major [4 rows per group][16 cols]: addr = (l>>4)*64 + (l&15)*4 elements?  probe: lane supplies its own 8-B chunk
  v4s r = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) v4s*)(L + threadIdx.x * 4));
  uint2 u = __builtin_bit_cast(uint2, r);
  o[2 * threadIdx.x] = u.x; o[2 * threadIdx.x + 1] = u.y;
  float a = in[threadIdx.x], b = in[threadIdx.x + 64];
  int pk = __builtin_amdgcn_cvt_pk_fp8_f32(a, b, 0, false);
  v2h h = __builtin_amdgcn_cvt_scalef32_pk_f16_fp8(pk, 1.0f, false);
  o[128 + threadIdx.x] = __builtin_bit_cast(unsigned, h);
  o[192 + threadIdx.x] = pk;
}
int main() {
  unsigned* o; float* in; hipMalloc(&o, 4096); hipMalloc(&in, 512);
  float hin[128]; for (int i = 0; i < 128; ++i) hin[i] = (i - 64) * 0.37f;
  hipMemcpy(in, hin, 512, hipMemcpyHostToDevice);
  k<<<1, 64>>>(o, in);
  unsigned ho[256]; hipMemcpy(ho, o, 1024, hipMemcpyDeviceToHost);
  for (int l = 0; l < 64; ++l) printf("lane %2d: %4u %4u %4u %4u\n", l, ho[2*l] & 0xffff, ho[2*l] >> 16, ho[2*l+1] & 0xffff, ho[2*l+1] >> 16);
  for (int l = 0; l < 8; ++l) { _Float16 x, y; unsigned v = ho[128 + l]; unsigned short s0 = v & 0xffff, s1 = v >> 16; std::memcpy(&x, &s0, 2); std::memcpy(&y, &s1, 2);
    printf("fp8 rt lane %d: in %g %g -> %g %g (pk %08x)\n", l, hin[l], hin[l + 64], (float)x, (float)y, ho[192 + l]); }
  return 0;
}
